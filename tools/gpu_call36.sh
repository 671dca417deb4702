#!/bin/bash
mkdir -p gpurun_out/c36
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --decode-steps 0 > gpurun_out/c36/bench.json 2> gpurun_out/c36/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c36/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['config'].get('c4',{}).get('tokens_per_s'), d['config']['kernel_ms_per_step'])
PY
tail -2 gpurun_out/c36/bench.err
timeout 300 python tools/c2_bench.py 10 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
