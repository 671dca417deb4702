#!/bin/bash
mkdir -p gpurun_out/c40
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c40/bench.json 2> gpurun_out/c40/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c40/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['config'].get('c4',{}).get('tokens_per_s'), d['config']['kernel_ms_per_step'], d.get('decode',{}).get('ms_per_step'))
PY
timeout 120 python tools/attn_bench.py 2>&1 | grep '"S"' | head -2
export VT_PARITY_REPORT=/root/repo/gpurun_out/c40/parity.json
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
