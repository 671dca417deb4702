#!/bin/bash
set -x
mkdir -p gpurun_out/c5
export VT_PARITY_REPORT=$PWD/gpurun_out/c5/parity_fullwidth.json
timeout 1800 python -m pytest tests -m gpu -q -s --durations=15 > gpurun_out/c5/pytest.log 2>&1
tail -32 gpurun_out/c5/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c5/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['config'].get('ms_per_step_hipevent_median'), d['config'].get('c4',{}).get('tokens_per_s'), d['config']['kernel_ms_per_step'])
print(d.get('decode',{}).get('ms_per_step'))
PY
