#!/bin/bash
set -x
mkdir -p gpurun_out/c6
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "fused_qkv or bench_distributed" > gpurun_out/c6/pytest.log 2>&1; tail -5 gpurun_out/c6/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --decode-steps 0 --c4-steps 0 > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c6/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['config'].get('ms_per_step_hipevent_median'), d['config']['kernel_ms_per_step'])
PY
