#!/bin/bash
mkdir -p gpurun_out/c9
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -q -s -k "fused or rope or decode" > gpurun_out/c9/pytest.log 2>&1; grep -E "K mismatch by layer|passed|failed|Error" gpurun_out/c9/pytest.log | cut -c1-300 | head
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --decode-steps 0 --c4-steps 0 > gpurun_out/c9/bench.json 2> gpurun_out/c9/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c9/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['config'].get('ms_per_step_hipevent_median'), d['config']['kernel_ms_per_step'])
PY
bash tools/gpu_profiles.sh r2b > gpurun_out/c9/prof.log 2>&1; tail -25 gpurun_out/c9/prof.log
python tools/kstat.py $(find gpurun_out/prof_r2b/bench_stats -name "*kernel_stats.csv" | head -1) gemm_p8 gemm_bt kv_tiles flash rmsnorm splitk layernorm row_slot region temporal
