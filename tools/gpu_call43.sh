#!/bin/bash
mkdir -p gpurun_out/c43
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/c43/bench.json 2> gpurun_out/c43/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c43/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['config'].get('ms_per_step_hipevent_median'), d['config'].get('c4',{}).get('tokens_per_s'), d['config']['kernel_ms_per_step'])
print(d.get('decode',{}).get('ms_per_step'), d.get('cpu_baseline',{}).get('value'))
PY
bash tools/gpu_profiles.sh r2 > gpurun_out/c43/prof.log 2>&1; tail -6 gpurun_out/c43/prof.log
python tools/kstat.py $(find gpurun_out/prof_r2/bench_stats -name "*kernel_stats.csv" | head -1) gemm_w4 gemm_p8 gemm_bt kv_tiles flash rmsnorm splitk
