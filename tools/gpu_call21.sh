#!/bin/bash
mkdir -p gpurun_out/c21
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c21/bench.json 2> gpurun_out/c21/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c21/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['config'].get('ms_per_step_hipevent_median'), d['config'].get('c4',{}).get('tokens_per_s'), d['config']['kernel_ms_per_step'])
print(d.get('decode',{}).get('ms_per_step'))
PY
tail -3 gpurun_out/c21/bench.err
S="4616,4096,1024,2;4608,4096,1024,1;4616,3072,1024,0;4616,1024,1024,4;4616,1024,4096,4;1088,12288,4096,0;1088,4096,4096,4;2560,4096,4096,4;2560,12288,4096,0"
timeout 300 tools/bin/gemm_ab "$S" 10,13,14,0 0.3 3 > gpurun_out/c21/ab.jsonl 2>&1
cat gpurun_out/c21/ab.jsonl
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -q 2>&1 | tail -5
