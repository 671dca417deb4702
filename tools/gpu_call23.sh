#!/bin/bash
mkdir -p gpurun_out/c23
S="4608,4096,1024,1;4616,4096,1024,1;4616,4096,1024,2;4616,4096,1024,0"
timeout 300 tools/bin/gemm_ab "$S" 10,13,14,0 0.3 3 > gpurun_out/c23/ab.jsonl 2>&1
cat gpurun_out/c23/ab.jsonl
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_fullwidth.py -q 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --decode-steps 0 --c4-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['kernel_ms_per_step'])"
