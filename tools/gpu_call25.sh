#!/bin/bash
mkdir -p gpurun_out/c25
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -3
S="5120,4096,4096,4;5120,4096,11008,4;4096,4096,4096,4;5120,4096,4096,0;5120,12288,4096,0;5120,22016,4096,6;4616,4096,1024,1"
timeout 300 tools/bin/gemm_ab "$S" 10,13,14,0 0.3 3 > gpurun_out/c25/ab.jsonl 2>&1
cat gpurun_out/c25/ab.jsonl
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --decode-steps 0 --c4-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['kernel_ms_per_step'])"
