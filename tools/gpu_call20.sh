#!/bin/bash
mkdir -p gpurun_out/c20
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "four_wave" 2>&1 | tail -5
S="5120,12288,4096,0;5120,4096,4096,4;5120,4096,11008,4;5120,22016,4096,6;4608,4096,4096,0;4608,4096,1024,0;2560,12288,4096,0;2560,4096,4096,4"
timeout 300 tools/bin/gemm_ab "$S" 10,13,14,0 0.4 3 > gpurun_out/c20/ab.jsonl 2>&1
cat gpurun_out/c20/ab.jsonl
