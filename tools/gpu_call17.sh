#!/bin/bash
mkdir -p gpurun_out/c17
S="5120,12288,4096,0;5120,22016,4096,6;4096,4096,11008,4;4096,4096,4096,4;5120,4096,4096,4;5120,4096,11008,4;4616,3072,1024,0;4616,4096,1024,2;4616,1024,4096,4;4608,4096,1024,1;4608,4096,4096,0;1088,12288,4096,0;1088,22016,4096,6;2048,12288,4096,0"
timeout 300 tools/bin/gemm_ab "$S" 10,13 0.4 3 > gpurun_out/c17/ab.jsonl 2>&1
cat gpurun_out/c17/ab.jsonl
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -3
