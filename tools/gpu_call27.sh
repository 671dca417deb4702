#!/bin/bash
S="4616,4096,1024,1;4616,4096,1024,0;4608,4096,4096,0;5120,4096,4096,4;5120,4096,11008,4"
timeout 300 tools/bin/gemm_ab "$S" 10,13,14 0.3 3 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm" 2>&1 | tail -2
