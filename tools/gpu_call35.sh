#!/bin/bash
S="2560,4096,4096,4;577,3072,1024,0;577,4096,1024,1;300,4096,11008,4;2048,4096,4096,4;1088,22016,4096,6;8704,4096,4096,4;640,4096,4096,0;5120,1024,4096,4"
timeout 300 tools/bin/gemm_ab "$S" 5,13,15,0 0.3 3 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm or profile_api" 2>&1 | tail -4
