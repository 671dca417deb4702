#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "ring_kernel" 2>&1 | tail -4
S="4616,1024,4096,4;4616,1024,1024,4;1088,4096,4096,4;1088,4096,11008,4;577,1024,4096,4;4616,3072,1024,0;4616,4096,1024,1"
timeout 300 tools/bin/gemm_ab "$S" 5,15,0 0.3 3 2>&1
