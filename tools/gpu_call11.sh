#!/bin/bash
mkdir -p gpurun_out/c11
timeout 300 tools/bin/gemm_ab "5120,12288,4096,0;5120,22016,4096,6;4096,4096,11008,4;4096,4096,4096,4;4616,3072,1024,0" 10,13 0.5 3 > gpurun_out/c11/w4.jsonl 2> gpurun_out/c11/w4.err
cat gpurun_out/c11/w4.jsonl gpurun_out/c11/w4.err
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "four_wave" 2>&1 | tail -5
