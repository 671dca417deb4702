#!/bin/bash
# GPU call 1 (round 2): p4x A/B on the decoder shapes, tile-config sweep on the mid-size shapes, correctness of cfg 11, same-box bench
set -x
mkdir -p gpurun_out/c1
B=tools/bin/gemm_ab
timeout 300 $B "5120,12288,4096,0;5120,22016,4096,6;4096,4096,11008,4;4096,4096,4096,4;5120,12288,4096,0" 10,11,8 0.6 3 > gpurun_out/c1/big.jsonl 2> gpurun_out/c1/big.err
timeout 400 $B "4616,1024,1024,4;4616,3072,1024,0;4616,4096,1024,1;4616,1024,4096,4;1024,4096,4096,4;1024,4096,11008,4;4608,4096,1024,1;4608,4096,4096,0" 0,2,3,4,5,10,11 0.25 3 > gpurun_out/c1/mid.jsonl 2> gpurun_out/c1/mid.err
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > gpurun_out/c1/pytest_gemm.log 2>&1
tail -5 gpurun_out/c1/pytest_gemm.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --decode-steps 0 > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err
cat gpurun_out/c1/big.jsonl gpurun_out/c1/mid.jsonl
