#!/bin/bash
set -x
mkdir -p gpurun_out/c4
B=tools/bin/gemm_ab
timeout 300 $B "5120,12288,4096,0;5120,22016,4096,6;4096,4096,11008,4;4096,4096,4096,4;4616,3072,1024,0" 10,12 0.5 3 > gpurun_out/c4/p2.jsonl 2> gpurun_out/c4/p2.err
cat gpurun_out/c4/p2.jsonl gpurun_out/c4/p2.err
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "two_phase or top_k or top_p" > gpurun_out/c4/pytest_k.log 2>&1; tail -5 gpurun_out/c4/pytest_k.log
timeout 600 python tools/parity_ops_fullwidth.py 1088 > gpurun_out/c4/ops.log 2>&1; cat gpurun_out/c4/ops.log | tail -20
export VT_PARITY_REPORT=$PWD/gpurun_out/c4/parity.json
timeout 900 python -m pytest tests/test_gpu_parity_fulldepth.py -q -s > gpurun_out/c4/fulldepth.log 2>&1; tail -5 gpurun_out/c4/fulldepth.log; grep parity-fulldepth gpurun_out/c4/fulldepth.log
