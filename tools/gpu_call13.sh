#!/bin/bash
mkdir -p gpurun_out/c13
S="4096,4096,4096,0;5120,12288,4096,0"
for v in 0 32; do
  echo "== VAR $v random"; VT_W4_VAR=$v timeout 120 tools/bin/gemm_ab "$S" 10,13 0.4 3 2>&1
  echo "== VAR $v zeros"; GEMM_AB_DATA=zeros VT_W4_VAR=$v timeout 120 tools/bin/gemm_ab "$S" 10,13 0.4 3 2>&1
done > gpurun_out/c13/var.txt
cat gpurun_out/c13/var.txt
VT_W4_VAR=32 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "four_wave" 2>&1 | tail -3
