#!/bin/bash
mkdir -p gpurun_out/c12
S="4096,4096,4096,0;5120,12288,4096,0"
for v in 0 1 2 3 4 8 16 20 28; do
  echo "== VAR $v random"; VT_W4_VAR=$v timeout 120 tools/bin/gemm_ab "$S" 10,13 0.4 3 2>&1
done > gpurun_out/c12/var_random.txt
for v in 0 1 3 4 16 28; do
  echo "== VAR $v zeros"; GEMM_AB_DATA=zeros VT_W4_VAR=$v timeout 120 tools/bin/gemm_ab "$S" 10,13 0.4 3 2>&1
done > gpurun_out/c12/var_zeros.txt
cat gpurun_out/c12/var_random.txt gpurun_out/c12/var_zeros.txt
VT_W4_VAR=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "four_wave" 2>&1 | tail -3
