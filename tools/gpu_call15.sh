#!/bin/bash
mkdir -p gpurun_out/c15
S="4096,4096,4096,0;5120,12288,4096,0"
for v in 64 65 66 68 72 69 79; do
  echo "== VAR $v zeros"; GEMM_AB_DATA=zeros VT_W4_VAR=$v timeout 120 tools/bin/gemm_ab "$S" 10,13 0.3 3 2>&1
done > gpurun_out/c15/var.txt
for v in 64 65 66 68 79; do
  echo "== VAR $v random"; VT_W4_VAR=$v timeout 120 tools/bin/gemm_ab "$S" 10,13 0.3 3 2>&1
done >> gpurun_out/c15/var.txt
cat gpurun_out/c15/var.txt
