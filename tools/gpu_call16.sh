#!/bin/bash
mkdir -p gpurun_out/c16
S="4096,4096,4096,0;5120,12288,4096,0"
for v in 80 101 102 116 117 118; do
  echo "== VAR $v zeros"; GEMM_AB_DATA=zeros VT_W4_VAR=$v timeout 120 tools/bin/gemm_ab "$S" 10,13 0.3 3 2>&1
  echo "== VAR $v random"; VT_W4_VAR=$v timeout 120 tools/bin/gemm_ab "$S" 10,13 0.3 3 2>&1
done > gpurun_out/c16/var.txt
cat gpurun_out/c16/var.txt
