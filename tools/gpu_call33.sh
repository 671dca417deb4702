#!/bin/bash
S="5120,22016,4096,6;5120,12288,4096,0;5120,4096,11008,4"
for r in 1 4; do echo "== ROTATE $r"; GEMM_AB_ROTATE=$r timeout 300 tools/bin/gemm_ab "$S" 10,13,14 0.4 3 2>&1; done
