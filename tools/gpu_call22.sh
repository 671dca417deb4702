#!/bin/bash
mkdir -p gpurun_out/c22
bash tools/gpu_profiles.sh r2b > gpurun_out/c22/prof.log 2>&1; tail -30 gpurun_out/c22/prof.log
python tools/kstat.py $(find gpurun_out/prof_r2b/bench_stats -name "*kernel_stats.csv" | head -1) gemm_w4 gemm_p8 gemm_bt kv_tiles flash rmsnorm splitk layernorm
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "profile_api" 2>&1 | tail -2
