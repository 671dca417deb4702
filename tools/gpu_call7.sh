#!/bin/bash
mkdir -p gpurun_out/c7
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "fused_qkv" > gpurun_out/c7/pytest.log 2>&1; grep -E "AssertionError|passed|failed" gpurun_out/c7/pytest.log | head
bash tools/gpu_profiles.sh r2a > gpurun_out/c7/prof.log 2>&1; tail -30 gpurun_out/c7/prof.log
