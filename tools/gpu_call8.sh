#!/bin/bash
mkdir -p gpurun_out/c8
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "fused_qkv" > gpurun_out/c8/pytest.log 2>&1; grep -E "K mismatch|passed|failed" gpurun_out/c8/pytest.log | cut -c1-1500 | head
