#!/bin/bash
mkdir -p gpurun_out/c24
GEMM_SUSTAINED_SECONDS=1.0 timeout 600 python tools/gemm_sustained.py all > gpurun_out/c24/sustained.jsonl 2>&1
cat gpurun_out/c24/sustained.jsonl
