#!/bin/bash
timeout 300 python tools/attn_vs_sdpa.py 2>&1 | grep -v amdgpu.ids | tail -5
