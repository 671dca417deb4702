#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --decode-steps 0 --c4-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernel_ms_per_step'])"
