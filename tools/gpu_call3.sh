#!/bin/bash
set -x
mkdir -p gpurun_out/c3
export VT_PARITY_REPORT=$PWD/gpurun_out/c3/parity_fullwidth.json
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/c3/pytest.log 2>&1
tail -25 gpurun_out/c3/pytest.log
grep "parity-fullwidth" gpurun_out/c3/pytest.log
