#!/bin/bash
# GPU call 2 (round 2): p4x fixed A/B, full GPU test suite incl. the new parity / acceptance tests, full bench line
set -x
mkdir -p gpurun_out/c2
B=tools/bin/gemm_ab
timeout 300 $B "5120,12288,4096,0;5120,22016,4096,6;4096,4096,11008,4;4096,4096,4096,4;4616,3072,1024,0;4616,4096,1024,1" 10,11,8 0.5 3 > gpurun_out/c2/big.jsonl 2> gpurun_out/c2/big.err
cat gpurun_out/c2/big.jsonl
export VT_PARITY_REPORT=$PWD/gpurun_out/c2/parity_fullwidth.json
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/c2/pytest.log 2>&1
tail -15 gpurun_out/c2/pytest.log
grep "parity-fullwidth" gpurun_out/c2/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c2/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['config'].get('ms_per_step_hipevent_median'), d['config'].get('c4'), d['config']['kernel_ms_per_step'])
print(d.get('decode',{}).get('ms_per_step'), d.get('cpu_baseline',{}).get('value'))
PY
