#!/bin/bash
mkdir -p gpurun_out/c19
for r in 1 2; do
for m in w4 p4; do
  if [ $m = p4 ]; then export VT_GEMM_FORCE_P4=1; else unset VT_GEMM_FORCE_P4; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --decode-steps 0 --c4-steps 0 > gpurun_out/c19/bench_${m}_$r.json 2> gpurun_out/c19/bench_${m}_$r.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/c19/bench_${m}_$r.json').read().strip().splitlines()[-1])
print('$m $r', d['value'], d['ms_per_step'], d['config']['kernel_ms_per_step'])
PY
done; done
