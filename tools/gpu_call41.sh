#!/bin/bash
R=$PWD; O=$R/gpurun_out/c41; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --decode-steps 0 --c4-steps 0 > $O/bench.json 2> $O/err.txt
cd $R
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['kernel_ms_per_step'])
PY
python tools/kstat.py $(find $O/run -name "*kernel_stats.csv" | head -1) gemm_w4 kv_tiles flash rmsnorm
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
