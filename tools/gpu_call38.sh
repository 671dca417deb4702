#!/bin/bash
R=$PWD; O=$R/gpurun_out/c38; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/run$i -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --decode-steps 0 --c4-steps 0 > $O/bench$i.json 2> $O/err$i.txt
  python - <<PY
import json,csv,glob
d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('run $i', d['ms_per_step'], d['config']['kernel_ms_per_step']['gemm_tile'])
f=glob.glob('$O/run$i/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'gemm_w4_kernel<1, 10' in n or 'gemm_w4_kernel<0, 8' in n or 'gemm_w4r' in n or 'gemm_w4_kernel<6' in n:
        print('   ', n[28:62], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1), round(float(r['MaxNs'])/1e3,1))
PY
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
