#!/bin/bash
R=$PWD; O=$R/gpurun_out/c44; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -k "attn or flash or llama or prefill or decode or generate" 2>&1 | tail -3
timeout 120 python tools/attn_bench.py 2>&1 | grep '"S"' | head -4
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --decode-steps 0 --c4-steps 0 > $O/bench.json 2> $O/err.txt
cd $R
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['kernel_ms_per_step'])
PY
python tools/kstat.py $(find $O/run -name "*kernel_stats.csv" | head -1) gemm_w4_kernel\<6 kv_tiles flash
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
