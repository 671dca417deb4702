#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the bench command, PMC traffic (separate passes), MFMA-utilisation counters of
# the dominant GEMM launches, and a clock / power trace of a sustained GEMM + attention run.   usage: bash tools/gpu_profiles.sh TAG
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --decode-steps 0 --c4-steps 0"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/bench_stats -- $BENCH > $O/bench_under_rocprof.json 2> $O/bench_stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --c4-steps 0 > /dev/null 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --c4-steps 0 > /dev/null 2> $O/pmc_write.err
# the four big decoder GEMMs of the benchmark step as the dispatcher runs them (cfg 0 = AUTO: gate/up on 256-row tiles, qkv / down_proj / o_proj on 320-row tiles)
G="$R/tools/bin/gemm_ab 5120,22016,4096,6;5120,12288,4096,0;5120,4096,11008,4;5120,4096,4096,4 0 0.04 1"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $O/pmc_gemm_a -- $G > /dev/null 2> $O/pmc_gemm_a.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES -f csv -d $O/pmc_gemm_b -- $G > /dev/null 2> $O/pmc_gemm_b.err
# clock / power trace: sampled every 0.2 s while a sustained GEMM window (5 s) runs
( while true; do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power"; sleep 0.2; done ) > $O/smi_trace_gemm.txt &
SMI=$!
$R/tools/bin/gemm_ab "5120,22016,4096,6" 0 5.0 1 > $O/smi_gemm_result.json 2>&1
kill $SMI
cd $R
find $O -name "*kernel_stats.csv" | head -3
python tools/pmc_summarize.py $O/pmc_traffic.json $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.txt 2>&1
python tools/pmc_summarize.py $O/pmc_gemm.json $O/pmc_gemm_a $O/pmc_gemm_b > $O/pmc_gemm.txt 2>&1
tail -8 $O/pmc_gemm.txt; tail -12 $O/smi_trace_gemm.txt; cat $O/smi_gemm_result.json
# keep the merged artefacts small: drop the raw per-dispatch CSVs except the stats
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
du -sh $O
