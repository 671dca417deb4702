#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the bench command (prefill step) and of the C5 decode flow, PMC traffic (separate
# passes), MFMA-utilisation counters of the dominant GEMM launches and of the flash-attention kernel, and a clock / power trace of a
# sustained GEMM window.   usage: bash tools/gpu_profiles.sh TAG [quick]     (quick: kernel stats + flash PMC only)
TAG=${1:-r3}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --decode-steps 0 --c4-steps 0 --c2-reps 0 --reps-224 0 --no-empirical-peaks --fp16-ab-steps 0 --no-parity --no-live-traffic"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/bench_stats -- python $R/bench.py --steps 5 --warmup 2 $LEAN > $O/bench_under_rocprof.json 2> $O/bench_stats.err
# the C5 decode flow (4 x (image + box) prefill + 256 greedy steps at batch 4) and C2 under the same tool
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/decode_stats -- python $R/tools/decode_bench.py 256 > $O/decode_under_rocprof.json 2> $O/decode_stats.err
# the reference-native 224 px shapes (round 5): C3-224 (S = 2560) and C2-224 (S = 768) alone, 3 warm-up + 10 timed steps each
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/c3_224_stats -- python $R/tools/prefill_bench.py clip 224 10 > $O/c3_224_under_rocprof.json 2> $O/c3_224_stats.err
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/c2_224_stats -- python $R/tools/prefill_bench.py image 224 10 > $O/c2_224_under_rocprof.json 2> $O/c2_224_stats.err
# flash attention (S = 5120 causal, 32 heads x 128): SQ counters, two passes of 8
A="python $R/tools/one_attn.py"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES -f csv -d $O/pmc_attn_a -- $A > /dev/null 2> $O/pmc_attn_a.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -f csv -d $O/pmc_attn_b -- $A > /dev/null 2> $O/pmc_attn_b.err
cd $R
python tools/pmc_summarize.py $O/pmc_flash_attn.json $O/pmc_attn_a $O/pmc_attn_b > $O/pmc_flash_attn.txt 2>&1
tail -6 $O/pmc_flash_attn.txt
if [ -z "$QUICK" ]; then
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 $LEAN > /dev/null 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 $LEAN > /dev/null 2> $O/pmc_write.err
# the four big decoder GEMMs of the benchmark step as the dispatcher runs them (cfg 0 = AUTO)
G="$R/tools/bin/gemm_ab 5120,22016,4096,6;5120,12288,4096,0;5120,4096,11008,4;5120,4096,4096,4 0 0.04 1"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $O/pmc_gemm_a -- $G > /dev/null 2> $O/pmc_gemm_a.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES -f csv -d $O/pmc_gemm_b -- $G > /dev/null 2> $O/pmc_gemm_b.err
# clock / power trace: sampled every 0.2 s while a sustained GEMM window (5 s) runs
( while true; do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power"; sleep 0.2; done ) > $O/smi_trace_gemm.txt &
SMI=$!
$R/tools/bin/gemm_ab "5120,22016,4096,6" 0 5.0 1 > $O/smi_gemm_result.json 2>&1
kill $SMI
cd $R
python tools/pmc_summarize.py $O/pmc_traffic.json $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.txt 2>&1
python tools/pmc_summarize.py $O/pmc_gemm.json $O/pmc_gemm_a $O/pmc_gemm_b > $O/pmc_gemm.txt 2>&1
tail -8 $O/pmc_gemm.txt; tail -12 $O/smi_trace_gemm.txt; cat $O/smi_gemm_result.json
fi
cd $R
find $O -name "*kernel_stats.csv" | head -4
# keep the merged artefacts small: drop the raw per-dispatch CSVs except the stats
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
du -sh $O
