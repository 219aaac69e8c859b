# Developer script: the measurements kept under profiles/ (run on the GPU box from the repo root).
#   bash tools/final_measure.sh r03
# Companion scripts (separate calls, each a few GPU-minutes): tools/launches_measure.sh (per-pair tables with DCVC_T32 on / off),
# tools/pmc_g32.sh (counters of the 32-pixel tail), tools/bench_stats.sh (per-kernel stats of a bench run), tools/t32_check.sh /
# h128_check.sh / conv_mt_check.sh (A/B timings of the round-3 kernels), tools/dec_profile.py (host profile of the decoder).
set -e
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python3 bench.py > $O/${R}_bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_steps20.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roof -- python3 bench.py --roofline-only > $O/${R}_roofline_under_rocprof.json 2>> $O/bench.err
cp $(find $O/prof_roof -name "*kernel_stats.csv" | head -1) $O/${R}_roofline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-exact-mode > $O/${R}_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/${R}_bench_kernel_stats.csv
rm -rf $O/prof_roof $O/prof_bench
echo "bench + rocprof stats done"
python3 bench.py --frame 3840x2160 --no-cpu-baseline --no-exact-mode > $O/${R}_bench_4k.json 2>> $O/bench.err
python3 tools/iframe_time.py > $O/${R}_iframe_time.txt 2>> $O/bench.err
python3 tools/kbench.py 2>/dev/null > $O/${R}_kbench.txt
DCVC_T128=0 python3 tools/kbench.py 256 136 240 2>/dev/null | sed 's/^/[64-pixel tails, DCVC_T128=0] /' >> $O/${R}_kbench.txt
DCVC_T128=0 python3 tools/kbench.py 320 136 240 2>/dev/null | sed 's/^/[64-pixel tails, DCVC_T128=0] /' >> $O/${R}_kbench.txt
DCVC_T128=0 python3 tools/kbench.py 384 136 240 2>/dev/null | sed 's/^/[64-pixel tails, DCVC_T128=0] /' >> $O/${R}_kbench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl4 -- python3 tools/count_launches.py 4 > /dev/null 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl12 -- python3 tools/count_launches.py 12 > /dev/null 2>> $O/bench.err
python3 tools/count_launches.py --diff $O/cl4 $O/cl12 8 > $O/${R}_launches_per_pair.txt
rm -rf $O/cl4 $O/cl12
echo "4k / iframe / kbench / launches done"
# hardware counters of the dominant kernel: separate passes, --pmc only (MI355X_MICROARCH.md, rocprofv3 PMC slots)
for set in "FETCH_SIZE" "WRITE_SIZE" "MfmaUtil VALUBusy" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TCC_BUSY_avr TCP_TOTAL_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  d=$O/pmc_$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 tools/kbench.py 256 136 240 > /dev/null 2>> $O/pmc.err || echo "pmc pass failed: $set"
done
python3 tools/pmc_summarize.py $O $O/${R}_pmc_dcb_tail.json $O/${R}_pmc_counters.txt
find $O -maxdepth 1 -type d -name "pmc_*" -exec rm -rf {} +
echo done
