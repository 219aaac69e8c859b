#!/bin/bash
# developer script: hardware counters of the 32-pixel ring tail at C=384, 68x120 (separate --pmc passes, no trace options)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc32
rm -rf $O && mkdir -p $O
for set in "MfmaUtil VALUBusy" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TA_BUSY_avr TCC_BUSY_avr TCP_TOTAL_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/pmc_$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 tools/kbench.py 384 68 120 > /dev/null 2>> $O/pmc.err || echo "pmc pass failed: $set"
done
python3 tools/pmc_summarize.py $O $O/g32_pmc.json $O/g32_pmc_counters.txt
sed -i 's/256 136 240/384 68 120 (dcb_tail128_kernel<384, G32>; "tail128" below = that kernel)/' $O/g32_pmc_counters.txt
find $O -maxdepth 1 -type d -name "pmc_*" -exec rm -rf {} +
cat $O/g32_pmc_counters.txt
