#!/bin/bash
# developer script: counters of the 32-pixel tail on the round's final build + the vector / matrix co-execution counters of the dominant kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_g32.sh > /dev/null 2>&1
O=gpurun_out/pmc_coexec
rm -rf $O && mkdir -p $O
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  d=$O/pmc_$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 tools/kbench.py 256 136 240 > /dev/null 2>> $O/pmc.err || echo "pmc pass failed: $set" >> $O/failed.txt
done
python3 tools/pmc_summarize.py $O $O/x.json $O/coexec_counters.txt > /dev/null 2>&1
find $O -maxdepth 1 -type d -name "pmc_*" -exec rm -rf {} +
cat gpurun_out/pmc32/g32_pmc_counters.txt; cat $O/coexec_counters.txt; cat $O/failed.txt 2>/dev/null
