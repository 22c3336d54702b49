#!/bin/bash
# rocprofv3 PMC passes over tools/mb_update_x3.py (one counter set per run, kernel trace only): per-kernel counters of the x3 chains
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_x3}; mkdir -p $out
root=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1)); d=/tmp/pmcx3_$i; rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python $root/tools/mb_update_x3.py ${E:-40000} > /tmp/pmcx3_$i.log 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "pass $i ($set): no output"; tail -5 /tmp/pmcx3_$i.log; continue; fi
  echo "# rocprofv3 --pmc $set --kernel-trace -- python tools/mb_update_x3.py ${E:-40000}" > $out/pass_$i.txt
  python tools/pmc_summary.py $f x3_ >> $out/pass_$i.txt 2>&1
done
cat $out/pass_*.txt
