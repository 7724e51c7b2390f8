#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)_$m
  rm -rf /tmp/pmcW2_$tag
  PATS_FINE_W2=$m rocprofv3 --pmc $set --output-format csv -d /tmp/pmcW2_$tag -- python $R/tools/_tmp/w2_pmc.py > /dev/null 2>&1
  echo "## PATS_FINE_W2=$m counters: $set"
  python $R/tools/pmc_sum.py /tmp/pmcW2_$tag "sinkhorn_blk145"
done
done
