#!/bin/bash
# SQ counter passes + kernel trace over the fused GNN layer (on the GPU box): -> gpurun_out/r04_gnn_pmc.txt, r04_gnn_layer_kernel_stats.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; : > $O/r04_gnn_pmc.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmcG_$tag
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcG_$tag -- python $R/tools/pmc_gnn.py > /dev/null 2>&1
  echo "## fused GNN layer, counters: $set" >> $O/r04_gnn_pmc.txt
  python $R/tools/pmc_sum.py /tmp/pmcG_$tag "gnn" >> $O/r04_gnn_pmc.txt 2>&1
done
rm -rf /tmp/ktG
rocprofv3 --kernel-trace --stats -d /tmp/ktG -- python $R/tools/pmc_gnn.py > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/ktG -name "*.db" | head -1) "tools/pmc_gnn.py: 4 x ops.attentional_propagation(25 920 x [128,65], eval BatchNorm, residual)" > $O/r04_gnn_layer_kernel_stats.md 2>&1
cat $O/r04_gnn_pmc.txt; head -30 $O/r04_gnn_layer_kernel_stats.md
