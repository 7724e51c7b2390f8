#!/bin/bash
# Round 5: kernel trace + SQ / cache counter passes over the fine level's three-kernel layer - a four-layer AttentionalGNN stack on
# 2 x 2048 rows of [264, 145] (tools/fine_layer_check.py --time-only --stack: 4 096 problems a layer) - on the GPU box
#   -> gpurun_out/r05_gnn_fine_kernel_stats.md, gpurun_out/r05_gnn_fine_pmc.txt
# (counter passes in their own runs, --pmc only: never together with a trace domain)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf /tmp/ktF
rocprofv3 --kernel-trace --stats -d /tmp/ktF -- python $R/tools/fine_layer_check.py --time-only --stack > /tmp/ktF.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/ktF -name "*.db" | head -1) "tools/fine_layer_check.py --time-only --stack: 4 x ops.attentional_gnn (4 layers) on 2 x 2048 rows of [264, 145] = 4 096 problems a layer" > $O/r05_gnn_fine_kernel_stats.md 2>&1
: > $O/r05_gnn_fine_pmc.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmcF_$tag
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcF_$tag -- python $R/tools/fine_layer_check.py --time-only --stack > /dev/null 2>&1
  echo "## fine-level stack (4 layers x 4 096 problems of [264, 145]; mean per dispatch), counters: $set" >> $O/r05_gnn_fine_pmc.txt
  python $R/tools/pmc_sum.py /tmp/pmcF_$tag "gnn_fine_tile" >> $O/r05_gnn_fine_pmc.txt 2>&1
  python $R/tools/pmc_sum.py /tmp/pmcF_$tag "gnn_fine_attn" >> $O/r05_gnn_fine_pmc.txt 2>&1
done
head -12 $O/r05_gnn_fine_kernel_stats.md | cut -c1-120
cat $O/r05_gnn_fine_pmc.txt
