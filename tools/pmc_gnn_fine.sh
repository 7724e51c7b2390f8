#!/bin/bash
# SQ counter passes over one AttentionalPropagation at the FINE level's shape (4096 x [264, 145]): conv_pk_kernel and
# attention145_kernel (on the GPU box) -> gpurun_out/r04_gnn_fine_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; : > $O/r04_gnn_fine_pmc.txt
export B=4096 C=264 NTOK=145 N=2
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmcF_$tag
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcF_$tag -- python $R/tools/pmc_gnn.py > /dev/null 2>&1
  echo "## fine-level GNN layer (4096 x [264, 145]), counters: $set" >> $O/r04_gnn_fine_pmc.txt
  python $R/tools/pmc_sum.py /tmp/pmcF_$tag "conv_pk" >> $O/r04_gnn_fine_pmc.txt 2>&1
  python $R/tools/pmc_sum.py /tmp/pmcF_$tag "attention145" >> $O/r04_gnn_fine_pmc.txt 2>&1
done
cat $O/r04_gnn_fine_pmc.txt
