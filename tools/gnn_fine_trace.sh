#!/bin/bash
# kernel trace of one AttentionalPropagation at the FINE level's shape (4096 x [264, 145]) -> gpurun_out/r04_gnn_fine_kernel_stats.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ktF
B=4096 C=264 NTOK=145 N=3 rocprofv3 --kernel-trace --stats -d /tmp/ktF -- python $R/tools/pmc_gnn.py > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/ktF -name "*.db" | head -1) "tools/pmc_gnn.py B=4096 C=264 NTOK=145: 3 x ops.attentional_propagation at the fine level's shape" --list conv_pk > $R/gpurun_out/r04_gnn_fine_kernel_stats.md 2>&1
head -12 $R/gpurun_out/r04_gnn_fine_kernel_stats.md | cut -c1-120; grep '^- ' $R/gpurun_out/r04_gnn_fine_kernel_stats.md | head -8
