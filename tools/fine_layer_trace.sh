#!/bin/bash
# kernel trace of the fine level's one-kernel layer (4096 x [264, 145]) -> gpurun_out/r05_gnn_fine_kernel_stats.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ktF
rocprofv3 --kernel-trace --stats -d /tmp/ktF -- python $R/tools/fine_layer_check.py --time-only $FINE_ARGS > /tmp/ktF.log 2>&1
tail -1 /tmp/ktF.log
python $R/tools/rocpd_stats.py $(find /tmp/ktF -name "*.db" | head -1) "tools/fine_layer_check.py --time-only: 6 x ops.attentional_propagation at 4096 x [264, 145]" > $R/gpurun_out/r05_gnn_fine_kernel_stats.md 2>&1
head -14 $R/gpurun_out/r05_gnn_fine_kernel_stats.md | cut -c1-150
