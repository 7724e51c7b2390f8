#!/bin/bash
# kernel trace of whole steps WITH the heads inside (bench.py --with-gnn) -> gpurun_out/r06_gnn_step_kernel_stats.md + the JSON report
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ktG
rocprofv3 --kernel-trace --stats -d /tmp/ktG -- python $R/bench.py --with-gnn --steps 2 --warmup 1 > /tmp/ktG.log 2>&1
tail -1 /tmp/ktG.log > $R/gpurun_out/r06_gnn_step.json
python $R/tools/rocpd_stats.py $(find /tmp/ktG -name "*.db" | head -1) "bench.py --with-gnn --steps 2: the headline step with the heads inside (GnnNets)" --between-markers --steps 2 > $R/gpurun_out/r06_gnn_step_kernel_stats.md 2>&1
head -30 $R/gpurun_out/r06_gnn_step_kernel_stats.md | cut -c1-150; cat $R/gpurun_out/r06_gnn_step.json | cut -c1-1500
