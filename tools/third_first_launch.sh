#!/bin/bash
# Experiment matrix for tools/third_first_launch.py (one fresh process per line); output -> gpurun_out/third_first_launch.log
# (round 4 ran it in two halves: profiles/r04_third_first_launch_1.log / _2.log)
out=gpurun_out/third_first_launch.log
mkdir -p gpurun_out; : > $out
# (the fp16-split build lives in the diagnostic library since round 4)
python -m pats_amd.build --diag > /dev/null 2>&1
export PATS_AMD_DIAG_LIB=1 PATS_THIRD_VARIANT=1350
run() { echo "=== $*" >> $out; env "$@" timeout 300 python tools/third_first_launch.py 2>&1 | grep -v amdgpu.ids >> $out; }
python -c "import torch" 2>/dev/null
for i in 1 2 3 4; do run SMI=1; done
for i in 1 2 3; do run SYNC_FIRST=1 HOST_QUIET=1; done
for i in 1 2; do run IDLE=30 SMI=1; done
for k in vec copy matmul same; do for i in 1 2 3; do run PREHEAT=$k:0.6; done; done
for i in 1 2; do run MATMUL_CHECK=1; done
for i in 1 2 3; do run PATS_THIRD_VARIANT=300; done
for i in 1 2 3; do run P=110136 L=6; done
echo "=== rocm-smi --setperflevel high" >> $out
rocm-smi --setperflevel high >> $out 2>&1
for i in 1 2 3 4; do run SMI=1 PERF=high; done
rocm-smi --setperflevel auto >> $out 2>&1
# second matrix (memory-translation theory, one warm-up launch, stagger, longer idle)
for i in 1 2 3 4; do run PREHEAT=touch:2; done
for i in 1 2 3 4; do run PREHEAT=other:0.6; done
for i in 1 2 3 4; do run PREHEAT=same1:0; done
for i in 1 2 3; do run PATS_STAGGER=0; done
for i in 1 2 3; do run PATS_STAGGER=5; done
for i in 1 2; do run IDLE=100; done
for i in 1 2 3 4 5 6; do run PATS_THIRD_VARIANT=300; done
grep RESULT $out
tail -5 $out
# third matrix: does it matter whether launch 0's OUTPUTS land in memory this process has written before?  (no: 2 of 6)
for i in 1 2 3 4 5 6; do run PREHEAT=outbufs:1; done
