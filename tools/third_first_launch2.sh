#!/bin/bash
# second experiment matrix (memory-translation theory, stagger, longer idle); output appended to gpurun_out/third_first_launch2.log
out=gpurun_out/third_first_launch2.log
mkdir -p gpurun_out; : > $out
run() { echo "=== $*" >> $out; env "$@" timeout 400 python tools/third_first_launch.py 2>&1 | grep -v amdgpu.ids >> $out; }
python -c "import torch" 2>/dev/null
for i in 1 2 3 4; do run PREHEAT=touch:2; done
for i in 1 2 3 4; do run PREHEAT=other:0.6; done
for i in 1 2 3 4; do run PREHEAT=same1:0; done
for i in 1 2 3; do run PATS_STAGGER=0; done
for i in 1 2 3; do run PATS_STAGGER=5; done
for i in 1 2; do run IDLE=100; done
for i in 1 2 3 4 5 6; do run PATS_THIRD_VARIANT=300; done
grep RESULT $out
