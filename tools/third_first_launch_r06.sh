#!/bin/bash
# Round 6, time-boxed (verdict item 6): the two untested explanations of the fp16-split third-level kernel's first-launch divergence.
#   iters1   one full-size launch of the SAME code object with a single sweep before launch 0: every CU has fetched and executed every
#            code path (instruction caches, scalar caches, L2 lines of the 60 KB kernel) at 1 % of the work
#   small    the full 100-sweep solve on 256 problems before launch 0: code in L2, a couple of CUs have executed it
#   reverse  workgroup b solves problem P - 1 - b: do the affected problems follow the dispatch order or the data?
# One fresh process per line; the diagnostic library must be in the tree (python -m pats_amd.build --diag, built before shipping).
out=gpurun_out/r06_third_first_launch.log
mkdir -p gpurun_out; : > $out
export PATS_AMD_DIAG_LIB=1 PATS_THIRD_VARIANT=1350
run() { echo "=== $*" >> $out; env "$@" timeout 300 python tools/third_first_launch.py 2>&1 | grep -v amdgpu.ids >> $out; }
python -c "import torch" 2>/dev/null
for i in 1 2 3 4 5 6; do run PLAIN=1; done
for i in 1 2 3 4 5 6; do run PREHEAT=iters1:0; done
for i in 1 2 3 4 5 6; do run PREHEAT=small:0; done
for i in 1 2 3 4 5 6; do run PATS_REVERSE_BLOCKS=1; done
grep -c . $out
grep "RESULT\|===" $out | paste - - | cut -c1-220
