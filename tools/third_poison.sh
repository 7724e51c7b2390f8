#!/bin/bash
# round 5: does the fp16-split third-level kernel consume leftovers (registers / LDS) of whatever ran before it?  Fresh processes.
cd $GRAFT_REPO_ROOT
export PATS_AMD_DIAG_LIB=1 PATS_THIRD_VARIANT=1350
OUT=gpurun_out/r05_third_poison.log
: > $OUT
for i in 1 2 3 4; do timeout 200 python tools/third_first_launch.py 2>&1 | grep -E "RESULT|launch [0-9]+:" >> $OUT; done
for i in 1 2 3 4; do POISON=0 timeout 200 python tools/third_first_launch.py 2>&1 | grep -E "RESULT|launch [0-9]+:" >> $OUT; done
for i in 1 2 3 4; do POISON=7fc00000 timeout 200 python tools/third_first_launch.py 2>&1 | grep -E "RESULT|launch [0-9]+:" >> $OUT; done
for i in 1 2; do POISON=7f800000 timeout 200 python tools/third_first_launch.py 2>&1 | grep -E "RESULT|launch [0-9]+:" >> $OUT; done
export PATS_THIRD_VARIANT=300
for i in 1 2; do POISON=7fc00000 timeout 200 python tools/third_first_launch.py 2>&1 | grep -E "RESULT|launch [0-9]+:" >> $OUT; done
cat $OUT
