#!/bin/bash
# round 5: the fp16-split third-level kernel with the s_nop fence on BOTH sides of its twelve MFMAs (libpats_amd_diag_prefence.so)
cd $GRAFT_REPO_ROOT
export PATS_AMD_DIAG_LIB=_prefence PATS_THIRD_VARIANT=1350
OUT=gpurun_out/r05_third_prefence.log
: > $OUT
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 200 python tools/third_first_launch.py 2>&1 | grep -E "RESULT|launch [0-9]+:" >> $OUT; done
cat $OUT
