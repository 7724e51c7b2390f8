#!/bin/bash
# Phase timeline of conv_pk_kernel's workgroups at the fine level's shape (4 096 x [264, 145]): the DIAGNOSTIC library
# (python -m pats_amd.build --diag; conv_pk.hip under -DPATS_DIAG carries s_memrealtime stamps) -> gpurun_out/r04_gnn_fine_timeline.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
{
  echo "conv_pk_kernel, one AttentionalPropagation at 4 096 x [264, 145] (tools/conv_pk_timeline.sh: libpats_amd_diag.so, PATS_PK_TL=1): s_memrealtime"
  echo "stamps (100 MHz) of thread 0 of every workgroup at its phase boundaries, mean per workgroup in us; two workgroups per CU.  The products"
  echo "with a channel-blocked output (message, hidden tensor) take the other epilogue and are not stamped."
  echo
  PATS_AMD_DIAG_LIB=1 PATS_PK_TL=1 B=4096 C=264 NTOK=145 N=1 python $R/tools/pmc_gnn.py 2>&1 | grep "conv_pk timeline"
} > $O/r04_gnn_fine_timeline.txt
cat $O/r04_gnn_fine_timeline.txt
