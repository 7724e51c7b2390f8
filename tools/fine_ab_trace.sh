#!/bin/bash
# kernel-level comparison on the 4-layer fine-level stack (2 x 2048 rows): one rocprofv3 kernel trace per argument,
#   <lib>[:<extra args of tools/fine_layer_check.py>]   lib = a PATS_AMD_DIAG_LIB value (0 = the production library)
# e.g.  bash tools/fine_ab_trace.sh 0 _base 0:--zeros
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for spec in "$@"; do
  v=${spec%%:*}; extra=""; [[ "$spec" == *:* ]] && extra=${spec#*:}
  i=$((i+1)); d=/tmp/kt_$i; rm -rf $d
  PATS_AMD_DIAG_LIB=$v rocprofv3 --kernel-trace --stats -d $d -- python $R/tools/fine_layer_check.py --time-only --stack $extra > $d.log 2>&1
  echo "=== lib $v $extra: $(grep stack_ms $d.log)"
  python $R/tools/rocpd_stats.py $(find $d -name "*.db" | head -1) "stack, lib $v $extra" 2>&1 | grep "gnn_fine" | grep -v pack | cut -c1-100
done
