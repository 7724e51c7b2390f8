#!/bin/bash
# kernel-overlap profile of one latency leg: usage latency_overlap.sh <leg> [pairs]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; leg=$1; n=${2:-10}
rm -rf /tmp/ko_$leg
rocprofv3 --kernel-trace -d /tmp/ko_$leg -- python $R/tools/latency.py --profile --pairs $n --legs $leg --no-gnn > /dev/null 2>&1
echo "== $leg ($n pairs)"; python $R/tools/trace_overlap.py $(find /tmp/ko_$leg -name "*.db" | head -1)
