#!/bin/bash
# rocprofv3 kernel trace of ONE latency leg's timed pairs (between the two profile markers) -> gpurun_out/<tag>_<leg>_kernel_stats.md
# usage: latency_profile.sh <tag> <leg> [pairs]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; leg=$2; n=${3:-10}
rm -rf /tmp/kt_$tag_$leg
rocprofv3 --kernel-trace --stats -d /tmp/kt_${tag}_$leg -- python $R/tools/latency.py --profile --pairs $n --legs $leg $([[ $leg == c_* ]] || echo --no-gnn) > /tmp/l_${tag}_$leg.json 2>/tmp/l_${tag}_$leg.err
tail -c 300 /tmp/l_${tag}_$leg.err
DB=$(find /tmp/kt_${tag}_$leg -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB "tools/latency.py --profile --pairs $n --legs $leg: the timed pairs only (per step = per pair)" --between-markers --steps $n > $R/gpurun_out/${tag}_latency_${leg}_kernel_stats.md
head -45 $R/gpurun_out/${tag}_latency_${leg}_kernel_stats.md | cut -c1-130
cat /tmp/l_${tag}_$leg.json >> $R/gpurun_out/${tag}_latency_profiled.jsonl
