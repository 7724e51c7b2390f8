#!/bin/bash
# rocprofv3 kernel trace of the bench's timed steps only (between the two profile markers) -> gpurun_out/<tag>_step_kernel_stats.md
# usage: step_profile.sh <tag> [extra bench.py flags, e.g. --maps nhwc or --wild 0.1]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" > /tmp/b_$tag.json 2>/tmp/b_$tag.err
tail -c 300 /tmp/b_$tag.err
DB=$(find /tmp/kt_$tag -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB "bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $*: the timed steps only" --between-markers --steps 5 > $R/gpurun_out/${tag}_step_kernel_stats.md
head -30 $R/gpurun_out/${tag}_step_kernel_stats.md | cut -c1-120
