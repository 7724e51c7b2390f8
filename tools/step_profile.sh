cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt1
rocprofv3 --kernel-trace --stats -d /tmp/kt1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-overlap > /tmp/b.json 2>/tmp/b.err
tail -c 600 /tmp/b.err
DB=$(find /tmp/kt1 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB "bench.py --steps 5 --warmup 2 --no-overlap: the timed steps only" --between-markers --steps 5 > $R/gpurun_out/r03_step_kernel_stats.md
head -24 $R/gpurun_out/r03_step_kernel_stats.md | cut -c1-110; tail -4 $R/gpurun_out/r03_step_kernel_stats.md
