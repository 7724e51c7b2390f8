#!/bin/bash
# FETCH_SIZE per access pattern -> gpurun_out/r04_fetch_patterns.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_patterns.hip -o /tmp/fetch_patterns 2>/dev/null || exit 1
rm -rf /tmp/pmcP
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcP -- /tmp/fetch_patterns > $R/gpurun_out/r04_fetch_patterns.txt 2>/dev/null
python $R/tools/pmc_sum.py /tmp/pmcP "pat_" >> $R/gpurun_out/r04_fetch_patterns.txt
rm -rf /tmp/ktP
rocprofv3 --kernel-trace --stats -d /tmp/ktP -- /tmp/fetch_patterns > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/ktP -name "*.db" | head -1) "fetch_patterns: durations" | grep pat_ >> $R/gpurun_out/r04_fetch_patterns.txt
cat $R/gpurun_out/r04_fetch_patterns.txt
