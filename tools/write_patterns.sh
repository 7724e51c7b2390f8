#!/bin/bash
# WRITE_SIZE per store pattern -> gpurun_out/r06_write_patterns.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 $R/tools/write_patterns.hip -o /tmp/write_patterns 2>/dev/null || exit 1
rm -rf /tmp/pmcW
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- /tmp/write_patterns > $R/gpurun_out/r06_write_patterns.txt 2>/dev/null
python $R/tools/pmc_sum.py /tmp/pmcW "wpat_" >> $R/gpurun_out/r06_write_patterns.txt
rm -rf /tmp/ktW
rocprofv3 --kernel-trace --stats -d /tmp/ktW -- /tmp/write_patterns > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/ktW -name "*.db" | head -1) "write_patterns: durations" | grep wpat_ >> $R/gpurun_out/r06_write_patterns.txt
cat $R/gpurun_out/r06_write_patterns.txt
