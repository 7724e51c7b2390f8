#!/bin/bash
# PMC passes over the bench's steps (on the GPU box, through gpurun): FETCH_SIZE and WRITE_SIZE in their own runs.
# usage: pmc_step.sh <nchw|nhwc>  -> gpurun_out/r04_pmc_step_<layout>.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${1:-nchw}
rm -rf /tmp/pmcF /tmp/pmcW
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python $R/bench.py --maps $L --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /tmp/pmc_bench.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python $R/bench.py --maps $L --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_step.py /tmp/pmcF /tmp/pmcW /tmp/pmc_bench.json > $R/gpurun_out/r04_pmc_step_$L.json
head -c 1200 $R/gpurun_out/r04_pmc_step_$L.json
