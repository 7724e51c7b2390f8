#!/bin/bash
# PMC passes over one bench step (on the GPU box, through gpurun): FETCH_SIZE and WRITE_SIZE in their own runs.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcF /tmp/pmcW
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_step.py /tmp/pmcF /tmp/pmcW 0 > $R/gpurun_out/r03_pmc_step.json
head -c 1500 $R/gpurun_out/r03_pmc_step.json
