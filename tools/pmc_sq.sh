#!/bin/bash
# two SQ counter passes over the bench's steps -> gpurun_out/r04_pmc_sq_<layout>.md      usage: pmc_sq.sh <nchw|nhwc>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=${1:-nchw}
rm -rf /tmp/sq1 /tmp/sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d /tmp/sq1 -- python $R/bench.py --maps $L --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d /tmp/sq2 -- python $R/bench.py --maps $L --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_sq.py /tmp/sq1 /tmp/sq2 "SQ counters of the bench's kernels (bench.py --maps $L --steps 3 --warmup 1, two rocprofv3 --pmc passes, round 4)" > $R/gpurun_out/r04_pmc_sq_$L.md
cat $R/gpurun_out/r04_pmc_sq_$L.md
