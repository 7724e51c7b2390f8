#!/bin/bash
# Final evidence run of a round (on the GPU box, through gpurun): bench line, kernel traces, counter passes over the kernels
# that changed since tools/gather_profiles.sh last ran.  Writes gpurun_out/r02f_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py > $O/r02f_bench.json 2> $O/r02f_bench.err
rocprofv3 --kernel-trace --stats -d /tmp/kt1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt1 -name "*.db" | head -1) "bench.py --steps 5 --warmup 2 (48 pairs per step; the table includes the set-up kernels that build the synthetic workload)" > $O/r02f_bench_kernel_stats.md 2>&1
python $R/tools/bench_config5.py > $O/r02f_config5.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python $R/tools/bench_config5.py > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt2 -name "*.db" | head -1) "tools/bench_config5.py (config 5: 4096^2 x 448 cost GEMM, 4097^2 Sinkhorn 200 sweeps)" > $O/r02f_config5_kernel_stats.md 2>&1
rm -f $O/r02f_pmc_raw.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcO_$tag -- python $R/tools/pmc_others.py > /dev/null 2>&1
  echo "## other kernels, counters: $set" >> $O/r02f_pmc_raw.txt
  python $R/tools/pmc_sum.py /tmp/pmcO_$tag "" >> $O/r02f_pmc_raw.txt
done
python $R/tools/bench_pipeline.py > $O/r02f_pipeline_latency.json 2>/dev/null
