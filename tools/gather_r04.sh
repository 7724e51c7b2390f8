#!/bin/bash
# Round-4 evidence run (on the GPU box, through gpurun): bench line, step-only kernel tables in both map layouts, the other
# workloads, the fused GNN layer's trace and counters, the fine-level layer's trace and counters, the 10 %-wild step.  Writes gpurun_out/r04_*; the PMC / SQ passes have their own scripts
# (tools/pmc_step.sh, tools/pmc_sq.sh, tools/fetch_patterns.sh).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py > $O/r04_z_bench.json 2> $O/r04_z_bench.err
tail -c 300 $O/r04_z_bench.err
bash $R/tools/step_profile.sh r04_nchw --maps nchw > /dev/null 2>&1
bash $R/tools/step_profile.sh r04_nhwc --maps nhwc > /dev/null 2>&1
: > $O/r04_workloads.jsonl
for wlk in scannet yfcc; do
  python $R/bench.py --workload $wlk --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 >> $O/r04_workloads.jsonl
done
bash $R/tools/pmc_gnn.sh > /dev/null 2>&1
BN=train bash -c "cd /tmp; rm -rf /tmp/ktGt; BN=train rocprofv3 --kernel-trace --stats -d /tmp/ktGt -- python $R/tools/pmc_gnn.py > /dev/null 2>&1; python $R/tools/rocpd_stats.py \$(find /tmp/ktGt -name '*.db' | head -1) 'tools/pmc_gnn.py BN=train: 4 x ops.attentional_propagation(25 920 x [128,65], BatchNorm on batch statistics, residual)' > $O/r04_gnn_layer_train_kernel_stats.md 2>&1"
python $R/tools/bench_gnn.py > $O/r04_gnn_layer.jsonl 2>/dev/null
bash $R/tools/gnn_fine_trace.sh > /dev/null 2>&1
bash $R/tools/pmc_gnn_fine.sh > /dev/null 2>&1
bash $R/tools/step_profile.sh r04_wild10 --wild 0.1 > /dev/null 2>&1
python - <<PY
import json
d = json.load(open("$O/r04_z_bench.json"))
print({k: d[k] for k in ("value", "value_nchw", "value_nhwc", "ms_per_step")})
print(d["roofline"]["kernel"][:60], round(d["roofline"]["frac"], 3), round(d["roofline"]["avg_launch_ms"], 3))
for r in d["roofline_secondary"]:
    print("  ", r["kernel"][:80], round(r.get("frac", 0), 3), round(r.get("avg_launch_ms", r.get("ms", r.get("ms_per_launch", 0))), 3), r.get("traffic"))
print(d["gnn"]["ms_per_step"], d["gnn"]["pairs_per_s_with_gnn"])
print(d["guard_trips"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["torch_cpu"]["value"], d["cpu_baseline"]["parity_sample"]["matches_pair0"])
print(d["step_determinism"]["identical"], d["gather_layouts"])
PY
cat $O/r04_workloads.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['config']['workload'][:40], d['value'], d['ms_per_step'])"
head -12 $O/r04_gnn_layer_train_kernel_stats.md | cut -c1-110
