#!/bin/bash
# Round-3 evidence run (on the GPU box, through gpurun): bench line, step-only kernel table, PMC passes.  Writes gpurun_out/r03_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py > $O/r03_z_bench.json 2> $O/r03_z_bench.err
bash $R/tools/step_profile.sh > /dev/null 2>&1
bash $R/tools/pmc_step.sh > /dev/null 2>&1
python - <<EOF
import json
d = json.load(open("$O/r03_z_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["torch_cpu"]["value"])
print([(x["kernel"][:36], round(x["frac"], 3), round(x.get("avg_launch_ms", x.get("ms")), 3)) for x in d["roofline_secondary"]])
print(d["guard_trips"])
EOF
