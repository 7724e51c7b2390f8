#!/bin/bash
# A/B of non-temporal operand loads in cost_mfma_kernel inside the bench's steps (PATS_COST_NT = 0 / 1) -> gpurun_out/r04_cost_nt_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
{
for nt in 0 1 0 1; do
  PATS_COST_NT=$nt python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
sec={s['kernel'].split(' ')[0]: s for s in d.get('roofline_secondary',[])}
print('PATS_COST_NT=$nt', 'nchw %.1f' % d['value_nchw'], 'pairs/s |', ' '.join('%s %.3f' % (k, v.get('avg_launch_ms', -1)) for k, v in sec.items() if v.get('avg_launch_ms')))
"
done
} 2>&1 | tee $O/r04_cost_nt_ab.txt
