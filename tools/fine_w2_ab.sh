#!/bin/bash
# A/B of the fine-level Sinkhorn kernels inside the bench's steps: PATS_FINE_W2 = 0 (four waves per problem, sinkhorn_blk.hip) /
# 1 (two waves per problem, sinkhorn_blk2w.hip) -> gpurun_out/r04_fine_w2_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
{
for m in 0 1 0 1; do
  PATS_FINE_W2=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
sec={s['kernel'].split(' ')[0]: s for s in d.get('roofline_secondary',[])}
print('PATS_FINE_W2=$m', 'nchw %.1f' % d['value_nchw'], 'nhwc %.1f' % d['value_nhwc'], 'pairs/s |', ' '.join('%s %.3f' % (k, v.get('avg_launch_ms', -1)) for k, v in sec.items() if v.get('avg_launch_ms')), '| determinism', d.get('step_determinism',{}).get('identical'), '| wild', [round(g['pairs_per_s'],1) for g in d.get('guard_trips',[])])
"
done
} 2>&1 | tee $O/r04_fine_w2_ab.txt
