#!/bin/bash
# A/B of the gathers' cache policy inside the bench's steps (PATS_GATHER_NT = 0 plain / 1 non-temporal map reads / 2 + non-temporal
# output stores), both map layouts: -> gpurun_out/r04_gather_nt_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
{
for pol in 0 1 2 0 1 2; do
  PATS_GATHER_NT=$pol python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
sec={s['kernel'].split(' ')[0]: s for s in d.get('roofline_secondary',[])}
r=d['roofline']
print('PATS_GATHER_NT=$pol', 'nchw %.1f' % d['value_nchw'], 'nhwc %.1f' % d['value_nhwc'], 'pairs/s |', r['kernel'].split(' ')[0], '%.3f ms' % r['avg_launch_ms'], '|', ' '.join('%s %.3f' % (k, v.get('avg_launch_ms', -1)) for k, v in sec.items()), '| gather_layouts', json.dumps(d.get('gather_layouts',{}).get('fine_desc_ms')), json.dumps(d.get('gather_layouts',{}).get('third_desc_ms')))
"
done
} 2>&1 | tee $O/r04_gather_nt_ab.txt
