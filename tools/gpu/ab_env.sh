#!/bin/bash
# same-box A/B of an environment switch: bench twice alternating   usage: ab_env.sh VAR=VALUE
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
one() { env $1 python bench.py --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-overlap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={e['kernel']:e for e in d.get('kernels',[])}
print('$1', d['value'], d['ms_per_step'], d.get('stage_ms',{}).get('backbone'), d.get('stage_ms',{}).get('hot_path_hip'), ' '.join('%s=%.1f'%(n.replace('_kernel',''),e['avg_launch_us']) for n,e in k.items()))"; }
for i in 1 2; do
  one _X=0
  one "$1"
done
