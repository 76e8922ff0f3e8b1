#!/bin/bash
# A/B of one environment switch in one box: `tools/gpu/ab_env.sh VAR A B` runs bench with VAR=A and VAR=B twice, back to back
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
one() { env $1=$2 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-overlap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={e['kernel']:e for e in d.get('kernels',[])}
print('$1=$2', d['value'], d['ms_per_step'], d.get('stage_ms',{}).get('backbone'), d.get('stage_ms',{}).get('hot_path_hip'), ' '.join('%s=%.1f'%(n.replace('_kernel',''),e['avg_launch_us']) for n,e in list(k.items())[:6]))"; }
for i in 1 2; do one $1 $2; one $1 $3; done
