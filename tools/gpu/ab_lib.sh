#!/bin/bash
# A/B of two library builds in one box: product .so vs loftr_amd/libloftr_hip_$1.so, bench back to back twice
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
V=$R/loftr_amd/libloftr_hip_$1.so
one() { python bench.py --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-overlap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={e['kernel']:e for e in d.get('kernels',[])}
print('$2', d['value'], d['ms_per_step'], d.get('stage_ms',{}).get('backbone'), d.get('stage_ms',{}).get('hot_path_hip'), ' '.join('%s=%.1f'%(n.replace('_kernel',''),e['avg_launch_us']) for n,e in k.items()))"; }
for i in 1 2; do
  one x new
  LOFTR_HIP_LIB=$V one x $1
done
if [ -n "$GEMM" ]; then
  echo new; python tools/micro/gemm_bench.py
  echo $1; LOFTR_HIP_LIB=$V python tools/micro/gemm_bench.py
fi
