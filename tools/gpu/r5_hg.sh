#!/bin/bash
# Round 5, item 1: head_grad_kernel co-residency fault -- diagnostic build + the two-per-CU build, then this box's baseline bench.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
for v in "$@"; do
  echo "=== variant $v"
  LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip_$v.so LOFTR_WGRAD_CHUNK=128 timeout 120 python tools/micro/hg_diag.py ${REPS:-4} 2>&1 | tail -12
done
echo "=== product"
LOFTR_WGRAD_CHUNK=128 timeout 120 python tools/micro/hg_diag.py 2 2>&1 | tail -5
if [ -n "$BENCH" ]; then
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r5_base_bench.json 2> $O/r5_base_bench.err
  python - <<PY
import json
d=json.load(open('$O/r5_base_bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms'])
print(' '.join('%s=%.1f' % (e['kernel'].replace('_kernel',''), e['avg_launch_us']) for e in d.get('kernels', [])))
PY
fi
