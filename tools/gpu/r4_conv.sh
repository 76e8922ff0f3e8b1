#!/bin/bash
# conv3x3_duo A/B: backbone tests, per-layer timings with the round-3 kernels (LOFTR_CONV_DUO=0) and the duo kernels, bench A/B.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 900 python -m pytest tests/test_hip_backbone.py tests/test_backbone_golden.py -m gpu -q -x --timeout 600 2>&1 | tail -8
for d in 0 3; do
  echo "== LOFTR_CONV_DUO=$d"
  LOFTR_CONV_DUO=$d timeout 300 python tools/micro/conv_layers.py 16 10 2>&1 | tee $O/r4_conv_layers_duo$d.txt
done
one() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={e['kernel']:e for e in d.get('kernels',[])}
print('$1', d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'], ' '.join('%s=%.1f'%(n.replace('_kernel',''),e['avg_launch_us']) for n,e in k.items() if 'conv' in n or 'fine_pair' in n))"; }
for i in 1 2; do
  LOFTR_CONV_DUO=0 one duo0
  LOFTR_CONV_DUO=3 one duo3
done
timeout 600 python -m pytest tests/test_e2e_golden.py tests/test_hip_fine_fused.py -m gpu -q -k "hip or fine" --timeout 500 2>&1 | tail -3
cut -c1-40,100-260 $O/parity_e2e.txt | tail -8
