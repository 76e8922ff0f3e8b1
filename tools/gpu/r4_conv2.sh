#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 900 python -m pytest tests/test_hip_backbone.py tests/test_backbone_golden.py -m gpu -q -x --timeout 600 2>&1 | tail -4
for l in "layer1 3x3" "layer2_outconv2.0" "layer1_outconv2.0"; do
  LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip_p_probe.so timeout 120 python tools/micro/conv_probe.py "$l" 2>&1 | grep -v "^W2026\|amdgpu.ids" | head -2
done
timeout 300 python tools/micro/conv_layers.py 16 10 3x3 2>&1 | grep -v "^W2026\|amdgpu.ids" | tee $O/r4_conv_layers_b.txt
one() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k={e['kernel']:e for e in d.get('kernels',[])}
print('$1', d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'], ' '.join('%s=%.1f'%(n.replace('_kernel',''),e['avg_launch_us']) for n,e in k.items() if 'conv' in n or 'fine_pair' in n))"; }
LOFTR_CONV_DUO=0 one duo0
one duo3
one duo3
rm -f $O/parity_e2e.txt
timeout 600 python -m pytest tests/test_e2e_golden.py -m gpu -q -k "hip" --timeout 500 2>&1 | tail -3
cut -c1-40,100-260 $O/parity_e2e.txt | tail -8
