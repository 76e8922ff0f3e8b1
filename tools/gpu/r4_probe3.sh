#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
for v in $VARIANTS; do
  for l in "layer1 3x3" "layer2_outconv2.0"; do
    echo "== variant $v"
    LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip_p_$v.so timeout 120 python tools/micro/conv_probe.py "$l" 2>&1 | grep -v "^W2026\|amdgpu.ids" | head -2
  done
done 2>&1 | tee $O/r4_probe3.txt
