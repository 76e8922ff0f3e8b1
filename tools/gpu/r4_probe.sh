#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
for st in 0 1.0; do
  for l in "layer1 3x3" "layer2_outconv2.0" "layer1_outconv2.0"; do
    echo "== stagger $st"
    LOFTR_CONV_STAGGER=$st LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip_probe.so timeout 120 python tools/micro/conv_probe.py "$l" 2>&1 | grep -v "^W2026\|amdgpu.ids"
  done
done 2>&1 | tee $O/r4_probe.txt
for st in 0 0.5 1.0 1.5; do
  echo "== LOFTR_CONV_STAGGER=$st"
  LOFTR_CONV_STAGGER=$st timeout 300 python tools/micro/conv_layers.py 16 10 3x3 2>&1 | grep -v "^W2026\|amdgpu.ids\|s2"
done 2>&1 | tee $O/r4_stagger.txt
