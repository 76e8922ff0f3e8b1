#!/bin/bash
# A/B of library variants on the per-layer convolution table:  VARIANTS="name1 name2" bash tools/gpu/ab_conv.sh [layer filter]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; python -c 'import torch' 2> /dev/null
for rep in 1 2; do
for v in base $VARIANTS; do
  L=""; [ "$v" != base ] && L=$R/loftr_amd/libloftr_hip_$v.so
  echo "== $v"; LOFTR_HIP_LIB=$L timeout 300 python tools/micro/conv_layers.py 16 10 "${1:-3x3 }" 2>&1 | grep -v "^W2026\|amdgpu.ids\|s2\|^layer"
done; done
