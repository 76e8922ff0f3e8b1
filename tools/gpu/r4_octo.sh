#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
for d in 3 12 6 9; do
  echo "== LOFTR_CONV_DUO=$d"
  LOFTR_CONV_DUO=$d timeout 600 python -m pytest tests/test_hip_backbone.py -m gpu -q -x --timeout 600 -k "conv" 2>&1 | tail -2
  LOFTR_CONV_DUO=$d timeout 300 python tools/micro/conv_layers.py 16 10 "3x3 " 2>&1 | grep -v "^W2026\|amdgpu.ids\|s2"
done 2>&1 | tee $O/r4_octo.txt
