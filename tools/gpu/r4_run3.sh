#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
rm -f $O/parity_e2e.txt
timeout 600 python -m pytest tests/test_e2e_golden.py -m gpu -q --timeout 500 2>&1 | tail -5
cut -c1-40,100-400 $O/parity_e2e.txt | grep "=hip" | tail -8
for v in "" prio1 prio2; do
  echo "== lib $v"
  L=""; [ -n "$v" ] && L=$R/loftr_amd/libloftr_hip_$v.so
  LOFTR_HIP_LIB=$L timeout 300 python tools/micro/conv_layers.py 16 10 "@1/2" 2>&1 | grep -v "^W2026\|amdgpu.ids"
done
