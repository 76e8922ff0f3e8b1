#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03d}; O=$R/gpurun_out; mkdir -p $O; cd $R
for M in 7700 1000 3; do
  timeout 120 python tools/micro/fine_bench.py $M 5 2>&1 | grep -v amdgpu.ids
  LOFTR_FUSED_FINE=0 timeout 120 python tools/micro/fine_bench.py $M 5 2>&1 | grep -v amdgpu.ids
done | tee $O/${T}_fine_ab.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 > $O/${T}_pytest.log
tail -n 12 $O/${T}_pytest.log
