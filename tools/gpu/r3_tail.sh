#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03g}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do
  python tools/micro/encoder_bench.py 8 5
  LOFTR_ENCODER_TAIL=0 python tools/micro/encoder_bench.py 8 5
done 2>&1 | grep -v amdgpu.ids | tee $O/${T}_tail_ab.txt
python tools/micro/encoder_bench.py 2 5 11025 2>&1 | grep -v amdgpu.ids | tee -a $O/${T}_tail_ab.txt
LOFTR_ENCODER_TAIL=0 python tools/micro/encoder_bench.py 2 5 11025 2>&1 | grep -v amdgpu.ids | tee -a $O/${T}_tail_ab.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "^$" | tail -12 > $O/${T}_pytest.log
grep -h "passed\|failed\|FAILED" $O/${T}_pytest.log
