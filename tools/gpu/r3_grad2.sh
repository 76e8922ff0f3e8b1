#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03g2}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_grad.py tests/test_hip_training.py -q --timeout 300 2>&1 | tail -30 > $O/${T}_grad.log
tail -25 $O/${T}_grad.log
timeout 300 python tools/micro/grad_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/${T}_grad_bench.txt
