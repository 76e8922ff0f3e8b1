#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_e2e_golden.py tests/test_hip_scale_sweep.py tests/test_ties.py -m gpu -q -x --timeout 600 2>&1 | tail -4
for n in 1 2 4; do timeout 120 python tools/micro/outdoor_bench.py $n 5 2>&1 | grep outdoor; done
bash tools/gpu/r4_outdoor2.sh 2 12 2>&1 | grep -A1 "score_sweep\|encoder_x"
