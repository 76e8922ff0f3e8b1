#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 900 python -m pytest tests/test_hip_grad.py tests/test_hip_training.py -m gpu -q -x --timeout 600 2>&1 | tail -5
bash tools/gpu/r3_backward_profile.sh r04
