#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null

rm -f $O/full_backward_margins.txt
timeout 1200 python -m pytest tests/test_hip_grad.py tests/test_hip_training.py -m gpu -x -q > $O/convbwd_pytest2.txt 2>&1; grep -E "passed|failed|Error|assert" $O/convbwd_pytest2.txt | tail -8
cat $O/full_backward_margins.txt

timeout 300 python tools/micro/train_step_bench.py 2 3 images 2>&1 | grep "N="
