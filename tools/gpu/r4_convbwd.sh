#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
rm -f $O/full_backward_margins.txt
timeout 1200 python -m pytest tests/test_hip_grad.py tests/test_hip_training.py -m gpu -x -q > $O/convbwd_pytest2.txt 2>&1; grep -E "passed|failed|Error|assert" $O/convbwd_pytest2.txt | tail -8
cat $O/full_backward_margins.txt | cut -c1-420
timeout 300 python tools/micro/train_step_bench.py 2 3 images 2>&1 | grep "N="
export TMPDIR=/tmp; cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_ts -o p -- python $R/tools/micro/train_step_bench.py 2 3 images > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/prof_ts -name '*.db' | head -1) 2>&1 | head -14 | cut -c1-150
rm -rf $O/prof_ts
