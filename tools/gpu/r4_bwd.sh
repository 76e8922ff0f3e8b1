#!/bin/bash
# backward kernels after tuning: gradient tests, schedule test, training-step timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 1200 python -m pytest tests/test_hip_grad.py tests/test_hip_training.py tests/test_hip_parity.py -m gpu -x -q > $O/bwd_pytest.txt 2>&1; grep -E "passed|failed|error" $O/bwd_pytest.txt | tail -3
timeout 200 python tools/micro/train_step_bench.py 2 3 2>&1 | grep "N="
export TMPDIR=/tmp; cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_ts -o p -- python $R/tools/micro/train_step_bench.py 2 3 > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $(find $O/prof_ts -name '*.db' | head -1) 2>&1 | head -16 | cut -c1-150
rm -rf $O/prof_ts
