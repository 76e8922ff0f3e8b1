#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
export TMPDIR=/tmp; cd /tmp
timeout -k 5 180 rocprofv3 --kernel-trace --stats -d $O/prof_outdoor -o p -- python $R/tools/micro/outdoor_bench.py ${1:-2} 3 > /dev/null 2> $O/prof_outdoor.err
cd $R
DB=$(find $O/prof_outdoor -name '*.db' | head -1)
for k in encoder_x proj_kv kv_finalize score_sweep conv3x3_duo "conv_kernel"; do echo "$k:"; python tools/rocpd_calls.py $DB "$k" ${2:-24}; done
rm -rf $O/prof_outdoor
