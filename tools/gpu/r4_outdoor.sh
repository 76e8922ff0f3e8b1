#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -6
for n in 1 2 4; do timeout 120 python tools/micro/outdoor_bench.py $n 5 2>&1 | grep outdoor; done
export TMPDIR=/tmp; cd /tmp
timeout -k 5 180 rocprofv3 --kernel-trace --stats -d $O/prof_outdoor -o p -- python $R/tools/micro/outdoor_bench.py 2 5 > /dev/null 2> $O/prof_outdoor.err
cd $R
python tools/rocpd_summary.py $(find $O/prof_outdoor -name '*.db' | head -1) 2>&1 | head -24 | cut -c1-150 | tee $O/r4_kernel_stats_outdoor.txt
rm -rf $O/prof_outdoor
