#!/bin/bash
# rocprofv3 kernel-trace summary of the heads' backward at the bench's sizes -> gpurun_out/<tag>_kernel_stats_backward.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
python -c 'import torch' 2> /dev/null
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_bwd_$T -o p -- python $R/tools/micro/grad_bench.py 8 7700 3 > $O/${T}_grad_bench.txt 2> $O/${T}_grad_bench.err
echo "rc=$?"; grep -v "amdgpu.ids\|^W2026" $O/${T}_grad_bench.txt | tail -2
cd $R
python tools/rocpd_summary.py $(find $O/prof_bwd_$T -name '*.db' | head -1) > $O/${T}_kernel_stats_backward.txt 2>&1
rm -rf $O/prof_bwd_$T
head -24 $O/${T}_kernel_stats_backward.txt | cut -c1-200
