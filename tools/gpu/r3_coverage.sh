#!/bin/bash
# GPU coverage / launch-gap evidence: rocprofv3 kernel trace of the default (two-stream) bench, 12 steps -> gpurun_out/<tag>_coverage.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp LOFTR_BENCH_NO_RETRY=1; cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/prof_cov_$T -o p -- python $R/bench.py --no-cpu-baseline --no-other-configs --warmup 2 --steps 12 > /dev/null 2> $O/${T}_cov.err
cd $R
python tools/rocpd_summary.py $(find $O/prof_cov_$T -name '*.db' | head -1) | grep "^# " > $O/${T}_coverage.txt
rm -rf $O/prof_cov_$T
cat $O/${T}_coverage.txt
