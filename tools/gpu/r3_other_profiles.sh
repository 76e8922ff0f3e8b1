#!/bin/bash
# rocprofv3 kernel-trace summaries of BASELINE configs[3] (outdoor 840x840 masked, N = 4) and configs[4] (indoor_ot) -> gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp LOFTR_BENCH_NO_RETRY=1; cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_out_$T -o p -- python $R/tools/micro/outdoor_bench.py 4 5 > $O/${T}_outdoor_n4.txt 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_ot_$T -o p -- python $R/bench.py --match-type sinkhorn --no-cpu-baseline --no-other-configs --warmup 2 --steps 5 --no-overlap > $O/${T}_ot_bench.json 2> $O/${T}_ot.err
cd $R
python tools/rocpd_summary.py $(find $O/prof_out_$T -name '*.db' | head -1) > $O/${T}_kernel_stats_outdoor.txt 2>&1
python tools/rocpd_summary.py $(find $O/prof_ot_$T -name '*.db' | head -1) > $O/${T}_kernel_stats_ot.txt 2>&1
rm -rf $O/prof_out_$T $O/prof_ot_$T
grep -v amdgpu $O/${T}_outdoor_n4.txt | tail -2; python tools/micro/outdoor_bench.py 2 5 2>&1 | tail -1
head -22 $O/${T}_kernel_stats_outdoor.txt; head -22 $O/${T}_kernel_stats_ot.txt
