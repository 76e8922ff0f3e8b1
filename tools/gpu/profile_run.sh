#!/bin/bash
# Round profile set: kernel-trace stats + three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ MFMA busy) of bench.py.
#   bash tools/gpu/profile_run.sh <tag>      -> gpurun_out/<tag>_kernel_stats.txt, <tag>_pmc_traffic.txt, pmc_traffic.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r02}
O=$R/gpurun_out
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --no-cpu-baseline --warmup 2"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- $B --steps 5 > $O/${T}_prof_bench.json 2> $O/${T}_prof.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$T -o p -- $B --steps 2 > /dev/null 2> $O/${T}_pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$T -o p -- $B --steps 2 > /dev/null 2> $O/${T}_pmc_write.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq_$T -o p -- $B --steps 2 > /dev/null 2> $O/${T}_pmc_sq.err
cd $R
db() { find $1 -name '*.db' | head -1; }
python tools/rocpd_summary.py $(db $O/prof_$T) > $O/${T}_kernel_stats.txt 2>&1
python tools/rocpd_pmc.py $(db $O/pmc_fetch_$T) $(db $O/pmc_write_$T) --sq $(db $O/pmc_sq_$T) --json $O/pmc_traffic.json > $O/${T}_pmc_traffic.txt 2>&1
head -30 $O/${T}_kernel_stats.txt; head -40 $O/${T}_pmc_traffic.txt
rm -rf $O/prof_$T $O/pmc_fetch_$T $O/pmc_write_$T $O/pmc_sq_$T
