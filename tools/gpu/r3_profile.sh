#!/bin/bash
# Round-3 profile set: bench line (driver flags), kernel-trace stats of the default (two-stream) and the serial run, three PMC passes.
#   bash tools/gpu/r3_profile.sh <tag>   -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats[_serial].txt, <tag>_pmc_traffic.txt, pmc_traffic.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03}
O=$R/gpurun_out
mkdir -p $O; export TMPDIR=/tmp LOFTR_BENCH_NO_RETRY=1; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs --warmup 2"
python -c 'import torch' 2> /dev/null     # page the image in before the first timed-out-able run
# (an aborted process under rocprofv3 does not exit by itself: short timeouts, and stop at the first failure)
fail() { echo "FAILED: $1"; grep -v "^W2026\|^E2026\|^I2026" $2 | tail -6 | cut -c1-300; exit 1; }
timeout -k 5 180 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- $B --steps 5 > $O/${T}_prof_bench.json 2> $O/${T}_prof.err
[ $? -eq 0 ] || fail "$O/prof_$T" $O/${T}_prof.err
timeout -k 5 180 rocprofv3 --kernel-trace --stats -d $O/profs_$T -o p -- $B --steps 5 --no-overlap > $O/${T}_prof_bench_serial.json 2> $O/${T}_profs.err
[ $? -eq 0 ] || fail "$O/profs_$T" $O/${T}_profs.err
timeout -k 5 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$T -o p -- $B --steps 2 > /dev/null 2> $O/${T}_pmc_fetch.err
[ $? -eq 0 ] || fail "$O/pmc_fetch_$T" $O/${T}_pmc_fetch.err
timeout -k 5 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$T -o p -- $B --steps 2 > /dev/null 2> $O/${T}_pmc_write.err
[ $? -eq 0 ] || fail "$O/pmc_write_$T" $O/${T}_pmc_write.err
timeout -k 5 180 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq_$T -o p -- $B --steps 2 > /dev/null 2> $O/${T}_pmc_sq.err
[ $? -eq 0 ] || fail "$O/pmc_sq_$T" $O/${T}_pmc_sq.err
cd $R
db() { find $1 -name '*.db' | head -1; }
python tools/rocpd_summary.py $(db $O/prof_$T) > $O/${T}_kernel_stats.txt 2>&1
python tools/rocpd_summary.py $(db $O/profs_$T) > $O/${T}_kernel_stats_serial.txt 2>&1
python tools/rocpd_pmc.py $(db $O/pmc_fetch_$T) $(db $O/pmc_write_$T) --sq $(db $O/pmc_sq_$T) --json $O/pmc_traffic.json > $O/${T}_pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
head -32 $O/${T}_kernel_stats_serial.txt; head -34 $O/${T}_pmc_traffic.txt
rm -rf $O/prof_$T $O/profs_$T $O/pmc_fetch_$T $O/pmc_write_$T $O/pmc_sq_$T
# the bench line as the driver runs it (with the PMC table of THIS build in place)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err
python -c "
import json
d=json.load(open('$O/${T}_bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'])
print('roofline', {k: d['roofline'][k] for k in ('achieved','frac','traffic','avg_launch_us')})
print('encoder', {k: d['roofline_encoder'].get(k) for k in ('achieved','frac','ms_per_step','traffic','mfma_busy')})
print('other', json.dumps(d.get('other_configs'))[:1500])
print('cpu', {k: v for k, v in d.get('cpu_baseline', {}).items() if k not in ('sample', 'reference_forward_note')})
"
