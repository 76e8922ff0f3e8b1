#!/bin/bash
# Round-6 profile set (same recipe as rounds 4 and 5), all from the library as built: (1) the whole GPU test suite, (2) parity margins (hot-path goldens + image-level goldens with
# the fp64 distances), (3) rocprofv3 kernel-trace stats of the default (two-stream) and the serial bench, (4) three PMC passes -> pmc_traffic.json,
# (5) coverage, (6) outdoor / Sinkhorn / backward kernel stats, per-layer convolution table, (7) the bench line as the driver runs it.
#   bash tools/gpu/r4_profile.sh [tag=r04]  ->  gpurun_out/<tag>_* and gpurun_out/pmc_traffic.json: copy what is to be judged into profiles/ -- pmc_traffic.json ALWAYS
#   (gpurun only merges gpurun_out/ back; the copy into profiles/ below happens on the GPU box, for the bench line of that run)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r06}
O=$R/gpurun_out
mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c 'import torch' 2> /dev/null
rm -f $O/parity_e2e.txt $O/parity_features.txt $O/full_backward_margins.txt $O/full_backward_all.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/${T}_pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/${T}_pytest_full.txt | tail -3 | tee $O/${T}_pytest_gpu.txt
{ echo "# parity margins of the HIP path, library of $(date -u +%FT%TZ), source hash $(python -c 'import bench; print(bench.source_hash())')";
  echo "# (1) hot path from the reference's backbone features (tools/parity_margins.py)"; timeout 600 python tools/parity_margins.py 2>&1 | grep -v "amdgpu.ids\|^W2026";
  echo "# (2) full forward from images, both backbones (tests/test_e2e_golden.py; max and RMS distances to the reference's fp64 forward, next to the reference's own fp32 forward)";
  python tools/margins_table.py $O/parity_e2e.txt; echo "# (2b) the raw lines"; cat $O/parity_e2e.txt; echo "# (3) margin guard on the feature-level case that carries the reference fp64 run (tests/test_hip_parity.py)"; cat $O/parity_features.txt; } > $O/${T}_parity_margins.txt; cp $O/full_backward_margins.txt $O/${T}_full_backward_margins.txt
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs --warmup 2"
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
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/prof_cov_$T -o p -- $B --steps 12 > /dev/null 2> $O/${T}_cov.err
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_out_$T -o p -- python $R/tools/micro/outdoor_bench.py 2 5 2>&1 | grep "^outdoor" > $O/${T}_outdoor_n2.txt
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_ot_$T -o p -- $B --match-type sinkhorn --steps 5 --no-overlap > /dev/null 2> $O/${T}_ot.err
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_bwd_$T -o p -- python $R/tools/micro/grad_bench.py 8 7700 3 > $O/${T}_grad_bench.txt 2> $O/${T}_grad_bench.err
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/prof_ts_$T -o p -- python $R/tools/micro/train_step_bench.py 2 3 2>&1 | grep -v "amdgpu.ids\|^W2026\|^E2026" > $O/${T}_train_step.txt
cd $R
db() { find $1 -name '*.db' | head -1; }
python tools/rocpd_summary.py $(db $O/prof_$T) > $O/${T}_kernel_stats.txt 2>&1
python tools/rocpd_summary.py $(db $O/profs_$T) > $O/${T}_kernel_stats_serial.txt 2>&1
python tools/rocpd_pmc.py $(db $O/pmc_fetch_$T) $(db $O/pmc_write_$T) --sq $(db $O/pmc_sq_$T) --json $O/pmc_traffic.json > $O/${T}_pmc_traffic.txt 2>&1
python tools/rocpd_summary.py $(db $O/prof_cov_$T) | grep "^# " > $O/${T}_coverage.txt
python tools/rocpd_summary.py $(db $O/prof_out_$T) > $O/${T}_kernel_stats_outdoor.txt 2>&1
python tools/rocpd_summary.py $(db $O/prof_ot_$T) > $O/${T}_kernel_stats_ot.txt 2>&1
python tools/rocpd_summary.py $(db $O/prof_bwd_$T) > $O/${T}_kernel_stats_backward.txt 2>&1
python tools/rocpd_summary.py $(db $O/prof_ts_$T) > $O/${T}_kernel_stats_train_step.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
rm -rf $O/prof_$T $O/profs_$T $O/pmc_fetch_$T $O/pmc_write_$T $O/pmc_sq_$T $O/prof_cov_$T $O/prof_out_$T $O/prof_ot_$T $O/prof_bwd_$T $O/prof_ts_$T
for n in 1 2 4; do timeout 120 python tools/micro/outdoor_bench.py $n 5 2>&1 | grep outdoor; done | tee -a $O/${T}_outdoor_n2.txt
for m in 2 4; do timeout 120 python tools/micro/outdoor_bench.py $m 5 sinkhorn 2>&1 | grep outdoor; done | tee -a $O/${T}_outdoor_n2.txt
[ -x tools/micro/dma_probe ] && timeout 60 tools/micro/dma_probe > $O/${T}_dma_probe.txt 2>&1
timeout 300 python -u tools/micro/pct_check.py 8 4800 4800 --trace --burn 30 2>&1 | grep -v 'amdgpu.ids\|^run ' > $O/${T}_pct_check.txt
for sh in '1 4800 4800' '2 11025 11025 mask' '8 4096 4096'; do timeout 300 python -u tools/micro/pct_check.py $sh --trace 2>&1 | grep -v 'amdgpu.ids\|^run ' | tail -9 >> $O/${T}_pct_check.txt; done
timeout 300 python tools/micro/conv_layers.py 16 10 2>&1 | grep -v "amdgpu.ids\|^W2026" > $O/${T}_conv_layers.txt
head -14 $O/${T}_kernel_stats_serial.txt | cut -c1-160; head -20 $O/${T}_pmc_traffic.txt | cut -c1-200; cat $O/${T}_coverage.txt
# the bench line as the driver runs it (with the PMC table of THIS build in place)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err
python - <<PY
import json
d=json.load(open('$O/${T}_bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'])
print('roofline', {k: d['roofline'].get(k) for k in ('achieved','frac','ms_per_step','traffic','mfma_busy','dominant_share_of_step')})
print('score', {k: d['roofline_score_volume'].get(k) for k in ('achieved','frac','traffic','avg_launch_us')})
print('backbone', {k: d['roofline_backbone'].get(k) for k in ('achieved','frac','ms_per_step','traffic','mfma_busy')})
print('other', json.dumps(d.get('other_configs'))[:1200])
print('cpu', json.dumps({k: v for k, v in d.get('cpu_baseline', {}).items() if k not in ('sample',)})[:700])
PY
