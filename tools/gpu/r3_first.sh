#!/bin/bash
# round-3 first GPU pass: full parity suite, fused-encoder A/B (serial streams), kernel-trace stats of the serial run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03a}
O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 > $O/${T}_pytest.log
tail -n 15 $O/${T}_pytest.log
STEPS=10 bash tools/gpu/ab_env.sh LOFTR_FUSED_ENCODER=0 2>&1 | tee $O/${T}_ab_fused.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p -- python $R/bench.py --no-cpu-baseline --warmup 2 --steps 5 --no-overlap > $O/${T}_prof_bench.json 2> $O/${T}_prof.err
cd $R
python tools/rocpd_summary.py $(find $O/prof_$T -name '*.db' | head -1) > $O/${T}_kernel_stats_serial.txt 2>&1
head -24 $O/${T}_kernel_stats_serial.txt
rm -rf $O/prof_$T
python -c "
import json
d=json.load(open('$O/${T}_bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline_encoder'])
"
