#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 900 python bench.py --steps 20 --warmup 5 $BENCH_ARGS > $O/r4_bench.json 2> $O/r4_bench.err
python - <<PY
import json
d=json.load(open('$O/r4_bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'])
print('roofline', {k: d['roofline'].get(k) for k in ('achieved','frac','ms_per_step','dominant_share_of_step','share_of_serial_step')})
print('score', {k: d['roofline_score_volume'].get(k) for k in ('achieved','frac','avg_launch_us')})
print('backbone', {k: d['roofline_backbone'].get(k) for k in ('achieved','frac','ms_per_step','share_of_serial_step')})
for k in d['kernels']: print('  %-22s %8.1f us x %5.1f = %6.3f ms  frac %.3f' % (k['kernel'], k['avg_launch_us'], k['launches_per_step'], k['ms_per_step'], k['frac']))
print('other', json.dumps(d.get('other_configs'))[:1800])
print('cpu', json.dumps({k: v for k, v in d.get('cpu_baseline', {}).items() if k not in ('sample',)})[:900])
PY
tail -2 $O/r4_bench.err
