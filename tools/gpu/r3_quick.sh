#!/bin/bash
# quick: encoder + fine micro-benchmarks, parity suite, one bench line (overlap on) and one serial
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03q}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/micro/encoder_bench.py 8 5 2>&1 | grep -v amdgpu.ids | tee $O/${T}_micro.txt
python tools/micro/fine_bench.py 7700 5 2>&1 | grep -v amdgpu.ids | head -1 | tee -a $O/${T}_micro.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -12 > $O/${T}_pytest.log
grep -h "passed\|failed" $O/${T}_pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap > $O/${T}_bench_serial.json 2> $O/${T}_bench_serial.err
python - <<PY
import json
for f in ("$O/${T}_bench.json", "$O/${T}_bench_serial.json"):
    d = json.load(open(f))
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'], 'enc', d['roofline_encoder']['ms_per_step'], d['roofline_encoder']['frac'])
    print('  ', ' '.join('%s=%.2f' % (k['kernel'].replace('_kernel', ''), k['ms_per_step']) for k in d['kernels']))
PY
