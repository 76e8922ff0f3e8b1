#!/bin/bash
# head-gradient parity tests + Sinkhorn tests (assign fill kernel) + the OT config's step time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03g}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_grad.py -q --timeout 300 2>&1 | tail -30 > $O/${T}_grad.log
tail -25 $O/${T}_grad.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "ot or sinkhorn or training or ties" 2>&1 | tail -8 > $O/${T}_ot.log
cat $O/${T}_ot.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
python - <<PY
import json
d = json.load(open("$O/${T}_bench.json"))
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('other_configs'))[:1500])
PY
