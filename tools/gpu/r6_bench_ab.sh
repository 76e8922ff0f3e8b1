#!/bin/bash
# Round 6: the bench step with the coarse transformer as launches / as the persistent kernel, with and without the side stream
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
for rep in ${REPS:-1 2 3}; do
for mode in launches persistent; do
  for ov in "" "--no-overlap"; do
    tag=${mode}_${ov:5:2}
    LOFTR_COARSE_MODE=$mode timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $ov > $O/ab_${tag}_$rep.json 2> $O/ab.err
    python - <<PY
import json
d=json.load(open('$O/ab_${tag}_$rep.json'))
r=d['roofline']
print('$mode', '$ov' or 'overlap', d['value'], d['ms_per_step'], 'stage', d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'],
      'roofline in-region frac', r['frac'], 'exec', r['executed_frac'], 'ms', r['ms_per_step'], 'alone', r.get('alone', {}).get('frac'), r.get('alone', {}).get('ms_per_step'))
PY
  done
done
done
