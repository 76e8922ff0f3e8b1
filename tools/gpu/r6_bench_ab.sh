#!/bin/bash
# Round 6: the bench step with the coarse transformer as launches / as the persistent kernel, with and without the side stream
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
for rep in 1 2; do
for mode in launches persistent; do
  for ov in "" "--no-overlap"; do
    LOFTR_COARSE_MODE=$mode timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $ov > $O/ab_${mode}_${ov:2:2}_$rep.json 2> $O/ab.err
    python - <<PY
import json
d=json.load(open('$O/ab_${mode}_${ov:2:2}_$rep.json'))
print('$mode', '$ov' or 'overlap', d['value'], d['ms_per_step'], 'stage', {k: round(v, 2) for k, v in d['stage_ms'].items()} if isinstance(d.get('stage_ms'), dict) else '')
PY
  done
done
done
