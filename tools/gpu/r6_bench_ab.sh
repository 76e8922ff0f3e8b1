#!/bin/bash
# Round 6: the bench step with the coarse transformer as launches / as the persistent kernel (optionally yielding after a quota of items),
# with and without the side stream
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
CONFIGS=${CONFIGS:-"launches: persistent: persistent:--debug-switch=pct_quota=4 persistent:--debug-switch=pct_quota=12 launches:--no-overlap persistent:--no-overlap"}
for rep in ${REPS:-1 2}; do
for c in $CONFIGS; do
    mode=${c%%:*}; extra=${c#*:}
    tag=$(echo "${mode}_${extra}" | tr -c 'a-zA-Z0-9_\n' '_')
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --coarse-mode $mode $extra > $O/ab_${tag}_$rep.json 2> $O/ab.err
    python - <<PY
import json
d=json.load(open('$O/ab_${tag}_$rep.json'))
r=d['roofline']
print('%-12s %-32s' % ('$mode', '$extra' or 'overlap'), d['value'], d['ms_per_step'], 'stage', d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'],
      '| roofline in-region frac', r['frac'], 'exec', r['executed_frac'], 'ms', r['ms_per_step'], 'alone', r.get('alone', {}).get('frac'), r.get('alone', {}).get('ms_per_step'))
PY
done
done
