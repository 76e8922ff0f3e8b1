#!/bin/bash
# same-box A/B: where the side stream (FPN fine branch) is joined -- before coarse matching (default) or after it (late) -- and the
# serial run; score_conf's in-region time is the bench line's roofline entry
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
one() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1 $2', d['value'], d['ms_per_step'], 'score_conf in-region us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'serial stages', d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'])"; }
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'n/a')"
for i in 1 2; do
  one LOFTR_FINE_JOIN=early ""
  one LOFTR_FINE_JOIN=late ""
  one LOFTR_FINE_JOIN=early "--no-overlap"
done
