#!/bin/bash
# Round 6: coarse transformer form (persistent work queue / per-call launches) inside the full forward, by batch size; indoor 640x480 through
# bench.py, outdoor 840x840 through tools/micro/outdoor_bench.py.  Sets ops.COARSE_AUTO_MIN_TILES by measurement.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
for b in 1 2 4 8 16; do
  for m in persistent launches; do
    python bench.py --no-cpu-baseline --no-other-configs --batch $b --steps 20 --warmup 5 --coarse-mode $m 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('indoor batch $b $m: %.3f ms per step, %.1f pairs/s' % (d['ms_per_step'], d['value']))"
  done
done
done
