#!/bin/bash
# same-box A/B: two-stream overlap of the FPN fine branch (default) vs everything on one stream
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
one() { python bench.py --steps ${STEPS:-15} --warmup 3 --no-cpu-baseline $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('${1:-overlap}', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do one ""; one --no-overlap; done
