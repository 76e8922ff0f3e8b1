#!/bin/bash
# same-box A/B of an environment switch on the score-volume micro-benchmark   usage: score_env_ab.sh VAR=VALUE [VAR=VALUE ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for round in 1 2; do
  echo "== default"; python tools/micro/score_bench.py 8 10 2>/dev/null | head -4
  for v in "$@"; do echo "== $v"; env $v python tools/micro/score_bench.py 8 10 2>/dev/null | head -4; done
done
