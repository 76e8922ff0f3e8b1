#!/bin/bash
# same-box A/B of the score-volume kernels: product library vs the named variants (loftr_amd/libloftr_hip_<v>.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for round in 1 2; do
  echo "== product"; python tools/micro/score_bench.py 8 10 2>/dev/null | head -4
  for v in "$@"; do
    echo "== $v"; LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip_$v.so python tools/micro/score_bench.py 8 10 2>/dev/null | head -4
  done
done
