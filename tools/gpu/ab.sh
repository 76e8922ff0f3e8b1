#!/bin/bash
# A/B in one box: two bench configurations back to back, twice
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline $AB_A > $O/bench_a$i.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline $AB_B > $O/bench_b$i.json 2>/dev/null
done
