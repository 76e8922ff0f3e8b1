#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_backbone.py -m gpu -q -x --timeout 600 2>&1 | tail -15
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_quick.json 2>/dev/null
