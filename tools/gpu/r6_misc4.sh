#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 600 python -m pytest tests/test_e2e_golden.py -m gpu -q -k "schedule_variants" 2>&1 | tail -5
CONFIGS="persistent: persistent:--backbone-halves=1 launches:--backbone-halves=1 launches:" REPS="1 2 3" bash tools/gpu/r6_bench_ab.sh
