#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
for dbg in "pct_skip=7" "pct_skip=7,pct_grid=1" "pct_skip=5" "pct_skip=1" "pct_skip=6" "pct_grid=1" "pct_grid=4" ""; do
  PCT_DEBUG=$dbg PCT_ONLY=persistent_call_order PCT_WATCHDOG=45 timeout -k 5 60 python -u tools/micro/pct_check.py 3 300 300 2>&1 | grep -v "amdgpu.ids" | tail -3 | sed "s/^/[$dbg] /"
done
