#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
PCT_DEBUG=pct_quota=5 timeout 300 python -u tools/micro/pct_check.py 8 4800 4800 > $O/pct_quota5.txt 2>&1; grep -v "amdgpu.ids\|^run " $O/pct_quota5.txt | tail -9
PCT_DEBUG=pct_quota=3 timeout 300 python -u tools/micro/pct_check.py 2 700 500 mask > $O/pct_quota3.txt 2>&1; grep -v "amdgpu.ids\|^run " $O/pct_quota3.txt | tail -5
bash tools/gpu/r6_bench_ab.sh
