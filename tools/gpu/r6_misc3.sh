#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 300 python tools/micro/pct_in_model.py > $O/r06_pct_in_model.txt 2>&1; grep -v amdgpu.ids $O/r06_pct_in_model.txt | tail -30
timeout 300 python -u tools/micro/pct_check.py 8 4800 4800 --flush 2>&1 | grep "^time"
