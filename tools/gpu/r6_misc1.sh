#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 300 python tools/micro/backbone_halves.py 10 > $O/r06_backbone_halves.txt 2>&1; grep -v amdgpu.ids $O/r06_backbone_halves.txt | tail -8
timeout 300 python -u tools/micro/pct_check.py 8 4800 4800 --trace --burn 30 > $O/r06_pct_check_8_4800.txt 2>&1; grep -v "amdgpu.ids\|^run " $O/r06_pct_check_8_4800.txt | tail -22
