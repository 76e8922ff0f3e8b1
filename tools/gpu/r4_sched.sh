#!/bin/bash
# scheduled coarse transformer (two-job launches): bit-identity against the call-by-call order, tests, timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
rm -f $O/enc_*.pt
for cfg in "8 4800 4800" "3 300 300" "2 700 500" "2 11025 11025 mask" "1 4800 4800" "16 1200 1200"; do
  LOFTR_ENCODER_SCHEDULE=0 timeout 120 python tools/micro/encoder_ab.py callwise $cfg 2>&1 | tail -1
  timeout 120 python tools/micro/encoder_ab.py scheduled $cfg 2>&1 | tail -1
done
rm -f $O/enc_*.pt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
