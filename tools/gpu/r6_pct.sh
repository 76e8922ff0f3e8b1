#!/bin/bash
# Round 6: the persistent coarse transformer against the per-call launches (tools/micro/pct_check.py) on the shapes of the bit-identity tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
SHAPES=${SHAPES:-"3_300_300 2_700_500_mask 8_4800_4800 1_4800_4800 2_11025_11025 8_4800_4800_mask"}
for tag in $SHAPES; do
  shape=$(echo $tag | tr '_' ' ')
  PCT_VERBOSE=1 timeout -k 5 ${TMO:-150} python -u tools/micro/pct_check.py $shape --trace --reps ${REPS:-3} > $O/pct_$tag.txt 2>&1
  rc=$?
  echo "== $shape: rc $rc"; grep -v "amdgpu.ids\|^W2026\|^E2026" $O/pct_$tag.txt | tail -${TAIL:-14}
  [ $rc -ne 0 ] && [ -n "$STOP_ON_FAIL" ] && break
done
