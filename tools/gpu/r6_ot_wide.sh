#!/bin/bash
# Round 6: outdoor Sinkhorn passes (2 x 11025^2): rows per round / threads per workgroup of otp::ot_pass_kernel's wide form
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "" _otr2 _otr2b; do
  echo "== libloftr_hip$v.so"
  rm -rf /tmp/prof_ot$v
  LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip$v.so rocprofv3 --kernel-trace --stats -d /tmp/prof_ot$v -o p -- python $R/tools/micro/ot_bench.py 2 10 105 105 2>&1 | grep "sinkhorn coarse"
  python $R/tools/rocpd_summary.py $(find /tmp/prof_ot$v -name "*.db" | head -1) | grep -E "ot_pass|score_sweep|ot_col" | cut -c1-150
done
