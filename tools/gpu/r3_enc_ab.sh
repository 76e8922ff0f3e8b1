#!/bin/bash
# round-3: same-box A/B of encoder_x_kernel variants (tools/micro/encoder_bench.py), then the parity suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-r03b}
O=$R/gpurun_out
mkdir -p $O; cd $R
shift
for rep in 1 2; do
  python tools/micro/encoder_bench.py 8 5
  for v in "$@"; do LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip_$v.so python tools/micro/encoder_bench.py 8 5; done
  LOFTR_FUSED_ENCODER=0 python tools/micro/encoder_bench.py 8 5
done 2>&1 | grep -v "^$" | tee $O/${T}_enc_ab.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -30 > $O/${T}_pytest.log
tail -n 8 $O/${T}_pytest.log
