#!/bin/bash
# Sinkhorn row-streaming passes generalised to unaligned / wide rows: parity, then outdoor timing for two variants
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_grad.py tests/test_hip_training.py -m gpu -x -q -k "sinkhorn or ot or golden" 2>&1 | tail -5
for lib in ""; do
  export LOFTR_HIP_LIB=${lib:+$R/loftr_amd/libloftr_hip_$lib.so}; [ -z "$lib" ] && unset LOFTR_HIP_LIB
  echo "== lib ${lib:-default}"
  timeout 120 python tools/micro/outdoor_bench.py 2 5 sinkhorn 2>&1 | grep outdoor
  timeout 120 python tools/micro/outdoor_bench.py 4 5 sinkhorn 2>&1 | grep outdoor
  export TMPDIR=/tmp; cd /tmp
  timeout -k 5 180 rocprofv3 --kernel-trace --stats -d $O/prof_oot -o p -- python $R/tools/micro/outdoor_bench.py 2 3 sinkhorn > /dev/null 2> $O/prof_oot.err
  cd $R
  python tools/rocpd_summary.py $(find $O/prof_oot -name '*.db' | head -1) 2>&1 | head -12 | cut -c1-170
  rm -rf $O/prof_oot
done
