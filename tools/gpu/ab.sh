#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip_lndma.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 600 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_a$i.json 2>/dev/null
LOFTR_HIP_LIB=$R/loftr_amd/libloftr_hip_lndma.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b$i.json 2>/dev/null
done
