#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_backbone.py -m gpu -q -x --timeout 600 2>&1 | tail -3
LOFTR_CONV_DMA=1 timeout 600 python -m pytest tests/test_hip_backbone.py -m gpu -q -x --timeout 600 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_a.json 2>/dev/null
LOFTR_CONV_DMA=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_b.json 2>/dev/null
