#!/bin/bash
# quick GPU loop: parity tests + GEMM microbench + bench (no cpu baseline, no rocprof)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15 > $O/pytest_gpu.log
python tools/micro/gemm_bench.py > $O/gemm_bench.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err
tail -n 3 $O/pytest_gpu.log; cat $O/gemm_bench.log; tail -n 3 $O/bench_quick.err
