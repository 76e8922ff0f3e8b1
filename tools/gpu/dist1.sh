#!/bin/bash
# The multi-rank code path of bench.py on ONE GPU (a gpurun box has one): torchrun launcher, nccl (= RCCL) process group,
# C-ABI communicator + count all-gather, per-rank timing gather -- everything but a second rank.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
LOFTR_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline
