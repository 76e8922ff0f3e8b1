#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_backbone.py -m gpu -q -x -s --timeout 600 2>&1 | tail -40 > $O/pytest_bb.log
tail -n 40 $O/pytest_bb.log
