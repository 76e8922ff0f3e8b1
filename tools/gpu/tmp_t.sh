cd $GRAFT_REPO_ROOT; python -c 'import torch' 2>/dev/null; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_backbone.py -m gpu -q -x -k "remainder or surviving" 2>&1 | tail -3
for sw in conv_rem=1 conv_rem=0 conv_rem=1 conv_rem=0; do echo "== $sw"; timeout 300 python tools/micro/conv_layers.py 16 10 ">196" $sw 2>&1 | grep "196"; done
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_rem -o p -- python $GRAFT_REPO_ROOT/tools/micro/conv_layers.py 16 5 "196>196" conv_rem=1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py $(find /tmp/prof_rem -name '*.db' | head -1) 2>&1 | head -6 | cut -c1-170
