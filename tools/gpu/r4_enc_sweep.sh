#!/bin/bash
# encoder_x_kernel duration against workgroups per launch: is a call's time set by whole rounds of 256 workgroups or by the work in it?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
export TMPDIR=/tmp
for cfg in "8 2048" "8 3072" "8 4096" "8 4800" "8 5120" "8 6144" "8 8192"; do
  set -- $cfg
  cd /tmp
  timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $O/prof_es -o p -- python $R/tools/micro/encoder_bench.py $1 3 $2 > $O/es.out 2> $O/es.err
  cd $R
  echo "N=$1 L=$2  cross WGs $(( $1 * $2 / 128 ))  self WGs $(( 2 * $1 * $2 / 128 ))  $(grep -o 'transformer [0-9.]* ms/call' $O/es.out)"
  python tools/rocpd_summary.py $(find $O/prof_es -name '*.db' | head -1) 2>&1 | grep -E "encoder_x|proj_kv|kv_finalize" | cut -c1-120
  rm -rf $O/prof_es
done
