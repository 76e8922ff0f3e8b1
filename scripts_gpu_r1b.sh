#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
./tools/micro/f16_denorm > $O/f16_denorm.log 2>&1
timeout 600 python tools/micro/backbone_variants.py > $O/backbone_variants.log 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
cat $O/f16_denorm.log $O/backbone_variants.log
ls -la $O/pmc_fetch $O/pmc_write
