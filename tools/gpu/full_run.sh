#!/bin/bash
# round-1 GPU session: parity tests, smoke, bench, rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -n 40 > $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r01 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r01 -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r01 -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
ls -R $O/prof | head -n 30
tail -n 3 $O/pytest_gpu.log $O/smoke.log
cat $O/bench.json
tail -n 5 $O/bench.err
