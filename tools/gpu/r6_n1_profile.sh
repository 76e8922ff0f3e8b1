# single-pair (BASELINE configs[0]) kernel table + coverage: where a 640x480 pair's ~4.3 ms go
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python -c 'import torch' 2> /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/prof_n1 -o p -- python $R/bench.py --no-cpu-baseline --no-other-configs --batch 1 --steps 20 --warmup 5 > /tmp/n1.json 2> /tmp/n1.err
python $R/tools/rocpd_summary.py $(find /tmp/prof_n1 -name "*.db" | head -1) | cut -c1-170 > $O/r06_kernel_stats_n1.txt
head -45 $O/r06_kernel_stats_n1.txt; tail -3 $O/r06_kernel_stats_n1.txt
python -c "
import json; d=json.loads(open('/tmp/n1.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
