cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof_n1 -o p -- python $R/bench.py --no-cpu-baseline --no-other-configs --batch 1 --steps 20 --warmup 5 > /tmp/n1.json 2> /tmp/n1.err
python $R/tools/rocpd_summary.py $(find /tmp/prof_n1 -name "*.db" | head -1) | head -40 | cut -c1-170
python -c "
import json; d=json.loads(open('/tmp/n1.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
