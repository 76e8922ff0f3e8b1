#!/bin/bash
# Round 6: the whole GPU suite (margins / gradient reports collected), then the bench line.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
rm -f $O/parity_e2e.txt $O/parity_features.txt $O/full_backward_margins.txt $O/full_backward_all.txt
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_K:+-k "$PYTEST_K"} > $O/r06_pytest_gpu_full.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/r06_pytest_gpu_full.txt | tee $O/r06_pytest_gpu.txt
cp $O/parity_e2e.txt $O/r06_parity_margins.txt 2> /dev/null
cat $O/parity_features.txt >> $O/r06_parity_margins.txt 2> /dev/null
if [ -n "$BENCH" ]; then
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err
  python - <<PY
import json
d=json.load(open('$O/r06_bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms']['backbone'], d['stage_ms']['hot_path_hip'])
print('roofline', {k: d['roofline'].get(k) for k in ('achieved','frac','in_region_frac','ms_per_step','in_region_ms_per_step','dominant_share_of_step')})
print('cpu', d.get('cpu_baseline_kind'), json.dumps(d.get('cpu_baseline', {}).get('parity_vs_reference'))[:900])
print(' '.join('%s=%.1f' % (e['kernel'], e['avg_launch_us']) for e in d.get('kernels', [])))
print('other', json.dumps({k: {kk: v.get(kk) for kk in ('ms_per_step', 'pairs_per_s')} for k, v in d.get('other_configs', {}).items()}))
PY
  tail -3 $O/r06_bench.err
fi
