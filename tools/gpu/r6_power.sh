#!/bin/bash
# Round 6: is the bench step power limited?  Socket power / shader clock sampled while the timed loop runs (rocm-smi, amd-smi if present)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
( for i in $(seq 1 60); do echo "t=$i"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|junction\|edge" ; sleep 0.25; done ) > $O/r06_power_samples.txt 2>&1 &
SP=$!
sleep 2
timeout 200 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r06_power_bench.json 2> /dev/null
wait $SP
python - <<PY
import json,re
d=json.load(open('$O/r06_power_bench.json')); print('bench', d['value'], d['ms_per_step'])
t=open('$O/r06_power_samples.txt').read()
print(t[:1500])
pw=[float(x) for x in re.findall(r'Power \(W\):\s*([0-9.]+)', t)] or [float(x) for x in re.findall(r'([0-9.]+)\s*W', t)]
ck=[int(x) for x in re.findall(r'sclk clock level: \d+: \((\d+)Mhz\)', t)]
print('power samples', len(pw), 'max', max(pw) if pw else None, 'median', sorted(pw)[len(pw)//2] if pw else None)
print('sclk samples', len(ck), sorted(set(ck)))
PY
rocm-smi --showmaxpower 2>/dev/null | grep -i power
