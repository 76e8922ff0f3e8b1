#!/bin/bash
# Round-4 first GPU call: (1) px / conf margins of the image-level goldens with the fused fine / encoder kernels on and off
# (which kernel spends the margin), (2) per-layer convolution timings, (3) the bench line with the reference's own CPU forward.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
run_margins() {   # tag, env...
  rm -f $O/parity_e2e.txt
  env "${@:2}" timeout 600 python -m pytest tests/test_e2e_golden.py -m gpu -q -k "hip" --timeout 500 2>&1 | tail -3
  mv $O/parity_e2e.txt $O/r4_margins_$1.txt 2> /dev/null
  echo "== $1"; cut -c1-40,100-260 $O/r4_margins_$1.txt
}
run_margins default X=1
run_margins nofusedfine LOFTR_FUSED_FINE=0
run_margins nofusedenc LOFTR_FUSED_ENCODER=0
timeout 300 python tools/micro/conv_layers.py 16 10 2>&1 | tee $O/r4_conv_layers.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r4a_bench.json 2> $O/r4a_bench.err
python - <<PY
import json
d=json.load(open('$O/r4a_bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms'])
print('roofline', {k: d['roofline'].get(k) for k in ('achieved','frac','ms_per_step','dominant_share_of_step','share_of_serial_step')})
print('backbone', {k: d['roofline_backbone'].get(k) for k in ('achieved','frac','ms_per_step','share_of_serial_step')})
print('cpu', json.dumps({k: v for k, v in d.get('cpu_baseline', {}).items() if k not in ('sample',)})[:1500])
PY
tail -3 $O/r4a_bench.err
