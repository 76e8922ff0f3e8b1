#!/bin/bash
# which box is this, and does the first HIP backbone forward survive on it?  (two unexplained crashes at the first GPU work of a call)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-box}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
  hostname; date
  rocm-smi --showuniqueid --showproductname --showmemuse 2>/dev/null | grep -i "unique\|card series\|gfx\|VRAM%" | head -6
  rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -i partition | head -4
  python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(p.name, p.gcnArchName, p.multi_processor_count, p.total_memory // 2**30, "GiB", torch.cuda.mem_get_info(0))
PY
  timeout -k 5 300 python -X faulthandler -m pytest tests/test_backbone_golden.py tests/test_e2e_golden.py -m gpu -q -x --timeout 200 2>&1 | tail -40
} > $O/${T}.txt 2>&1
head -12 $O/${T}.txt | cut -c1-200; tail -3 $O/${T}.txt | cut -c1-200
