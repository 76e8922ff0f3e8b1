#!/bin/bash
# How sensitive are the training step's gradients to forward differences of the size the HIP backbone has (features ~1e-5 from the
# reference's)?  The CPU-backbone variant of the full-backward test (identical features) rerun with the features perturbed by 1e-5 / 1e-6.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O; cd $R
python -c 'import torch' 2> /dev/null
for eps in 0 1e-6 1e-5; do
  rm -f $O/full_backward_all.txt $O/full_backward_margins.txt
  LOFTR_TEST_PERTURB_FEATURES=$eps timeout 600 python -m pytest tests/test_hip_training.py -m gpu -q -k "full_backward and tfull_ds-cpu" 2>&1 | tail -1
  echo "== features perturbed by $eps (relative): top tensors of tfull_ds, CPU-mirror backbone"
  grep -v "^==" $O/full_backward_all.txt | grep -v backbone | head -8
done 2>&1 | tee $O/r05_backward_sensitivity.txt
