"""BASELINE configs[3]: MegaDepth-style outdoor pairs (840x840, padded to 840x560 + masks + scales) through the
full forward on MI355X: sanity (finite outputs, matches inside the valid region) and timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loftr_amd import LoFTR, get_cfg
from loftr_amd.synth import make_weights
torch.manual_seed(0)
cfg = get_cfg(thr=0.0)
m = LoFTR(cfg).eval()
m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}, strict=False)
m = m.cuda()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator().manual_seed(3)
img0 = torch.rand(N, 1, 840, 840, generator=g); img1 = torch.roll(img0, (8, 16), (2, 3))
img0[:, :, 560:] = 0; img1[:, :, 560:] = 0
mask = torch.zeros(N, 105, 105, dtype=torch.bool); mask[:, :70] = True
def run():
    d = {"image0": img0.cuda(), "image1": img1.cuda(), "mask0": mask.cuda(), "mask1": mask.cuda(),
         "scale0": torch.full((N, 2), 1.9).cuda(), "scale1": torch.full((N, 2), 1.9).cuda()}
    m(d); torch.cuda.synchronize(); return d
d = run(); d = run()
t = time.perf_counter(); reps = 5
for _ in range(reps): d = run()
dt = (time.perf_counter() - t) / reps
M = d["mconf"].shape[0]
assert torch.isfinite(d["mkpts1_f"]).all() and torch.isfinite(d["conf_matrix"]).all()
assert (d["i_ids"] // 105 < 70).all() and (d["j_ids"] // 105 < 70).all(), "match in the padded region"
print(f"outdoor 840x840 N={N}: {dt*1e3:.1f} ms per batch = {N/dt:.1f} pairs/s, M = {M} ({M/N:.0f} per pair), conf_matrix {tuple(d['conf_matrix'].shape)}")
