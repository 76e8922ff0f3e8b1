"""PCIe-inclusive throughput of the BASELINE configs[1] workload: every step starts from uint8 images
in HOST memory (what a data loader hands over after cv2.imread + cv2.resize), uploads them (1 B / pixel, pinned
staging), packs them on the device (loftr_pack_gray_u8) and runs LoFTR.forward.  bench.py's `value` has the
float32 images already resident in HBM; this is the number to quote when the boundary receives host buffers."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from loftr_amd import LoFTR, default_cfg, inputs

N, H, W, steps, warm = 8, 480, 640, 20, 3
cfg = copy.deepcopy(default_cfg); cfg["match_coarse"]["thr"] = 0.0
torch.manual_seed(0)
m = LoFTR(config=cfg).eval().to("cuda:0")
rng = np.random.default_rng(1234)
host0 = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(N)]
host1 = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(N)]

def step_host():
    batch = inputs.pack_pairs(host0, host1)
    m(batch)
    return batch

resident = inputs.pack_pairs(host0, host1)
def step_resident():
    batch = {"image0": resident["image0"], "image1": resident["image1"]}
    m(batch)
    return batch

for name, fn in (("resident fp32 images", step_resident), ("host uint8 images (upload + pack + forward)", step_host)):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print(f"{name}: {dt*1e3:.2f} ms / step, {N/dt:.1f} pairs/s")
