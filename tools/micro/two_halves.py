"""Upper bound of sub-batch pipelining: two half batches (4 pairs each) run concurrently from two host threads on
their own streams vs one batch of 8 pairs on one stream."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loftr_amd import LoFTR, get_cfg
from loftr_amd.synth import make_images, make_weights
def make():
    torch.manual_seed(0)
    cfg = get_cfg(thr=0.0); cfg["coarse"]["temp_bug_fix"] = True
    m = LoFTR(cfg).eval()
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}, strict=False)
    return m.cuda()
i0, i1 = make_images(1234, 8, 480, 640)
img0, img1 = torch.from_numpy(i0).cuda(), torch.from_numpy(i1).cuda()
def loop(model, a, b, steps, stream):
    with torch.cuda.stream(stream):
        for _ in range(steps):
            model({"image0": a, "image1": b})
m1, m2 = make(), make()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
loop(m1, img0, img1, 3, s0); loop(m2, img0[:4], img1[:4], 3, s1); loop(m1, img0[4:], img1[4:], 2, s0)
torch.cuda.synchronize()
steps = 10
t = time.perf_counter(); loop(m1, img0, img1, steps, s0); torch.cuda.synchronize(); t8 = (time.perf_counter() - t) / steps
t = time.perf_counter()
th = [threading.Thread(target=loop, args=(m1, img0[:4], img1[:4], steps, s0)), threading.Thread(target=loop, args=(m2, img0[4:], img1[4:], steps, s1))]
[x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize(); t44 = (time.perf_counter() - t) / steps
t = time.perf_counter(); loop(m1, img0[:4], img1[:4], steps, s0); torch.cuda.synchronize(); t4 = (time.perf_counter() - t) / steps
print(f"1 x batch 8: {t8*1e3:.2f} ms/step = {8/t8:.1f} pairs/s | 2 threads x batch 4: {t44*1e3:.2f} ms = {8/t44:.1f} pairs/s | 1 x batch 4: {t4*1e3:.2f} ms = {4/t4:.1f} pairs/s")
