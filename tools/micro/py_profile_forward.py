import cProfile, pstats, sys, io, time
sys.path.insert(0, '/root/repo')
import torch
from loftr_amd import LoFTR
from loftr_amd.config import get_cfg
from loftr_amd.synth import make_images
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
cfg = get_cfg(thr=0.0); cfg["coarse"]["temp_bug_fix"] = True
model = LoFTR(cfg).eval().cuda()
i0, i1 = make_images(1234, NB, 480, 640)
a, b = torch.from_numpy(i0).cuda(), torch.from_numpy(i1).cuda()
for _ in range(5): model({"image0": a, "image1": b})
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100): model({"image0": a, "image1": b})
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('cumulative')
ps.print_stats(45)
print(s.getvalue()[:9000])
