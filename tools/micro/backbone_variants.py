"""Time the PyTorch-ROCm ResNet-FPN backbone (fp32) under MIOpen knobs: batch 16 x 480x640."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loftr_amd.backbone import build_backbone
from loftr_amd.config import default_cfg

def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

torch.manual_seed(0)
m = build_backbone(default_cfg).eval().cuda()
x = torch.rand(16, 1, 480, 640, device="cuda")
with torch.no_grad():
    ref = m(x)
    print("default           %.2f ms" % timeit(lambda: m(x)))
    torch.backends.cudnn.benchmark = True
    print("cudnn.benchmark   %.2f ms" % timeit(lambda: m(x)))
    mc = m.to(memory_format=torch.channels_last); xc = x.contiguous(memory_format=torch.channels_last)
    print("channels_last+bm  %.2f ms" % timeit(lambda: mc(xc)))
    o = mc(xc); print("  out strides", o[0].stride(), o[1].stride(), "max diff", (o[0]-ref[0]).abs().max().item(), (o[1]-ref[1]).abs().max().item())
    torch.backends.cudnn.benchmark = False
    print("channels_last     %.2f ms" % timeit(lambda: mc(xc)))
    m2 = build_backbone(default_cfg).eval().cuda().half(); xh = x.half()
    print("fp16 (reference only, not parity-safe) %.2f ms" % timeit(lambda: m2(xh)))
