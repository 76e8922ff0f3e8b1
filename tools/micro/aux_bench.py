"""Roofline check of the two HBM-bound helper kernels around the path:
loftr_epipolar_errors (28 B / match algorithmic) and loftr_pack_gray_u8 (1 B in, 4 B + 1 B out per pixel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from loftr_amd import ops, inputs, _lib
import ctypes as C

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
M, N = 16_000_000, 64
p0 = (torch.rand(M, 2, generator=g) * 600).to(dev); p1 = (torch.rand(M, 2, generator=g) * 600).to(dev)
bids = torch.sort(torch.randint(0, N, (M,), generator=g))[0].to(dev)
T = torch.eye(4).repeat(N, 1, 1); T[:, :3, 3] = torch.randn(N, 3, generator=g); T = T.to(dev)
K = torch.tensor([[500.0, 0, 320], [0, 500, 240], [0, 0, 1]]).repeat(N, 1, 1).to(dev)
for _ in range(3): ops.epipolar_errors(p0, p1, bids, T, K, K)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): ops.epipolar_errors(p0, p1, bids, T, K, K)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"epipolar_errors M={M}: {dt*1e6:.1f} us  {M*28/dt/1e9:.0f} GB/s algorithmic (28 B/match)")

lib = _lib.load()
Nimg, P = 64, 840
src = torch.randint(0, 256, (Nimg, P, P), dtype=torch.uint8, generator=g).to(dev)
hw = torch.tensor([[840, 560]] * Nimg, dtype=torch.int32).to(dev)
img = torch.empty(Nimg, 1, P, P, device=dev); mask = torch.empty(Nimg, P, P, dtype=torch.uint8, device=dev)
mc = torch.empty(Nimg, P // 8, P // 8, dtype=torch.uint8, device=dev)
def run():
    ops.check(lib.loftr_pack_gray_u8(ops._ptr(src), P * P, P, ops._ptr(hw), Nimg, P, P, ops._ptr(img), ops._ptr(mask), ops._ptr(mc), 8, ops._stream()), "pack")
for _ in range(3): run()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
px = Nimg * P * P
print(f"pack_gray_u8 {Nimg} x {P}x{P}: {dt*1e6:.1f} us  {px*(560/840 + 5)/dt/1e9:.0f} GB/s algorithmic (valid bytes in + 5 B/px out)")
