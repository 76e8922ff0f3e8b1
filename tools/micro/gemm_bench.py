"""Isolated GEMM micro-benchmark through the C-ABI (loftr_linear_fwd): the encoder's shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loftr_amd import ops
shapes = [(76800, 512, 512), (76800, 256, 256), (76800, 768, 256), (38400, 512, 512), (76800, 256, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1].split(","))]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05
    for _ in range(3): ops.linear(a, w)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): ops.linear(a, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    ref = (a[:256].double() @ w.double().T)
    err = (ops.linear(a, w)[:256].double() - ref).abs().max().item()
    print(f"M={M} N={N} K={K}: {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:7.1f} TFLOP/s (fp32-equivalent)  max err {err:.2e}")
