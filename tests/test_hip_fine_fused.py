"""The fused fine-level transformer (csrc/fine_fused.hip: both layers of a match in one launch) against a float64
restatement of the reference layers (transformer.py:35-58,80-101, linear_attention.py:20-47), over match counts that
leave partly filled workgroups, window sizes below the 32 token slots, and input magnitudes that push the attention
operands (V / S, K^T V, z Q) to both ends of the fp16 range -- the kernel scales them by run-time powers of two."""
import numpy as np
import pytest
import torch

from loftr_amd import LoFTR, get_cfg, ops
from loftr_amd.synth import make_weights

pytestmark = pytest.mark.gpu


def _enc64(x, s, w, p):
    W = lambda n: torch.from_numpy(np.asarray(w[p + n])).double().to(x.device)
    q, k, v = x @ W("q_proj.weight").T, s @ W("k_proj.weight").T, s @ W("v_proj.weight").T
    B, L, _ = q.shape
    Q = torch.nn.functional.elu(q.view(B, L, 8, 16)) + 1
    K = torch.nn.functional.elu(k.view(B, -1, 8, 16)) + 1
    V = v.view(B, -1, 8, 16) / s.shape[1]
    KV = torch.einsum("nshd,nshv->nhdv", K, V)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
    msg = (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * s.shape[1]).reshape(B, L, 128)
    msg = torch.nn.functional.layer_norm(msg @ W("merge.weight").T, (128,), W("norm1.weight"), W("norm1.bias"))
    h = torch.relu(torch.cat([x, msg], 2) @ W("mlp.0.weight").T) @ W("mlp.2.weight").T
    return x + torch.nn.functional.layer_norm(h, (128,), W("norm2.weight"), W("norm2.bias"))


@pytest.fixture(scope="module")
def fine():
    cfg = get_cfg(thr=0.0)
    w = make_weights(3, cfg)
    model = LoFTR(cfg).eval()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in w.items()}, strict=False)
    return model.cuda().loftr_fine, w


@pytest.mark.parametrize("M,T,scale", [(3, 25, 1.0), (130, 25, 1.0), (1100, 25, 1.0), (257, 25, 40.0), (257, 25, 300.0),
                                       (257, 25, 0.01), (64, 9, 1.0), (33, 32, 5.0), (5, 1, 1.0)])
def test_fused_fine_transformer_vs_fp64(fine, M, T, scale):
    tf, w = fine
    g = torch.Generator(device="cpu").manual_seed(M * 131 + T)
    f0 = (scale * torch.randn(M, T, 128, generator=g)).cuda()
    f1 = (0.6 * f0 + 0.8 * scale * torch.randn(M, T, 128, generator=g).cuda()).contiguous()
    with torch.no_grad():
        o0, o1 = tf(f0, f1)
        x0, x1 = f0.double(), f1.double()
        x0, x1 = _enc64(x0, x0, w, "loftr_fine.layers.0."), _enc64(x1, x1, w, "loftr_fine.layers.0.")
        x0 = _enc64(x0, x1, w, "loftr_fine.layers.1.")
        x1 = _enc64(x1, x0, w, "loftr_fine.layers.1.")
    assert torch.isfinite(o0).all() and torch.isfinite(o1).all()
    # the residual stream dominates the magnitude: fp32-class agreement relative to it (the per-layer kernels measure 3e-6 at scale 1)
    tol = 2e-6 * max(1.0, float(x0.abs().max())) + 1e-5
    assert float((o0.double() - x0).abs().max()) <= tol and float((o1.double() - x1).abs().max()) <= tol
    assert (f0 - (scale * torch.randn(M, T, 128, generator=torch.Generator(device="cpu").manual_seed(M * 131 + T))).cuda()).abs().max() == 0   # inputs untouched (not inplace)


def test_fused_path_is_the_one_that_runs(fine):
    """The per-layer kernels stay as the fallback for other shapes; the [self, cross] / C = 128 / T <= 32 case must take fine_pair_kernel."""
    import ctypes as C
    from loftr_amd import _lib
    tf, _ = fine
    lib = _lib.load()
    ids = {lib.loftr_hip_timing_kernel_name(i).decode(): i for i in range(lib.loftr_hip_timing_kernel_count())}
    f = torch.randn(2, 8, 25, 128, device="cuda")
    lib.loftr_hip_timing_enable(1 << ids["fine_pair_kernel"])
    with torch.no_grad():
        tf(f[0].contiguous(), f[1].contiguous())
    torch.cuda.synchronize()
    lib.loftr_hip_timing_enable(0)
    ms, n = C.c_double(0), C.c_longlong(0)
    lib.loftr_hip_timing_read(ids["fine_pair_kernel"], C.byref(ms), C.byref(n), 1)
    assert n.value == 1
