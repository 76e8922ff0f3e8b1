#!/usr/bin/env python
"""The backward entry points at the sizes of a batch-8 training step (grids of far more than 256 workgroups -- the sizes at which
head_grad_kernel's co-residency fault showed and which the goldens, N <= 2, do not reach) against float64 torch autograd of the same
mathematics, restated here in a few lines each:
  LoFTREncoderLayer (transformer.py:35-58 with LinearAttention, linear_attention.py:20-47), coarse (16 x 4800 x 256) and fine
  (15000 windows x 25 x 128); the dual-softmax head (coarse_matching.py:105-119) at N = 8.
Reading the output: a hidden unit whose pre-activation is zero to rounding gets a different ReLU mask in float32 and float64, which
moves ONE token's gradient by ~1e-2 of the maximum; at millions of hidden units a few such tokens are float32's own (torch's float32
autograd is printed next to ours).  A fault of the kernels shows as MANY rows off (the co-residency fault: thousands).
    python tools/micro/backward_scale_check.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loftr_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def layer_ref(x, src, w, H, xm=None, sm=None):
    nb, L, C = x.shape; S = src.shape[1]; D = C // H
    q = (x @ w["q_proj"].T).view(nb, L, H, D); k = (src @ w["k_proj"].T).view(nb, S, H, D); v = (src @ w["v_proj"].T).view(nb, S, H, D)
    Q, K = F.elu(q) + 1, F.elu(k) + 1
    if xm is not None:
        Q = Q * xm[:, :, None, None]
    if sm is not None:
        K = K * sm[:, :, None, None]; v = v * sm[:, :, None, None]
    v = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
    msg = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S
    msg = msg.reshape(nb, L, C) @ w["merge"].T
    msg = F.layer_norm(msg, (C,), w["norm1_w"], w["norm1_b"])
    h = torch.relu(torch.cat([x, msg], -1) @ w["mlp0"].T) @ w["mlp2"].T
    return x + F.layer_norm(h, (C,), w["norm2_w"], w["norm2_b"])


def check_layer(nb, L, S, C, H, tag):
    g = torch.Generator().manual_seed(nb + L)
    x = torch.randn(nb, L, C, generator=g).to(dev); src = x if S == L else torch.randn(nb, S, C, generator=g).to(dev)
    w = {k: (torch.randn(shp, generator=g) / (shp[-1] ** 0.5 if len(shp) == 2 else 4.0) + (1.0 if k.endswith("_w") else 0.0)).to(dev)
         for k, shp in ops.GRAD_FIELD_SHAPES(C).items()}
    go = (torch.randn(nb, L, C, generator=g) * 1e-3).to(dev)
    gx, gs, gw = ops.encoder_layer_bwd(x, src, w, go, H)
    # float64 autograd, chunked over the batch to bound memory (the weight gradients add up)
    wd = {k: v.double().requires_grad_(True) for k, v in w.items()}
    rx, rs = torch.empty_like(x, dtype=torch.float64), torch.empty_like(src, dtype=torch.float64)
    step = max(1, nb // 8)
    for i in range(0, nb, step):
        xi = x[i:i + step].double().requires_grad_(True)
        si = xi if S == L and src is x else src[i:i + step].double().requires_grad_(True)
        out = layer_ref(xi, si, wd, H)
        out.backward(go[i:i + step].double())
        if si is xi:
            rx[i:i + step] = xi.grad; rs[i:i + step] = 0
        else:
            rx[i:i + step] = xi.grad; rs[i:i + step] = si.grad
    # float32 torch autograd of the same restatement: its own distance to float64 is the yardstick (a hidden unit whose pre-activation
    # is zero to rounding flips its ReLU mask between two evaluations: errors of 1e-2 of the maximum are float32's own at these sizes)
    wf = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    fx = torch.empty_like(x)
    for i in range(0, nb, step):
        xi = x[i:i + step].clone().requires_grad_(True)
        si = xi if S == L and src is x else src[i:i + step].clone().requires_grad_(True)
        layer_ref(xi, si, wf, H).backward(go[i:i + step])
        fx[i:i + step] = xi.grad
    f32x = float((fx.double() - rx).abs().max() / rx.abs().max())
    f32w = max(float((wf[k].grad.double() - wd[k].grad).abs().max() / wd[k].grad.abs().max()) for k in wf)
    self_attn = S == L and src is x
    tot = gx.double() + (gs.double() if self_attn else 0)
    bad_rows = int(((tot - rx).abs().amax(-1) > 1e-5 * rx.abs().max()).sum())
    bad_rows32 = int(((fx.double() - rx).abs().amax(-1) > 1e-5 * rx.abs().max()).sum())
    ex = float((gx.double() + (gs.double() if self_attn else 0) - rx).abs().max() / rx.abs().max())
    es = 0.0 if self_attn else float((gs.double() - rs).abs().max() / rs.abs().max())
    worst = max((float((gw[k].double() - wd[k].grad).abs().max() / wd[k].grad.abs().max()), k) for k in gw)
    print(f"{tag}: nb={nb} L={L} S={S} C={C}: grad_x {ex:.1e} grad_source {es:.1e} worst weight gradient {worst[0]:.1e} ({worst[1]})   | torch float32: grad_x {f32x:.1e} weights {f32w:.1e}"
          f"   | token rows of grad_x off by > 1e-5 of the maximum: {bad_rows} of {nb * L} (torch float32: {bad_rows32})", flush=True)
    if os.environ.get("ALL"):
        print("   " + "  ".join(f"{k} {float((gw[k].double() - wd[k].grad).abs().max() / wd[k].grad.abs().max()):.1e}" for k in gw), flush=True)
        out = ops.encoder_layer(x, src, ops.layer_weights_struct(w), H)
        ref = torch.cat([layer_ref(x[i:i + step].double(), (x if src is x else src)[i:i + step].double(), {k: v.detach() for k, v in wd.items()}, H) for i in range(0, nb, step)])
        print(f"   forward (ops.encoder_layer) {float((out.double() - ref).abs().max() / ref.abs().max()):.1e}", flush=True)


def check_dual_softmax(N, L, C):
    g = torch.Generator().manual_seed(N)
    f0 = torch.randn(N, L, C, generator=g).to(dev); f1 = torch.randn(N, L, C, generator=g).to(dev)
    gc = (torch.rand(N, L, L, generator=g) * 1e-3).to(dev)
    hw = (60, 80)
    dsim = ops.dual_softmax_bwd(f0, f1, gc, hw, hw, 0.1)
    g0, g1 = ops.head_feat_grads(dsim, f0, f1, 1.0 / (C * 0.1))
    worst = 0.0
    for n in range(N):
        a, b = f0[n].double().requires_grad_(True), f1[n].double().requires_grad_(True)
        sim = (a / C ** 0.5) @ (b / C ** 0.5).T / 0.1
        conf = F.softmax(sim, 0) * F.softmax(sim, 1)
        conf.backward(gc[n].double())
        worst = max(worst, float((g0[n].double() - a.grad).abs().max() / a.grad.abs().max()), float((g1[n].double() - b.grad).abs().max() / b.grad.abs().max()))
    print(f"dual-softmax head: N={N} L=S={L}: worst feature gradient error {worst:.1e}", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "list":            # list nb,L,S,C nb,L,S,C ...
    for spec in sys.argv[2:]:
        nb, L, S, C = (int(v) for v in spec.split(","))
        check_layer(nb, L, S, C, 8, "layer")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "small":
    check_layer(2, 300, 300, 256, 8, "coarse self layer, small")
    check_layer(2, 300, 293, 256, 8, "coarse cross layer, small")
    check_layer(2, 4800, 4800, 256, 8, "coarse self layer, N=1")
    check_layer(4, 4800, 4800, 256, 8, "coarse self layer, N=2")
    check_layer(40, 25, 25, 128, 8, "fine self layer, 20 matches")
    check_layer(2000, 25, 25, 128, 8, "fine self layer, 1000 matches")
    sys.exit(0)
check_layer(16, 4800, 4800, 256, 8, "coarse self layer, batch 8")
check_layer(8, 4800, 4800 - 7, 256, 8, "coarse cross layer, batch 8")
check_layer(15000, 25, 25, 128, 8, "fine self layer, 7500 matches")
check_dual_softmax(8, 4800, 256)
