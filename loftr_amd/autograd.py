"""torch.autograd nodes of the model: forward AND backward are the HIP kernels behind the C-ABI (heads, losses' inputs, encoder
layers, FinePreprocess, position encoding, and -- at the end of this file -- the backbone's convolutions).

What the reference gets from autograd between ``batch['loss']`` (src/lightning/lightning_loftr.py:112-133) and the heads'
inputs, for the dual-softmax configuration (the Sinkhorn one swaps the first node):

    feat_c0, feat_c1 --CoarseMatching (coarse_matching.py:105-119)--> conf_matrix --LoFTRLoss (loftr_loss.py:22-99)---> loss_c
    feat_f0, feat_f1 --FineMatching   (fine_matching.py:43-57)-----> expec_f     --LoFTRLoss (loftr_loss.py:108-157)-> loss_f

The loss nodes are loftr_amd.training.LoFTRLoss (same mechanism).  The Sinkhorn head (coarse_matching.py:121-143,
conf_matrix_with_bin, the bin_score parameter) has its backward too: _SinkhornMatch.  Round 4: LoFTREncoderLayer is a node as well
(_EncoderLayer, csrc/encoder_bwd.hip), so both LocalFeatureTransformers are differentiable in their inputs and weights when they
run layer by layer (LocalFeatureTransformer.forward does that whenever a gradient is wanted); so are FinePreprocess
(_FinePreprocess), the position encoding (_PosEncodeFlatten) and every convolution of the backbone (_Conv2d).

No CPU fallback: the nodes call loftr_amd.ops, which raises on non-GPU tensors.
"""
import torch
from torch.autograd.function import once_differentiable

from . import ops


class _DualSoftmaxMatch(torch.autograd.Function):
    """conf_matrix = softmax(sim, 1) * softmax(sim, 2) with the match selection riding along (non-differentiable, like the
    reference's @torch.no_grad() get_coarse_match): `holder` receives ops.coarse_match's result dict."""

    @staticmethod
    def forward(ctx, feat_c0, feat_c1, hw0_c, hw1_c, kw, holder):
        r = ops.coarse_match(feat_c0.detach(), feat_c1.detach(), hw0_c, hw1_c, **kw)
        holder.update(r)
        ctx.save_for_backward(feat_c0, feat_c1)
        ctx.meta = (hw0_c, hw1_c, kw["temperature"], kw.get("mask0"), kw.get("mask1"))
        return r["conf_matrix"]

    @staticmethod
    @once_differentiable                                   # the backward kernels' outputs are constants to autograd: no double backward
    def backward(ctx, grad_conf):
        feat_c0, feat_c1 = ctx.saved_tensors
        hw0_c, hw1_c, temperature, mask0, mask1 = ctx.meta
        f0, f1 = feat_c0.detach().contiguous(), feat_c1.detach().contiguous()
        dsim = ops.dual_softmax_bwd(f0, f1, grad_conf.contiguous(), hw0_c, hw1_c, temperature, mask0, mask1)
        k = 1.0 / (f0.shape[-1] * temperature)           # sim = <feat_c0, feat_c1> / (C T): the two GEMMs are csrc/head_grads.hip
        g0, g1 = ops.head_feat_grads(dsim, f0, f1, k, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return g0, g1, None, None, None, None


def dual_softmax_match(feat_c0, feat_c1, hw0_c, hw1_c, **kw):
    """ops.coarse_match(match_type='dual_softmax') whose 'conf_matrix' carries the graph back to feat_c0 / feat_c1."""
    assert kw.get("match_type", "dual_softmax") == "dual_softmax" and kw.get("want_conf", True)
    holder = {}
    conf = _DualSoftmaxMatch.apply(feat_c0, feat_c1, tuple(hw0_c), tuple(hw1_c), kw, holder)
    holder["conf_matrix"] = conf
    return holder


class _SinkhornMatch(torch.autograd.Function):
    """conf_matrix_with_bin = exp(log_optimal_transport(sim, bin_score, iters)) (coarse_matching.py:121-143), differentiable with
    respect to feat_c0, feat_c1 and the bin_score parameter; the match selection rides along in `holder`."""

    @staticmethod
    def forward(ctx, feat_c0, feat_c1, bin_score, hw0_c, hw1_c, kw, holder):
        r = ops.coarse_match(feat_c0.detach(), feat_c1.detach(), hw0_c, hw1_c, **dict(kw, bin_score=float(bin_score.detach()), want_assign=True))
        holder.update(r)
        ctx.save_for_backward(feat_c0, feat_c1, bin_score)
        ctx.meta = (hw0_c, hw1_c, kw["skh_iters"], kw.get("mask0"), kw.get("mask1"))
        return r["conf_matrix_with_bin"]

    @staticmethod
    @once_differentiable                                   # the backward kernels' outputs are constants to autograd: no double backward
    def backward(ctx, grad_assign):
        feat_c0, feat_c1, bin_score = ctx.saved_tensors
        hw0_c, hw1_c, iters, mask0, mask1 = ctx.meta
        f0, f1 = feat_c0.detach().contiguous(), feat_c1.detach().contiguous()
        dz, dbin = ops.sinkhorn_bwd(f0, f1, grad_assign.contiguous(), hw0_c, hw1_c, float(bin_score.detach()), iters, mask0, mask1)
        L, S = f0.shape[1], f1.shape[1]
        dsim = dz[:, :L, :S]
        if mask0 is not None:                          # masked_fill_ cuts the graph at the padding (:124-127)
            dsim = dsim.masked_fill(~(mask0.bool()[..., None] & mask1.bool()[:, None]), 0.0)
        k = 1.0 / f0.shape[-1]                         # sim = <feat_c0, feat_c1> / C (no temperature, :123)
        g0, g1 = ops.head_feat_grads(dsim, f0, f1, k, ctx.needs_input_grad[0], ctx.needs_input_grad[1])    # a strided view: no copy
        gb = dbin.reshape(bin_score.shape).to(bin_score.dtype) if ctx.needs_input_grad[2] else None
        return g0, g1, gb, None, None, None, None


def sinkhorn_match(feat_c0, feat_c1, bin_score, hw0_c, hw1_c, **kw):
    """ops.coarse_match(match_type='sinkhorn') whose 'conf_matrix_with_bin' (and 'conf_matrix', its interior view, :133) carry the
    graph back to feat_c0 / feat_c1 / bin_score.  Training only: skh_prefilter never applies there (:136)."""
    assert kw.get("match_type") == "sinkhorn" and not kw.get("skh_prefilter")
    holder = {}
    assign = _SinkhornMatch.apply(feat_c0, feat_c1, bin_score, tuple(hw0_c), tuple(hw1_c), kw, holder)
    holder["conf_matrix_with_bin"] = assign
    holder["conf_matrix"] = assign[:, :-1, :-1]
    return holder


class _FineMatch(torch.autograd.Function):
    """expec_f [M,3] (differentiable) and the refined key points' offsets (not: get_fine_match is @torch.no_grad())."""

    @staticmethod
    def forward(ctx, feat_f0, feat_f1, mkpts1_c, b_ids, scale, scale1):
        expec, mk1f = ops.fine_match(feat_f0.detach(), feat_f1.detach(), mkpts1_c, b_ids, scale, scale1)
        ctx.save_for_backward(feat_f0, feat_f1)
        ctx.mark_non_differentiable(mk1f)
        return expec, mk1f

    @staticmethod
    @once_differentiable                                   # the backward kernels' outputs are constants to autograd: no double backward
    def backward(ctx, grad_expec, _):
        feat_f0, feat_f1 = ctx.saved_tensors
        g0, g1 = ops.fine_match_bwd(feat_f0.detach().contiguous(), feat_f1.detach().contiguous(), grad_expec.contiguous())
        return (g0 if ctx.needs_input_grad[0] else None), (g1 if ctx.needs_input_grad[1] else None), None, None, None, None


def fine_match(feat_f0, feat_f1, mkpts1_c, b_ids, scale, scale1=None):
    return _FineMatch.apply(feat_f0, feat_f1, mkpts1_c, b_ids, scale, scale1)


class _EncoderLayer(torch.autograd.Function):
    """One LoFTREncoderLayer (transformer.py:35-58): forward = the fused inference kernels (ops.encoder_layer), backward =
    loftr_encoder_layer_bwd (csrc/encoder_bwd.hip: recomputes the layer from (x, source), then differentiates it).  Differentiable
    in x, source and the ten weight tensors."""
    FIELDS = ("q_proj", "k_proj", "v_proj", "merge", "mlp0", "mlp2", "norm1_w", "norm1_b", "norm2_w", "norm2_b")

    @staticmethod
    def forward(ctx, x, source, x_mask, source_mask, nhead, *weights):
        w = dict(zip(_EncoderLayer.FIELDS, (t.detach() for t in weights)))
        out = ops.encoder_layer(x.detach().contiguous(), source.detach().contiguous(), ops.layer_weights_struct(w), nhead, x_mask, source_mask)
        ctx.save_for_backward(x, source, *weights)
        ctx.meta = (x_mask, source_mask, nhead)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, source, *weights = ctx.saved_tensors
        x_mask, source_mask, nhead = ctx.meta
        w = dict(zip(_EncoderLayer.FIELDS, (t.detach().contiguous() for t in weights)))
        gx, gs, gw = ops.encoder_layer_bwd(x.detach().contiguous(), source.detach().contiguous(), w, grad_out.contiguous(), nhead,
                                           x_mask, source_mask)
        need = ctx.needs_input_grad
        return ((gx if need[0] else None), (gs if need[1] else None), None, None, None,
                *[(gw[f] if need[5 + i] else None) for i, f in enumerate(_EncoderLayer.FIELDS)])


def encoder_layer(x, source, weights, nhead, x_mask=None, source_mask=None):
    """weights: dict(field -> tensor) (LoFTREncoderLayer.weight_tensors()).  The same tensor may be passed as x and source (self
    attention): autograd adds the two gradients."""
    return _EncoderLayer.apply(x, source, x_mask, source_mask, nhead, *[weights[f] for f in _EncoderLayer.FIELDS])


class _FinePreprocess(torch.autograd.Function):
    """FinePreprocess (fine_preprocess.py:29-59): forward ops.fine_preprocess, backward loftr_fine_preprocess_bwd (csrc/fine_bwd.hip).
    Differentiable in the two fine maps, the two coarse token tensors and the four parameters; the two outputs are returned as ONE
    stacked tensor [2M, WW, Cf] (the fine transformer runs on its halves)."""

    @staticmethod
    def forward(ctx, feat_f0, feat_f1, feat_c0, feat_c1, down_w, down_b, merge_w, merge_b, ids, geo):
        b_ids, i_ids, j_ids = ids
        hw0_c, hw1_c, W, stride = geo
        o0, o1 = ops.fine_preprocess(feat_f0.detach(), feat_f1.detach(), feat_c0.detach(), feat_c1.detach(), b_ids, i_ids, j_ids, hw0_c, hw1_c,
                                     W, stride, down_w=down_w.detach(), down_b=down_b.detach(), merge_w=merge_w.detach(),
                                     merge_b=merge_b.detach())
        ctx.save_for_backward(feat_f0, feat_f1, feat_c0, feat_c1, down_w, down_b, merge_w)
        ctx.meta = (ids, geo)
        stacked = ops.stacked_halves(o0, o1)
        return stacked if stacked is not None else torch.cat([o0, o1], 0)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        feat_f0, feat_f1, feat_c0, feat_c1, down_w, down_b, merge_w = ctx.saved_tensors
        (b_ids, i_ids, j_ids), (hw0_c, hw1_c, W, stride) = ctx.meta
        M = b_ids.shape[0]
        g = grad.contiguous()
        r = ops.fine_preprocess_bwd(feat_f0.detach(), feat_f1.detach(), feat_c0.detach().contiguous(), feat_c1.detach().contiguous(), b_ids,
                                    i_ids, j_ids, hw0_c, hw1_c, W, stride, down_w.detach(), down_b.detach(), merge_w.detach(), g[:M], g[M:])
        need = ctx.needs_input_grad
        return tuple(r[k] if need[k] else None for k in range(8)) + (None, None)


def fine_preprocess(feat_f0, feat_f1, feat_c0, feat_c1, ids, geo, down_w, down_b, merge_w, merge_b):
    """-> (feat_f0_unfold, feat_f1_unfold) [M, WW, Cf] with the graph."""
    st = _FinePreprocess.apply(feat_f0, feat_f1, feat_c0, feat_c1, down_w, down_b, merge_w, merge_b, ids, geo)
    M = ids[0].shape[0]
    return st[:M], st[M:]


class _PosEncodeFlatten(torch.autograd.Function):
    """(x + pe) rearranged 'n c h w -> n (h w) c' (position_encoding.py:37-42, loftr.py:58-59): the gradient is the upstream gradient
    in the input's layout (a view, no arithmetic)."""

    @staticmethod
    def forward(ctx, x, pe):
        ctx.shape = x.shape
        return ops.pos_encode_flatten(x.detach(), pe)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        n, c, h, w = ctx.shape
        return grad.reshape(n, h, w, c).permute(0, 3, 1, 2), None      # (reshape: the incoming gradient need not be contiguous)


def pos_encode_flatten(x, pe):
    return _PosEncodeFlatten.apply(x, pe)


def wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# ---- backbone training: nn.Conv2d(bias=False) with HIP forward, input gradient and weight gradient -------------------------------------
class _Conv2d(torch.autograd.Function):
    """F.conv2d(x, weight, None, stride, padding) of the backbone's bias-free convolutions (resnet_fpn.py:5-13, :52, :58-77) as an
    autograd node on the HIP convolutions.
      forward:  x -> channels-last -> scaled SP -> loftr_conv_bn_act (no BatchNorm folded) -> fp32;
      dL/dx:    the same kernels on the flipped, transposed filter (stride 2: on dy with zeros interleaved; 1 x 1 stride 2: the
                low-resolution product scattered to the even pixels);
      dL/dw:    loftr_conv_wgrad (split-K over the output pixels, ordered partial sums).
    BatchNorm (batch statistics), the activations with the residual add and the bilinear upsampling are HIP autograd nodes as well
    (_BatchNormTrain / _Act / _Upsample2x below, csrc/train_glue.hip); the stem's im2col and the FPN adds stay PyTorch autograd."""
    calls = 0                                                  # forward applications (tests check that the node is the one that ran)

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        _Conv2d.calls += 1
        Cout, Cin, KH, KW = weight.shape
        xn = x.detach().permute(0, 2, 3, 1).contiguous()
        if Cin % 4:                                            # the one-channel stem: its KH x KW patches as channels (1 x 1 problem)
            assert Cin == 1
            cols = torch.nn.functional.unfold(x.detach(), (KH, KW), padding=pad, stride=stride)       # [B, KH*KW, Ho*Wo]
            B, _, H, W = x.shape
            Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
            kk = (KH * KW + 3) // 4 * 4
            xn = torch.zeros(B, Ho, Wo, kk, dtype=x.dtype, device=x.device)
            xn[..., :KH * KW] = cols.transpose(1, 2).reshape(B, Ho, Wo, KH * KW)
            w1 = torch.zeros(Cout, kk, 1, 1, dtype=weight.dtype, device=weight.device)
            w1[:, :KH * KW, 0, 0] = weight.detach().reshape(Cout, KH * KW)
            xs, inv = ops.sp_from_nhwc(xn, scaled=True)
            y = ops.conv_raw(xs, kk, w1, 1, 0, x_inv_scale=inv)
            ctx.stem = (KH, KW, kk)
        else:
            xs, inv = ops.sp_from_nhwc(xn, scaled=True)
            y = ops.conv_raw(xs, Cin, weight.detach().contiguous(), stride, pad, x_inv_scale=inv)
            ctx.stem = None
        ctx.save_for_backward(xn, weight)
        ctx.geom = (stride, pad, tuple(x.shape))
        return y.permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xn, weight = ctx.saved_tensors
        stride, pad, xshape = ctx.geom
        Cout, Cin, KH, KW = weight.shape
        dyn = dy.permute(0, 2, 3, 1).contiguous()
        B, Ho, Wo, _ = dyn.shape
        dw = dx = None
        if ctx.needs_input_grad[1]:
            if ctx.stem is not None:
                kk = ctx.stem[2]
                dw = ops.conv_wgrad(dyn, xn, 1, 1, 1, 0)[:, :KH * KW, 0, 0].reshape(Cout, 1, KH, KW).contiguous()
            else:
                dw = ops.conv_wgrad(dyn, xn, KH, KW, stride, pad)
        if ctx.needs_input_grad[0]:
            if ctx.stem is not None:
                raise RuntimeError("the stem's input gradient is not implemented (images are leaves without gradients)")
            H, W = xshape[2], xshape[3]
            wt = weight.detach().flip(2, 3).transpose(0, 1).contiguous()                  # [Cin, Cout, KH, KW], taps reversed
            if stride == 1:
                ds, inv = ops.sp_from_nhwc(dyn, scaled=True)
                dxn = ops.conv_raw(ds, Cout, wt, 1, KH - 1 - pad, x_inv_scale=inv)
            elif KH == 1 and KW == 1 and pad == 0:
                ds, inv = ops.sp_from_nhwc(dyn, scaled=True)
                low = ops.conv_raw(ds, Cout, wt, 1, 0, x_inv_scale=inv)
                dxn = torch.zeros(B, H, W, Cin, dtype=dyn.dtype, device=dyn.device)
                dxn[:, 0:stride * Ho:stride, 0:stride * Wo:stride] = low
            else:
                # transposed convolution = stride-1 convolution of dy with stride - 1 zeros between its pixels (and the rows / columns
                # the forward never reached appended), filter flipped, padding K - 1 - pad
                Hz, Wz = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1
                dz = torch.zeros(B, Hz, Wz, Cout, dtype=dyn.dtype, device=dyn.device)
                dz[:, 0:stride * Ho:stride, 0:stride * Wo:stride] = dyn
                ds, inv = ops.sp_from_nhwc(dz, scaled=True)
                dxn = ops.conv_raw(ds, Cout, wt, 1, KH - 1 - pad, x_inv_scale=inv)
            assert tuple(dxn.shape) == (B, H, W, Cin), (dxn.shape, xshape)
            dx = dxn.permute(0, 3, 1, 2)
        return dx, dw, None, None


def conv2d(x, weight, stride=1, padding=0):
    """Differentiable bias-free convolution on the HIP kernels (see _Conv2d)."""
    return _Conv2d.apply(x, weight, int(stride), int(padding))


# ---- training-mode glue of the backbone: BatchNorm with batch statistics, activations (+ residual add), bilinear x2 upsampling ---------
class _BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm2d.forward in .train() mode (resnet_fpn.py:25-26,36,68 ...) on csrc/train_glue.hip; the running statistics are updated by
    the module (backbone.BatchNorm2d) from the returned batch mean / unbiased variance."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        xc = x.detach()
        xc = xc if (xc.is_contiguous() or xc.is_contiguous(memory_format=torch.channels_last)) else xc.contiguous()
        y, mean, invstd, varu = ops.bn_train_fwd(xc, None if gamma is None else gamma.detach(), None if beta is None else beta.detach(), eps)
        ctx.save_for_backward(xc, mean, invstd, gamma)
        ctx.mark_non_differentiable(mean, varu)
        return y, mean, varu

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dm, _dv):
        x, mean, invstd, gamma = ctx.saved_tensors
        dx, dgamma, dbeta = ops.bn_train_bwd(dy, x, mean, invstd, None if gamma is None else gamma.detach())
        return dx, (dgamma if gamma is not None else None), (dbeta if ctx.needs_input_grad[2] else None), None


def batch_norm_train(x, gamma, beta, eps):
    return _BatchNormTrain.apply(x, gamma, beta, eps)


class _Act(torch.autograd.Function):
    """act(a [+ b]): nn.ReLU / nn.LeakyReLU, and BasicBlock's relu(x + y) (resnet_fpn.py:40) in one pass; the backward needs the output only."""

    @staticmethod
    def forward(ctx, a, b, act, slope):
        y = ops.act_fwd(a.detach(), None if b is None else b.detach(), act, slope)
        ctx.save_for_backward(y)
        ctx.act, ctx.slope, ctx.has_b = act, slope, b is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = ops.act_bwd(dy, y, ctx.act, ctx.slope)
        return dx, (dx if ctx.has_b else None), None, None


def act(a, b=None, kind="relu", slope=0.01):
    return _Act.apply(a, b, kind, slope)


class _Upsample2x(torch.autograd.Function):
    """F.interpolate(x, scale_factor=2., mode='bilinear', align_corners=True) (resnet_fpn.py:110,115) and its adjoint."""

    @staticmethod
    def forward(ctx, x):
        return ops.upsample2x_bilinear(x.detach())

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.upsample2x_bilinear_bwd(dy)


def upsample2x(x):
    return _Upsample2x.apply(x)
