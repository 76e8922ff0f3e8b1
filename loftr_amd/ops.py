"""Tensor-level wrappers over the C-ABI (one function per entry point of include/loftr_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic of the matching
path happens in the HIP kernels.  Every wrapper requires CUDA(ROCm) float32 contiguous tensors
and raises otherwise -- there is no CPU fallback.
"""
import ctypes as C
import os
import weakref
import functools

import torch

from . import _lib
from ._lib import CoarseParams, FMap, LayerWeights, MatchOut, check

_WS = {}          # device index -> cached workspace tensor (grown on demand, never shrunk)

LAYER_FIELDS = (("q_proj", "q_proj.weight"), ("k_proj", "k_proj.weight"), ("v_proj", "v_proj.weight"),
                ("merge", "merge.weight"), ("mlp0", "mlp.0.weight"), ("mlp2", "mlp.2.weight"),
                ("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"),
                ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"))


# Prepared (re-encoded) weights per module, keyed weakly by the module: kept OUT of the modules' __dict__ so that
# pickle / torch.save(model) / spawn-based launchers keep working after a forward (weak references do not pickle).
_PREPARED = weakref.WeakKeyDictionary()


def _tensors_in(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors_in(o)
    elif isinstance(obj, dict):                   # the batch dict of training.spvs_* / LoFTRLoss: its tensors decide the device
        for o in obj.values():
            yield from _tensors_in(o)
    elif isinstance(obj, torch.nn.Module):
        for prm in obj.parameters():
            yield prm
            break


def _on_device(fn):
    """Run a wrapper with the GPU of its tensor arguments current: the C entry points launch on the stream handle they
    are given and never call hipSetDevice, so stream, workspace and pointers must all belong to ONE device -- the
    tensors' device, not whatever device happens to be current (a model on cuda:1 without torch.cuda.set_device).
    Tensors on different devices are rejected."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = None
        for t in _tensors_in(list(args) + list(kwargs.values())):
            if t.is_cuda:
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise _lib.LoftrHipError(f"{fn.__name__}: tensors on different devices ({dev} and {t.device})")
        if dev is None:
            return fn(*args, **kwargs)            # no GPU tensor: the body raises its own 'expected a GPU tensor'
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _need(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.LoftrHipError(f"{name}: expected a GPU tensor (the HIP matching path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.LoftrHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.LoftrHipError(f"{name}: expected a contiguous tensor")
    return t


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def debug_set(key, value):
    """loftr_hip_debug_set: a named A/B switch of the library (include/loftr_hip.h; process-global, the library reads no environment variable)."""
    check(_lib.load().loftr_hip_debug_set(key.encode(), int(value)), f"loftr_hip_debug_set({key})")


def debug_get(key):
    """(value, default) of a debug switch."""
    v, d = C.c_int(0), C.c_int(0)
    check(_lib.load().loftr_hip_debug_get(key.encode(), C.byref(v), C.byref(d)), f"loftr_hip_debug_get({key})")
    return v.value, d.value


class debug_switch:
    """``with ops.debug_switch(conv_duo=0, conv_persist_cap=8): ...`` -- switches set for the block, restored afterwards."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = debug_get(k)[0]
            debug_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            debug_set(k, v)
        return False


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes, device):
    """Cached scratch buffer of at least nbytes on `device`, one per (device, that device's current stream): calls
    on one stream may share it (stream ordered), concurrent streams must not."""
    idx = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(idx)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _WS[idx] = buf
    return buf


def _mask_u8(m, name):
    if m is None:
        return None
    if m.dtype == torch.bool:
        m = m.to(torch.uint8)
    return _need(m.contiguous(), name, torch.uint8)


# ---------------------------------------------------------------------------------------------
@_on_device
def linear(a, w):
    """a [M,K] @ w[N,K]^T on the split-fp16 GEMM core (building block, exposed for tests)."""
    _need(a, "a"); _need(w, "w")
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    lib = _lib.load()
    ws = workspace(lib.loftr_linear_workspace_bytes(M, N, K), a.device)
    check(lib.loftr_linear_fwd(_ptr(a), _ptr(w), _ptr(out), M, N, K, _ptr(ws), ws.numel(), _stream()), "loftr_linear_fwd")
    return out


@_on_device
def pos_encode_flatten(feat, pe):
    """feat [N,C,H,W] (any strides, e.g. channels-last) + pe[:, :H, :W], flattened to [N, H*W, C]."""
    if not feat.is_cuda or feat.dtype != torch.float32:
        raise _lib.LoftrHipError("feat: expected a float32 GPU tensor (the HIP matching path has no CPU fallback)")
    _need(pe, "pe")
    N, Cc, H, W = feat.shape
    out = torch.empty(N, H * W, Cc, device=feat.device, dtype=torch.float32)
    fm = _fmap(feat)
    check(_lib.load().loftr_pos_encode_flatten(C.byref(fm), _ptr(pe), pe.shape[-2], pe.shape[-1], _ptr(out),
                                               N, Cc, _stream()), "loftr_pos_encode_flatten")
    return out


def layer_weights_struct(tensors):
    """dict(field -> tensor) -> LayerWeights (keeps nothing alive: caller holds the tensors)."""
    lw = LayerWeights()
    for field, _ in LAYER_FIELDS:
        setattr(lw, field, tensors[field].data_ptr())
    return lw


@_on_device
def encoder_layer(x, source, w_struct, nhead, x_mask=None, source_mask=None, out=None):
    """One LoFTREncoderLayer.  x [nb,L,C], source [nb,S,C] -> [nb,L,C]."""
    _need(x, "x"); _need(source, "source")
    nb, L, Cc = x.shape
    S = source.shape[1]
    xm, sm = _mask_u8(x_mask, "x_mask"), _mask_u8(source_mask, "source_mask")
    if x_mask is not None and x_mask is source_mask:
        sm = xm
    out = torch.empty_like(x) if out is None else out
    lib = _lib.load()
    nbytes = lib.loftr_encoder_workspace_bytes(nb, L, S, Cc)
    ws = workspace(nbytes, x.device)
    check(lib.loftr_encoder_layer_fwd(_ptr(x), _ptr(source), _ptr(xm), _ptr(sm), C.byref(w_struct), _ptr(out),
                                      nb, L, S, Cc, nhead, _ptr(ws), ws.numel(), _stream()),
          "loftr_encoder_layer_fwd")
    return out


GRAD_FIELD_SHAPES = lambda Cc: {"q_proj": (Cc, Cc), "k_proj": (Cc, Cc), "v_proj": (Cc, Cc), "merge": (Cc, Cc), "mlp0": (2 * Cc, 2 * Cc),
                                "mlp2": (Cc, 2 * Cc), "norm1_w": (Cc,), "norm1_b": (Cc,), "norm2_w": (Cc,), "norm2_b": (Cc,)}


@_on_device
def encoder_layer_bwd(x, source, weights, grad_out, nhead, x_mask=None, source_mask=None):
    """Backward of one LoFTREncoderLayer (transformer.py:35-58 under autograd): weights = dict(field -> tensor) as for
    layer_weights_struct.  Returns (grad_x [nb,L,C], grad_source [nb,S,C], dict(field -> weight gradient))."""
    _need(x, "x"); _need(source, "source"); _need(grad_out, "grad_out")
    nb, L, Cc = x.shape
    S = source.shape[1]
    assert tuple(grad_out.shape) == (nb, L, Cc)
    xm, sm = _mask_u8(x_mask, "x_mask"), _mask_u8(source_mask, "source_mask")
    dev = x.device
    gx, gs = torch.empty_like(x), torch.empty_like(source)
    grads = {k: torch.empty(shp, device=dev, dtype=torch.float32) for k, shp in GRAD_FIELD_SHAPES(Cc).items()}
    lib = _lib.load()
    ws = workspace(lib.loftr_encoder_layer_bwd_workspace_bytes(nb, L, S, Cc, nhead), dev)
    wst, gst = layer_weights_struct(weights), layer_weights_struct(grads)
    check(lib.loftr_encoder_layer_bwd(_ptr(x), _ptr(source), _ptr(xm), _ptr(sm), C.byref(wst), _ptr(grad_out), _ptr(gx), _ptr(gs),
                                      C.byref(gst), nb, L, S, Cc, nhead, _ptr(ws), ws.numel(), _stream()), "loftr_encoder_layer_bwd")
    return gx, gs, grads


def stacked_halves(a, b):
    """The tensor [a; b] WITHOUT a copy when a and b already are the two batch halves of one buffer
    (e.g. ``x.split(n)`` of a stacked pair batch, or the outputs of pos_encode_flatten / fine_preprocess);
    None otherwise."""
    if (a.shape[1:] != b.shape[1:] or a.stride() != b.stride() or a.dtype != b.dtype or a.device != b.device
            or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
            or b.storage_offset() != a.storage_offset() + a.shape[0] * a.stride(0)):
        return None
    return torch.as_strided(a, (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())


@_on_device
def transformer_prepare(layer_structs, Cc, device):
    """All layer matrices re-encoded once into the library's GEMM operand format (loftr_transformer_prepare)."""
    n_layers = len(layer_structs)
    lib = _lib.load()
    buf = torch.empty(lib.loftr_transformer_prepared_bytes(n_layers, Cc), dtype=torch.uint8, device=device)
    arr = (LayerWeights * n_layers)(*layer_structs)
    with torch.cuda.device(device):
        check(lib.loftr_transformer_prepare(arr, n_layers, Cc, _ptr(buf), buf.numel(), _stream()), "loftr_transformer_prepare")
    return buf


# The coarse transformer's work queue for a shape (loftr_coarse_plan_build), per device: (device, kinds, N, L, S, order) -> uint8 tensor
_COARSE_PLANS = {}
# "persistent": one dependency-driven launch per coarse transformer call;  "persistent_call_order": the same kernel on the
# reference's call order (bit-identical results; A/B and tests);  "launches": the per-call launches of loftr_transformer_fwd;
# "auto" (default): persistent from 8 pairs on.  The dependencies are per pair, so with few pairs the 256 resident workgroups wait on each
# other -- and a resident workgroup holds its CU, which the FPN fine branch on the side stream then cannot use.  Alone the persistent form wins
# from 2 pairs on (8 pairs 3.19 vs 3.56 ms, 2 pairs of 840 x 840 2.36 vs 2.42 ms, a single 640 x 480 pair 2.06 vs 1.51 ms: tools/micro/pct_check.py,
# profiles/r06_pct_check.txt); INSIDE the forward it loses below 8 pairs (640 x 480: 1 pair 5.1-6.0 vs 4.5 ms, 2 pairs 7.5 vs 6.6, 4 pairs 11.6 vs
# 10.9, 8 pairs 20.0 vs 20.1-20.2, 16 pairs 39.3 vs 39.3; 840 x 840: 1 / 2 / 4 pairs 8.2 / 13.2 / 24.3 vs 7.3 / 12.6 / 23.9 ms;
# profiles/r06_mode_sweep.txt)
COARSE_MODE = os.environ.get("LOFTR_COARSE_MODE", "auto")
COARSE_AUTO_MIN_PAIRS = 8


def coarse_plan(kinds, N, L, S, device, order=0):
    """The persistent coarse transformer's plan for this shape, built once (None: the shape has no persistent form)."""
    key = (str(device), tuple(kinds), N, L, S, order)
    if key not in _COARSE_PLANS:
        lib = _lib.load()
        arr = (C.c_int * len(kinds))(*kinds)
        nbytes = lib.loftr_coarse_plan_bytes(arr, len(kinds), N, L, S)
        plan = None
        if nbytes:
            plan = torch.empty(nbytes, dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                check(lib.loftr_coarse_plan_build(arr, len(kinds), N, L, S, order, _ptr(plan), plan.numel(), _stream()), "loftr_coarse_plan_build")
        _COARSE_PLANS[key] = plan
    return _COARSE_PLANS[key]


@_on_device
def transformer(feat0, feat1, layer_structs, layer_names, nhead, mask0=None, mask1=None, inplace=False, prepared=None, mode=None,
                diag=None, skip_padded=False):
    """LocalFeatureTransformer.forward.  Returns new (feat0, feat1); inputs are not modified unless
    ``inplace`` (then, when feat0 / feat1 are the contiguous halves of one buffer, the layers run on
    that buffer directly instead of on a torch.cat copy of it).  ``mode``: see COARSE_MODE; ``diag``: uint8 tensor for
    loftr_transformer_fwd_planned's status word / per-item trace.  ``skip_padded`` (with masks): the caller does not read the
    features of padding tokens -- 128-token tiles without a valid token keep their input values (loftr_transformer_fwd_padded;
    per-call launches whatever ``mode`` says: the persistent form computes every tile)."""
    _need(feat0, "feat0"); _need(feat1, "feat1")
    N, L, Cc = feat0.shape
    S = feat1.shape[1]
    m0, m1 = _mask_u8(mask0, "mask0"), _mask_u8(mask1, "mask1")
    if L == S:          # stack -> the two self-attention calls of a layer run as one batch of 2N
        both = stacked_halves(feat0, feat1) if inplace and feat0.shape[0] == feat1.shape[0] else None
        if both is None:
            both = torch.cat([feat0, feat1], 0)
        f0, f1 = both[:N], both[N:]
        if m0 is not None:
            mb = torch.cat([m0, m1], 0)
            m0, m1 = mb[:N], mb[N:]
    else:
        f0, f1 = feat0.clone(), feat1.clone()
    n_layers = len(layer_names)
    arr = (LayerWeights * n_layers)(*layer_structs)
    kind_list = [{"self": 0, "cross": 1}[n] for n in layer_names]    # KeyError like the reference
    kinds = (C.c_int * n_layers)(*kind_list)
    lib = _lib.load()
    nbytes = lib.loftr_encoder_workspace_bytes(2 * N, L, S, Cc)
    ws = workspace(nbytes, feat0.device)
    mode = mode or COARSE_MODE
    skip_padded = bool(skip_padded) and m0 is not None
    if skip_padded:
        mode = "launches"
    if mode == "auto":
        mode = "persistent" if N >= COARSE_AUTO_MIN_PAIRS else "launches"
    plan = None
    if mode != "launches" and Cc == 256 and nhead == 8 and N > 0:
        order = 1 if mode == "persistent_call_order" else 0
        plan = coarse_plan(kind_list, N, L, S, feat0.device, order)
    if plan is not None:
        check(lib.loftr_transformer_fwd_planned(_ptr(f0), _ptr(f1), _ptr(m0), _ptr(m1), arr, kinds, n_layers, N, L, S, Cc, nhead,
                                                _ptr(prepared), prepared.numel() if prepared is not None else 0,
                                                _ptr(ws), ws.numel(), _ptr(plan), plan.numel(), order,
                                                _ptr(diag), diag.numel() if diag is not None else 0, _stream()),
              "loftr_transformer_fwd_planned")
    elif skip_padded:
        check(lib.loftr_transformer_fwd_padded(_ptr(f0), _ptr(f1), _ptr(m0), _ptr(m1), arr, kinds, n_layers, N, L, S, Cc, nhead,
                                               _ptr(prepared), prepared.numel() if prepared is not None else 0,
                                               _ptr(ws), ws.numel(), 1, _stream()), "loftr_transformer_fwd_padded")
    else:
        check(lib.loftr_transformer_fwd(_ptr(f0), _ptr(f1), _ptr(m0), _ptr(m1), arr, kinds, n_layers, N, L, S, Cc, nhead,
                                        _ptr(prepared), prepared.numel() if prepared is not None else 0,
                                        _ptr(ws), ws.numel(), _stream()), "loftr_transformer_fwd")
    return f0, f1


@_on_device
def coarse_match(feat_c0, feat_c1, hw0_c, hw1_c, thr, border_rm, scale, match_type="dual_softmax",
                 temperature=0.1, bin_score=None, skh_iters=3, skh_prefilter=False, mask0=None, mask1=None,
                 scale0=None, scale1=None, want_conf=True, want_assign=False):
    """CoarseMatching.forward (eval).  Returns dict(conf_matrix, [conf_matrix_with_bin], b_ids, i_ids,
    j_ids, mconf, mkpts0_c, mkpts1_c, counts).  One host sync (the match count), like torch.where
    in the reference (coarse_matching.py:194)."""
    _need(feat_c0, "feat_c0"); _need(feat_c1, "feat_c1")
    N, L, Cc = feat_c0.shape
    S = feat_c1.shape[1]
    dev = feat_c0.device
    assert L == hw0_c[0] * hw0_c[1] and S == hw1_c[0] * hw1_c[1]
    m0, m1 = _mask_u8(mask0, "mask0"), _mask_u8(mask1, "mask1")
    s0 = None if scale0 is None else _need(scale0.to(torch.float32).contiguous(), "scale0")
    s1 = None if scale1 is None else _need(scale1.to(torch.float32).contiguous(), "scale1")
    cap = max(N * L, 1)
    b_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    i_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    j_ids = torch.empty(cap, dtype=torch.int64, device=dev)
    mconf = torch.empty(cap, dtype=torch.float32, device=dev)
    mk0 = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    mk1 = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    counts = torch.zeros(1 + N, dtype=torch.int32, device=dev)
    p = CoarseParams(N, hw0_c[0], hw0_c[1], hw1_c[0], hw1_c[1], Cc, float(thr), int(border_rm), float(scale),
                     m0.data_ptr() if m0 is not None else None, m1.data_ptr() if m1 is not None else None,
                     s0.data_ptr() if s0 is not None else None, s1.data_ptr() if s1 is not None else None)
    mo = MatchOut(b_ids.data_ptr(), i_ids.data_ptr(), j_ids.data_ptr(), mconf.data_ptr(), mk0.data_ptr(),
                  mk1.data_ptr(), counts.data_ptr())
    lib = _lib.load()
    ws = workspace(lib.loftr_coarse_match_workspace_bytes(N, L, S, Cc), dev)
    out = {}
    if match_type == "dual_softmax":
        conf = torch.empty(N, L, S, device=dev, dtype=torch.float32) if want_conf else None
        check(lib.loftr_coarse_match_dual_softmax(_ptr(feat_c0), _ptr(feat_c1), C.byref(p), float(temperature),
                                                  _ptr(conf), C.byref(mo), _ptr(ws), ws.numel(), _stream()),
              "loftr_coarse_match_dual_softmax")
    elif match_type == "sinkhorn":
        conf = torch.empty(N, L, S, device=dev, dtype=torch.float32)
        assign = torch.empty(N, L + 1, S + 1, device=dev, dtype=torch.float32) if want_assign else None
        check(lib.loftr_coarse_match_sinkhorn(_ptr(feat_c0), _ptr(feat_c1), C.byref(p), float(bin_score),
                                              int(skh_iters), int(bool(skh_prefilter)), _ptr(conf), _ptr(assign),
                                              C.byref(mo), _ptr(ws), ws.numel(), _stream()),
              "loftr_coarse_match_sinkhorn")
        if want_assign:
            out["conf_matrix_with_bin"] = assign
    else:
        raise NotImplementedError(match_type)
    M = int(counts[0].item()) if N > 0 else 0            # the one D2H sync of the path
    out.update(conf_matrix=conf, b_ids=b_ids[:M], i_ids=i_ids[:M], j_ids=j_ids[:M], mconf=mconf[:M],
               mkpts0_c=mk0[:M], mkpts1_c=mk1[:M], counts=counts)
    return out


def _fmap(t):
    """[N,C,H,W] tensor with arbitrary (e.g. channels-last) strides -> FMap."""
    sn, sc, sh, sw = t.stride()
    return FMap(t.data_ptr(), sn, sc, sh, sw, t.shape[2], t.shape[3])


@_on_device
def fine_preprocess(feat_f0, feat_f1, feat_c0, feat_c1, b_ids, i_ids, j_ids, hw0_c, hw1_c, W, stride,
                    down_w=None, down_b=None, merge_w=None, merge_b=None):
    """FinePreprocess.forward for M > 0.  Returns (feat_f0_unfold, feat_f1_unfold) [M, W*W, Cf]."""
    for t, n in ((feat_f0, "feat_f0"), (feat_f1, "feat_f1")):
        if not t.is_cuda or t.dtype != torch.float32:
            raise _lib.LoftrHipError(f"{n}: expected a float32 GPU tensor")
    _need(feat_c0, "feat_c0"); _need(feat_c1, "feat_c1")
    M = b_ids.shape[0]
    Cf = feat_f0.shape[1]
    dev = feat_f0.device
    out = torch.empty(2 * M, W * W, Cf, device=dev, dtype=torch.float32)     # one buffer: the fine transformer
    out0, out1 = out[:M], out[M:]                                            # runs on it in place (no torch.cat)
    if M == 0:
        return out0, out1
    lib = _lib.load()
    ws = workspace(lib.loftr_fine_preprocess_workspace_bytes(M, W, Cf), dev)
    f0, f1 = _fmap(feat_f0), _fmap(feat_f1)
    check(lib.loftr_fine_preprocess(C.byref(f0), C.byref(f1), _ptr(feat_c0), _ptr(feat_c1), feat_c0.shape[1],
                                    feat_c1.shape[1], feat_c0.shape[2], _ptr(_need(b_ids, "b_ids", torch.int64)),
                                    _ptr(_need(i_ids, "i_ids", torch.int64)), _ptr(_need(j_ids, "j_ids", torch.int64)),
                                    M, hw0_c[1], hw1_c[1], int(stride), int(W), Cf, _ptr(down_w), _ptr(down_b),
                                    _ptr(merge_w), _ptr(merge_b), _ptr(out0), _ptr(out1), _ptr(ws), ws.numel(),
                                    _stream()), "loftr_fine_preprocess")
    return out0, out1


@_on_device
def fine_preprocess_bwd(feat_f0, feat_f1, feat_c0, feat_c1, b_ids, i_ids, j_ids, hw0_c, hw1_c, W, stride, down_w, down_b, merge_w,
                        grad_out0, grad_out1):
    """Backward of fine_preprocess (fine_preprocess.py:29-59 under autograd).  Returns (grad_feat_f0, grad_feat_f1 [laid out like the
    inputs], grad_feat_c0, grad_feat_c1, grad_down_w, grad_down_b, grad_merge_w, grad_merge_b)."""
    _need(feat_c0, "feat_c0"); _need(feat_c1, "feat_c1"); _need(grad_out0, "grad_out0"); _need(grad_out1, "grad_out1")
    M = b_ids.shape[0]
    Cf, Cc = feat_f0.shape[1], feat_c0.shape[2]
    dev = feat_f0.device
    gf0, gf1 = torch.zeros_like(feat_f0), torch.zeros_like(feat_f1)             # (preserve_format: same strides as the inputs)
    gc0, gc1 = torch.zeros_like(feat_c0), torch.zeros_like(feat_c1)
    gdw, gdb = torch.zeros_like(down_w), torch.zeros_like(down_b)
    gmw, gmb = torch.zeros(Cf, 2 * Cf, device=dev), torch.zeros(Cf, device=dev)
    if M == 0:
        return gf0, gf1, gc0, gc1, gdw, gdb, gmw, gmb
    lib = _lib.load()
    ws = workspace(lib.loftr_fine_preprocess_bwd_workspace_bytes(M, int(W), Cf, Cc), dev)
    f0, f1, g0, g1 = _fmap(feat_f0), _fmap(feat_f1), _fmap(gf0), _fmap(gf1)
    check(lib.loftr_fine_preprocess_bwd(C.byref(f0), C.byref(f1), _ptr(feat_c0), _ptr(feat_c1), feat_c0.shape[1], feat_c1.shape[1], Cc,
                                        _ptr(_need(b_ids, "b_ids", torch.int64)), _ptr(_need(i_ids, "i_ids", torch.int64)),
                                        _ptr(_need(j_ids, "j_ids", torch.int64)), M, hw0_c[1], hw1_c[1], int(stride), int(W), Cf,
                                        _ptr(_need(down_w, "down_w")), _ptr(_need(down_b, "down_b")), _ptr(_need(merge_w, "merge_w")),
                                        _ptr(grad_out0), _ptr(grad_out1), C.byref(g0), C.byref(g1), _ptr(gc0), _ptr(gc1), _ptr(gdw), _ptr(gdb),
                                        _ptr(gmw), _ptr(gmb), _ptr(ws), ws.numel(), _stream()), "loftr_fine_preprocess_bwd")
    return gf0, gf1, gc0, gc1, gdw, gdb, gmw, gmb


@_on_device
def fine_match(feat_f0, feat_f1, mkpts1_c, b_ids, scale, scale1=None):
    """FineMatching for M > 0.  Returns (expec_f [M,3], mkpts1_f [M,2])."""
    _need(feat_f0, "feat_f0"); _need(feat_f1, "feat_f1")
    M, WW, Cf = feat_f0.shape
    dev = feat_f0.device
    expec = torch.empty(M, 3, device=dev, dtype=torch.float32)
    mk1f = torch.empty(M, 2, device=dev, dtype=torch.float32)
    s1 = None if scale1 is None else _need(scale1.to(torch.float32).contiguous(), "scale1")
    check(_lib.load().loftr_fine_match(_ptr(feat_f0), _ptr(feat_f1), M, WW, Cf,
                                       _ptr(_need(mkpts1_c.contiguous(), "mkpts1_c")),
                                       _ptr(_need(b_ids, "b_ids", torch.int64)), float(scale), _ptr(s1), _ptr(expec),
                                       _ptr(mk1f), _stream()), "loftr_fine_match")
    return expec, mk1f


# ---------------------------------------------------------------------------------------------
# Backward of the two matching heads (include/loftr_hip.h: "backward of the matching heads"); the autograd.Function
# wrappers that call these live in loftr_amd/autograd.py.
@_on_device
def dual_softmax_bwd(feat_c0, feat_c1, grad_conf, hw0_c, hw1_c, temperature, mask0=None, mask1=None):
    """dL/d sim_matrix [N,L,S] from dL/d conf_matrix (coarse_matching.py:110-119)."""
    _need(feat_c0, "feat_c0"); _need(feat_c1, "feat_c1"); _need(grad_conf, "grad_conf")
    N, L, Cc = feat_c0.shape
    S = feat_c1.shape[1]
    assert tuple(grad_conf.shape) == (N, L, S) and L == hw0_c[0] * hw0_c[1] and S == hw1_c[0] * hw1_c[1]
    m0, m1 = _mask_u8(mask0, "mask0"), _mask_u8(mask1, "mask1")
    p = CoarseParams(N, hw0_c[0], hw0_c[1], hw1_c[0], hw1_c[1], Cc, 0.0, 0, 1.0,
                     m0.data_ptr() if m0 is not None else None, m1.data_ptr() if m1 is not None else None, None, None)
    dsim = torch.empty(N, L, S, device=feat_c0.device, dtype=torch.float32)
    lib = _lib.load()
    ws = workspace(lib.loftr_coarse_match_workspace_bytes(N, L, S, Cc), feat_c0.device)
    check(lib.loftr_dual_softmax_bwd(_ptr(feat_c0), _ptr(feat_c1), C.byref(p), float(temperature), _ptr(grad_conf), _ptr(dsim),
                                     _ptr(ws), ws.numel(), _stream()), "loftr_dual_softmax_bwd")
    return dsim


@_on_device
def head_feat_grads(dsim, feat_c0, feat_c1, alpha, want0=True, want1=True):
    """(alpha * dsim @ feat_c1, alpha * dsim^T @ feat_c0): the einsum of coarse_matching.py:110-114 / :122-123 in reverse.
    dsim [N,L,S] fp32, possibly a strided view (last stride 1): the interior of the Sinkhorn head's [N,L+1,S+1] gradient."""
    _need(feat_c0, "feat_c0"); _need(feat_c1, "feat_c1")
    N, L, Cc = feat_c0.shape
    S = feat_c1.shape[1]
    if not (dsim.is_cuda and dsim.dtype == torch.float32 and tuple(dsim.shape) == (N, L, S) and dsim.stride(2) == 1):
        raise _lib.LoftrHipError("head_feat_grads: dsim must be a float32 GPU tensor [N,L,S] with unit last stride")
    g0 = torch.empty_like(feat_c0) if want0 else None
    g1 = torch.empty_like(feat_c1) if want1 else None
    if want0 or want1:
        check(_lib.load().loftr_head_feat_grads(_ptr(dsim), dsim.stride(1), dsim.stride(0), _ptr(feat_c0), _ptr(feat_c1), N, L, S, Cc,
                                                float(alpha), _ptr(g0), _ptr(g1), _stream()), "loftr_head_feat_grads")
    return g0, g1


@_on_device
def sinkhorn_bwd(feat_c0, feat_c1, grad_assign, hw0_c, hw1_c, bin_score, iters, mask0=None, mask1=None):
    """(dL/d couplings [N,L+1,S+1], dL/d bin_score [1]) from dL/d conf_matrix_with_bin (coarse_matching.py:121-143)."""
    _need(feat_c0, "feat_c0"); _need(feat_c1, "feat_c1"); _need(grad_assign, "grad_assign")
    N, L, Cc = feat_c0.shape
    S = feat_c1.shape[1]
    assert tuple(grad_assign.shape) == (N, L + 1, S + 1) and L == hw0_c[0] * hw0_c[1] and S == hw1_c[0] * hw1_c[1]
    m0, m1 = _mask_u8(mask0, "mask0"), _mask_u8(mask1, "mask1")
    p = CoarseParams(N, hw0_c[0], hw0_c[1], hw1_c[0], hw1_c[1], Cc, 0.0, 0, 1.0,
                     m0.data_ptr() if m0 is not None else None, m1.data_ptr() if m1 is not None else None, None, None)
    dev = feat_c0.device
    dz = torch.empty(N, L + 1, S + 1, device=dev, dtype=torch.float32)
    z = torch.empty(N, L, S, device=dev, dtype=torch.float32)
    dbin = torch.zeros(1, device=dev, dtype=torch.float32)
    lib = _lib.load()
    ws = workspace(lib.loftr_sinkhorn_bwd_workspace_bytes(N, L, S, Cc, int(iters)), dev)
    check(lib.loftr_sinkhorn_bwd(_ptr(feat_c0), _ptr(feat_c1), C.byref(p), float(bin_score), int(iters), _ptr(grad_assign), _ptr(z), _ptr(dz),
                                 _ptr(dbin), _ptr(ws), ws.numel(), _stream()), "loftr_sinkhorn_bwd")
    return dz, dbin


@_on_device
def fine_match_bwd(feat_f0, feat_f1, grad_expec):
    """(dL/d feat_f0, dL/d feat_f1) [M,WW,C] from dL/d expec_f [M,3] (fine_matching.py:43-57)."""
    _need(feat_f0, "feat_f0"); _need(feat_f1, "feat_f1"); _need(grad_expec, "grad_expec")
    M, WW, Cf = feat_f0.shape
    assert tuple(grad_expec.shape) == (M, 3)
    g0, g1 = torch.empty_like(feat_f0), torch.empty_like(feat_f1)
    check(_lib.load().loftr_fine_match_bwd(_ptr(feat_f0), _ptr(feat_f1), M, WW, Cf, _ptr(grad_expec), _ptr(g0), _ptr(g1), _stream()),
          "loftr_fine_match_bwd")
    return g0, g1


# ---------------------------------------------------------------------------------------------
# ResNet-FPN building blocks.  An SP activation is carried as (tensor int32 [B,H,W,Cp], C).
def ceil32(c):
    return (c + 31) // 32 * 32


@_on_device
def sp_from_nhwc(x_nhwc, scaled=False):
    """fp32 [B,H,W,C] contiguous -> SP int32 [B,H,W,ceil32(C)].
    scaled=True: the tensor is stored times the power of two that lifts its maximum to [2^13, 2^14) (csrc/gemm.h:
    full split-fp16 precision whatever the tensor's magnitude); returns (sp, inv_scale) with inv_scale a device float
    to hand to conv_bn_act(x_inv_scale=...)."""
    _need(x_nhwc, "x_nhwc")
    B, H, W, Cc = x_nhwc.shape
    out = torch.empty(B, H, W, ceil32(Cc), dtype=torch.int32, device=x_nhwc.device)
    if scaled:
        inv = torch.empty(1, dtype=torch.float32, device=x_nhwc.device)
        check(_lib.load().loftr_sp_from_f32_scaled(_ptr(x_nhwc), _ptr(out), B * H * W, Cc, _ptr(inv), _stream()),
              "loftr_sp_from_f32_scaled")
        return out, inv
    check(_lib.load().loftr_sp_from_f32(_ptr(x_nhwc), _ptr(out), B * H * W, Cc, _stream()), "loftr_sp_from_f32")
    return out


@_on_device
def sp_to_nhwc(x_sp, Cc):
    """SP int32 [B,H,W,Cp] -> fp32 [B,H,W,C]."""
    B, H, W, _ = x_sp.shape
    out = torch.empty(B, H, W, Cc, dtype=torch.float32, device=x_sp.device)
    check(_lib.load().loftr_sp_to_f32(_ptr(x_sp), _ptr(out), B * H * W, Cc, _stream()), "loftr_sp_to_f32")
    return out


def _prepared_conv(conv, bn):
    """Folded-BN SP filter of (conv, bn), cached on the module and rebuilt when any of the tensors it was built
    from is modified in place (tensor._version), replaced (data_ptr) or moved.  Inference weights are constant: the
    per-call weight preparation of loftr_conv_bn_act (two launches + a memset per convolution) runs once."""
    w = conv.weight
    if not w.is_cuda or w.dtype != torch.float32:
        raise _lib.LoftrHipError("conv.weight: expected a float32 GPU tensor")
    if bn is not None and bn.training:
        raise _lib.LoftrHipError("conv_bn_act folds eval-mode BatchNorm only; call .eval()")
    assert conv.bias is None and conv.dilation == (1, 1) and conv.groups == 1
    bnp = [] if bn is None else [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    key = tuple((t.data_ptr(), t._version, tuple(t.stride())) for t in [w] + bnp) + (None if bn is None else float(bn.eps), str(w.device))
    cached = _PREPARED.get(conv)
    # ... or replaced by a NEW tensor object the allocator put at the same address (weak references to the originals)
    if cached is not None and cached[0] == key and all(r() is t for r, t in zip(cached[2], [w] + bnp)):
        return cached[1]
    Cout, Cin, KH, KW = w.shape
    lib = _lib.load()
    buf = torch.empty(lib.loftr_conv_workspace_bytes(Cin, Cout, KH, KW), dtype=torch.uint8, device=w.device)
    wst = (C.c_long * 4)(*w.stride())                       # contiguous or channels-last storage
    ptrs = [_ptr(t) for t in bnp] if bnp else [None] * 4
    check(lib.loftr_conv_prepare(_ptr(w), wst, Cin, Cout, KH, KW, *ptrs, float(bn.eps) if bn is not None else 0.0, _ptr(buf),
                                 buf.numel(), _stream()), "loftr_conv_prepare")
    _PREPARED[conv] = (key, buf, [weakref.ref(t) for t in [w] + bnp])
    return buf


CONV_SHARED_GPU = 0x100      # include/loftr_hip.h: LOFTR_CONV_SHARED_GPU


CONV_REM = False      # True: conv_bn_act hands 193 .. 199-channel 3x3 layers a scratch buffer (the tap-decomposed remainder form, see below)


@_on_device
def conv_bn_act(x_sp, Cin, conv, bn=None, act=0, residual=None, want_sp=True, want_f32=False, low_sp=None, shared_gpu=False,
                x_inv_scale=None):
    """nn.Conv2d(bias=False) [+ eval BatchNorm2d] [+ residual] [+ act] on an SP activation.

    x_sp int32 [B,H,W,ceil32(Cin)]; returns (y_sp or None, y_f32 [B,Ho,Wo,Cout] or None).
    low_sp (FPN top-down step): y = conv1x1(x) + bilinear_x2(low_sp), SP in / out.
    shared_gpu: the launch runs next to another stream's work (no persistent workgroups, see loftr_hip.h)."""
    w = conv.weight
    Cout, Cin_w, KH, KW = w.shape
    assert Cin_w == Cin
    stride, pad = conv.stride[0], conv.padding[0]
    B, H, W, Cp = x_sp.shape
    assert Cp == ceil32(Cin) and x_sp.dtype == torch.int32 and x_sp.is_contiguous()
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    dev = x_sp.device
    prepared = _prepared_conv(conv, bn)
    y_sp = torch.empty(B, Ho, Wo, ceil32(Cout), dtype=torch.int32, device=dev) if want_sp else None
    y_f32 = torch.empty(B, Ho, Wo, Cout, dtype=torch.float32, device=dev) if want_f32 else None
    lib = _lib.load()
    # 193 .. 199 output channels (LoFTR's 196): the channels beyond 192 as a tap-decomposed product through a scratch buffer (include/loftr_hip.h);
    # a fresh tensor per call, not the cached workspace: two streams may run convolutions at the same time
    # MEASURED AND NOT ADOPTED (CONV_REM = False): the 192-column kernel is no faster than the 224-column one -- a step of these kernels is bound by
    # its weight stream and barrier, not by its MFMA count (profiles/r06_conv_rem_ab.txt: 2586 + 217 us against 2370 us at 1/2 resolution)
    nscr = lib.loftr_conv_scratch_bytes(B, H, W, Cout, KH, KW, stride) if (CONV_REM and low_sp is None and not want_f32) else 0
    scratch = torch.empty(nscr, dtype=torch.uint8, device=dev) if nscr else None
    check(lib.loftr_conv_bn_act_prepared_scratch(_ptr(x_sp), B, H, W, Cin, _ptr(prepared), prepared.numel(), Cout, KH, KW,
                                                 stride, pad, int(act) | (CONV_SHARED_GPU if shared_gpu else 0), _ptr(residual),
                                                 _ptr(low_sp), _ptr(y_sp), _ptr(y_f32), _ptr(x_inv_scale), _ptr(scratch), nscr,
                                                 _stream()), "loftr_conv_bn_act_prepared_scratch")
    return y_sp, y_f32


@_on_device
def conv_raw(x_sp, Cin, weight, stride, pad, x_inv_scale=None):
    """Bias-free convolution of an SP activation with a filter given as a TENSOR [Cout,Cin,KH,KW] (no BatchNorm folded, no
    activation, filter encoded per call): the training forward and the input-gradient convolutions.  -> fp32 [B,Ho,Wo,Cout]."""
    _need(weight, "weight")
    Cout, Cin_w, KH, KW = weight.shape
    assert Cin_w == Cin
    B, H, W, Cp = x_sp.shape
    assert Cp == ceil32(Cin) and x_sp.dtype == torch.int32 and x_sp.is_contiguous()
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    lib = _lib.load()
    ws = workspace(lib.loftr_conv_workspace_bytes(Cin, Cout, KH, KW), x_sp.device)
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.float32, device=x_sp.device)
    wst = (C.c_long * 4)(*weight.stride())
    check(lib.loftr_conv_bn_act(_ptr(x_sp), B, H, W, Cin, _ptr(weight), wst, Cout, KH, KW, stride, pad, None, None, None, None, 0.0, 0,
                                None, None, _ptr(y), _ptr(ws), ws.numel(), _ptr(x_inv_scale), _stream()), "loftr_conv_bn_act")
    return y


@_on_device
def conv_wgrad(dy_nhwc, x_nhwc, KH, KW, stride, pad):
    """dL/dweight [Cout,Cin,KH,KW] of a bias-free convolution from dy [B,Ho,Wo,Cout] and its input x [B,H,W,Cin] (fp32, channels last)."""
    _need(dy_nhwc, "dy_nhwc"); _need(x_nhwc, "x_nhwc")
    B, H, W, Cin = x_nhwc.shape
    _, Ho, Wo, Cout = dy_nhwc.shape
    assert Ho == (H + 2 * pad - KH) // stride + 1 and Wo == (W + 2 * pad - KW) // stride + 1 and dy_nhwc.shape[0] == B
    lib = _lib.load()
    ws = workspace(lib.loftr_conv_wgrad_workspace_bytes(B, Ho, Wo, Cin, Cout, KH, KW), x_nhwc.device)
    taps = torch.empty(KH * KW, Cout, Cin, dtype=torch.float32, device=x_nhwc.device)
    check(lib.loftr_conv_wgrad(_ptr(dy_nhwc), _ptr(x_nhwc), B, H, W, Cin, Cout, KH, KW, stride, pad, _ptr(taps), _ptr(ws), ws.numel(),
                               _stream()), "loftr_conv_wgrad")
    return taps.view(KH, KW, Cout, Cin).permute(2, 3, 0, 1).contiguous()


@_on_device
def stem_conv_bn_relu(x, conv, bn):
    """conv1 (7x7, stride 2, one input channel) + eval bn1 + relu -> SP int32 [B,Ho,Wo,ceil32(C0)]."""
    if not x.is_cuda or x.dtype != torch.float32 or x.shape[1] != 1:
        raise _lib.LoftrHipError("stem: expected a float32 GPU tensor [B,1,H,W]")
    w = conv.weight
    assert tuple(w.shape[1:]) == (1, 7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.bias is None
    if bn.training:
        raise _lib.LoftrHipError("stem folds eval-mode BatchNorm only; call .eval()")
    B, _, H, W = x.shape
    C0 = w.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, ceil32(C0), dtype=torch.int32, device=x.device)
    xs, wst = (C.c_long * 4)(*x.stride()), (C.c_long * 4)(*w.stride())
    check(_lib.load().loftr_stem_conv_bn_relu(_ptr(x), xs, B, H, W, _ptr(w), wst, C0, _ptr(bn.weight), _ptr(bn.bias),
                                              _ptr(bn.running_mean), _ptr(bn.running_var), float(bn.eps), _ptr(y),
                                              _stream()), "loftr_stem_conv_bn_relu")
    return y


def conv1x1_upsample_add(x_sp, Cin, conv, low_sp):
    """conv1x1(x) + bilinear x2 (align_corners=True) of low in one launch (one FPN top-down step); SP in / out."""
    Cout = conv.weight.shape[0]
    assert tuple(conv.weight.shape[1:]) == (Cin, 1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
    B, H, W, _ = x_sp.shape
    if low_sp.shape != (B, H // 2, W // 2, ceil32(Cout)) or H % 2 or W % 2 or not low_sp.is_contiguous():
        raise _lib.LoftrHipError(f"conv1x1_upsample_add: low map {tuple(low_sp.shape)} is not half of {tuple(x_sp.shape)}")
    return conv_bn_act(x_sp, Cin, conv, low_sp=low_sp)[0]


@_on_device
def upsample2x_add(low_sp, lateral_sp, Cc):
    """lateral + bilinear x2 (align_corners=True) of low; SP in, SP out."""
    B, Hl, Wl, Cp = low_sp.shape
    assert lateral_sp.shape == (B, 2 * Hl, 2 * Wl, Cp)
    out = torch.empty_like(lateral_sp)
    check(_lib.load().loftr_upsample2x_add(_ptr(low_sp), _ptr(lateral_sp), _ptr(out), B, Hl, Wl, Cc, _stream()),
          "loftr_upsample2x_add")
    return out


@_on_device
def epipolar_errors(mkpts0_f, mkpts1_f, m_bids, T_0to1, K0, K1):
    """Squared symmetric epipolar distance of every match (metrics.py:31-68) -> float32 [M], match order."""
    for name, t, dt in (("mkpts0_f", mkpts0_f, torch.float32), ("mkpts1_f", mkpts1_f, torch.float32), ("m_bids", m_bids, torch.int64),
                        ("T_0to1", T_0to1, torch.float32), ("K0", K0, torch.float32), ("K1", K1, torch.float32)):
        if not t.is_cuda or t.dtype != dt:
            raise _lib.LoftrHipError(f"{name}: expected a {dt} GPU tensor (the evaluation kernels have no CPU fallback)")
    M, N = mkpts0_f.shape[0], T_0to1.shape[0]
    assert mkpts0_f.shape == (M, 2) and mkpts1_f.shape == (M, 2) and m_bids.shape == (M,)
    assert T_0to1.shape == (N, 4, 4) and K0.shape == (N, 3, 3) and K1.shape == (N, 3, 3)
    out = torch.empty(M, dtype=torch.float32, device=mkpts0_f.device)
    args = [t.contiguous() for t in (mkpts0_f, mkpts1_f, m_bids, T_0to1, K0, K1)]
    check(_lib.load().loftr_epipolar_errors(*[_ptr(t) for t in args], M, N, _ptr(out), _stream()), "loftr_epipolar_errors")
    return out


# ---- training-mode glue of the backbone (csrc/train_glue.hip; resnet_fpn.py:22-40,66-77,110-116) ------------------------------------------
def _dense4(t, name):
    """A 4-D fp32 GPU tensor [N,C,H,W] stored densely either NCHW or NHWC (channels_last: what the convolution nodes produce); returns
    (tensor as given or made contiguous, channels_last flag)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
        raise _lib.LoftrHipError(f"{name}: expected a 4-D float32 GPU tensor")
    if t.data_ptr() % 16:                                        # (a view into the middle of a buffer: the kernels use 16-byte accesses)
        t = t.clone(memory_format=torch.preserve_format)
    if t.is_contiguous():
        return t, 0
    if t.is_contiguous(memory_format=torch.channels_last) and t.shape[1] % 4 == 0 and t.shape[1] <= 1024:
        return t, 1
    return t.contiguous(), 0


def _like_layout(t, cl):
    """t in the layout `cl` names (a copy only when it is not already there, or not 16-byte aligned)."""
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.preserve_format)
    if cl:
        return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)
    return t.contiguous()


@_on_device
def bn_train_fwd(x, gamma, beta, eps):
    """nn.BatchNorm2d in .train() mode on an [N,C,H,W] fp32 tensor (NCHW or channels-last storage, kept): (y, mean [C], invstd [C], unbiased
    variance [C])."""
    x, cl = _dense4(x, "x")
    N, Cc, H, W = x.shape
    lib = _lib.load()
    y = torch.empty_like(x)                                      # preserves the memory format
    mean, invstd, varu = (torch.empty(Cc, dtype=torch.float32, device=x.device) for _ in range(3))
    ws = workspace(lib.loftr_bn_train_workspace_bytes(N, Cc, H * W), x.device)
    check(lib.loftr_bn_train_fwd(_ptr(x), N, Cc, H * W, cl, _ptr(gamma), _ptr(beta), float(eps), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(varu),
                                 _ptr(ws), ws.numel(), _stream()), "loftr_bn_train_fwd")
    return y, mean, invstd, varu


@_on_device
def bn_train_bwd(dy, x, mean, invstd, gamma):
    """(dx, dgamma, dbeta) of bn_train_fwd; dx in x's layout."""
    x, cl = _dense4(x, "x")
    dy = _like_layout(dy, cl)
    N, Cc, H, W = x.shape
    lib = _lib.load()
    dx = torch.empty_like(x)
    dgamma, dbeta = (torch.empty(Cc, dtype=torch.float32, device=x.device) for _ in range(2))
    ws = workspace(lib.loftr_bn_train_workspace_bytes(N, Cc, H * W), x.device)
    check(lib.loftr_bn_train_bwd(_ptr(dy), _ptr(x), N, Cc, H * W, cl, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                 _ptr(ws), ws.numel(), _stream()), "loftr_bn_train_bwd")
    return dx, dgamma, dbeta


ACT_CODES = {"none": 0, "relu": 1, "leaky_relu": 2}


@_on_device
def act_fwd(a, b, act, slope=0.01):
    """act(a + b) (b may be None), act in ACT_CODES; elementwise: any dense layout, the result takes a's."""
    a, cl = _dense4(a, "a") if a.dim() == 4 else (_need(a, "a"), 0)
    if b is not None:
        assert b.shape == a.shape
        b = _like_layout(b, cl) if a.dim() == 4 else _need(b, "b")
    y = torch.empty_like(a)
    check(_lib.load().loftr_act_fwd(_ptr(a), _ptr(b), a.numel(), ACT_CODES[act], float(slope), _ptr(y), _stream()), "loftr_act_fwd")
    return y


@_on_device
def act_bwd(dy, y, act, slope=0.01):
    y, cl = _dense4(y, "y") if y.dim() == 4 else (_need(y, "y"), 0)
    dy = _like_layout(dy, cl) if y.dim() == 4 else _need(dy, "dy")
    dx = torch.empty_like(y)
    check(_lib.load().loftr_act_bwd(_ptr(dy), _ptr(y), dy.numel(), ACT_CODES[act], float(slope), _ptr(dx), _stream()), "loftr_act_bwd")
    return dx


@_on_device
def upsample2x_bilinear(x):
    """F.interpolate(x, scale_factor=2., mode='bilinear', align_corners=True) of an [N,C,H,W] fp32 tensor, in x's memory format."""
    x, cl = _dense4(x, "x")
    N, Cc, H, W = x.shape
    y = torch.empty(N, Cc, 2 * H, 2 * W, dtype=torch.float32, device=x.device, memory_format=torch.channels_last if cl else torch.contiguous_format)
    check(_lib.load().loftr_upsample2x_bilinear_fwd(_ptr(x), N, Cc, H, W, cl, _ptr(y), _stream()), "loftr_upsample2x_bilinear_fwd")
    return y


@_on_device
def upsample2x_bilinear_bwd(dy):
    dy, cl = _dense4(dy, "dy")
    N, Cc, Ho, Wo = dy.shape
    dx = torch.empty(N, Cc, Ho // 2, Wo // 2, dtype=torch.float32, device=dy.device, memory_format=torch.channels_last if cl else torch.contiguous_format)
    check(_lib.load().loftr_upsample2x_bilinear_bwd(_ptr(dy), N, Cc, Ho // 2, Wo // 2, cl, _ptr(dx), _stream()), "loftr_upsample2x_bilinear_bwd")
    return dx
