"""Training-side consumers of the matching path's outputs: forward values, and the gradients of
the two losses with respect to the heads' outputs.

Host-side mirror of the reference's interface: ``compute_supervision_coarse(data, config)`` /
``compute_supervision_fine(data, config)`` (src/loftr/utils/supervision.py:110-151) and ``LoFTRLoss(config)(data)``
(src/losses/loftr_loss.py:7-192) mutate the batch dict with the same keys.  The arithmetic runs in csrc/train.hip
behind the C-ABI (loftr_spvs_coarse, loftr_spvs_fine, loftr_coarse_loss_sums, loftr_fine_loss_sums); there is no CPU
fallback.  Backward: when conf_matrix / expec_f carry an autograd graph (loftr_amd/autograd.py: the dual-softmax and
FineMatching heads, and the Sinkhorn head with its bin_score parameter), LoFTRLoss is differentiable through
loftr_coarse_loss_grad / loftr_fine_loss_grad.  With ``LoFTR.head_grads`` the chain ends at the heads' inputs; with
``LoFTR.full_grads`` it runs through the whole model (autograd.py: transformers, FinePreprocess, position encoding and the
backbone's convolutions), and ``trainval_inference`` below is the reference's training step.  The RNG-dependent
ground-truth padding of CoarseMatching's training branch (coarse_matching.py:200-236) lives in
loftr_amd/loftr.py:CoarseMatching._train_sample.

Signature deviation: ``LoFTRLoss.compute_coarse_loss(conf, data)`` takes the batch dict (ground-truth id lists,
padding masks) where the reference takes ``(conf, conf_gt, weight=None)`` -- the dense conf_matrix_gt / weight volumes are
never needed here.  Value deviation: with no ground truth AND padding masks the reference zeroes weight[0, 0, 0], which also
removes that one cell from the NEGATIVE term; here it stays in (1 / (N L S) relative, below the test tolerance)."""
import ctypes as C

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from .ops import _need, _ptr, _stream, _mask_u8, workspace, _on_device


def _cfg(config, *keys):
    """config['LOFTR']['RESOLUTION'] of a yacs node, or the lower-case dict layout."""
    node = config
    for k in keys:
        node = node[k] if k in node else node[k.lower()]
    return node


@_on_device
def spvs_coarse(data, config, want_conf_gt=True):
    """supervision.py:22-109.  Update: conf_matrix_gt (optional), spv_b_ids / spv_i_ids / spv_j_ids, spv_w_pt0_i, spv_pt1_i."""
    img0, img1 = data["image0"], data["image1"]
    dev = img0.device
    N, _, H0, W0 = img0.shape
    _, _, H1, W1 = img1.shape
    scale = int(_cfg(config, "LOFTR", "RESOLUTION")[0])
    h0, w0, h1, w1 = H0 // scale, W0 // scale, H1 // scale, W1 // scale
    L, S = h0 * w0, h1 * w1
    f = lambda k: _need(data[k].to(torch.float32).contiguous(), k)
    d0, d1, T01, T10, K0, K1 = f("depth0"), f("depth1"), f("T_0to1"), f("T_1to0"), f("K0"), f("K1")
    s0 = f("scale0") if "scale0" in data else None
    s1 = f("scale1") if "scale0" in data else None
    m0 = _mask_u8(data["mask0"].flatten(-2), "mask0") if "mask0" in data else None
    m1 = _mask_u8(data["mask1"].flatten(-2), "mask1") if "mask0" in data else None
    p = _lib.SpvsParams(N, H0, W0, H1, W1, scale, d0.shape[1], d0.shape[2], d1.shape[1], d1.shape[2],
                        *[t.data_ptr() if t is not None else None for t in (d0, d1, T01, T10, K0, K1, s0, s1, m0, m1)])
    w_pt0 = torch.empty(N, L, 2, device=dev)
    pt1 = torch.empty(N, S, 2, device=dev)
    ids = [torch.empty(N * L, dtype=torch.int64, device=dev) for _ in range(3)]
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    gt = torch.empty(N, L, S, device=dev) if want_conf_gt else None
    lib = _lib.load()
    ws = workspace(lib.loftr_spvs_coarse_workspace_bytes(N, L, S), dev)
    _lib.check(lib.loftr_spvs_coarse(C.byref(p), _ptr(w_pt0), _ptr(pt1), _ptr(ids[0]), _ptr(ids[1]), _ptr(ids[2]), _ptr(count),
                                     _ptr(gt), _ptr(ws), ws.numel(), _stream()), "loftr_spvs_coarse")
    M = int(count.item())
    if M == 0:      # supervision.py:94-99: "this won't affect fine-level loss calculation"
        b_ids = i_ids = j_ids = torch.zeros(1, dtype=torch.int64, device=dev)
    else:
        b_ids, i_ids, j_ids = (t[:M] for t in ids)
    data.update({"conf_matrix_gt": gt, "spv_b_ids": b_ids, "spv_i_ids": i_ids, "spv_j_ids": j_ids, "spv_w_pt0_i": w_pt0,
                 "spv_pt1_i": pt1, "_spv_count": M})


def compute_supervision_coarse(data, config, want_conf_gt=True):
    assert len(set(data["dataset_name"])) == 1, "Do not support mixed datasets training!"
    data_source = data["dataset_name"][0]
    if data_source.lower() in ["scannet", "megadepth"]:
        spvs_coarse(data, config, want_conf_gt)
    else:
        raise ValueError(f"Unknown data source: {data_source}")


@_on_device
def spvs_fine(data, config):
    """supervision.py:124-142.  Update: expec_f_gt [M,2]."""
    w_pt0, pt1 = data["spv_w_pt0_i"], data["spv_pt1_i"]
    scale = float(_cfg(config, "LOFTR", "RESOLUTION")[1])
    radius = float(_cfg(config, "LOFTR", "FINE_WINDOW_SIZE") // 2)
    b, i, j = (_need(data[k].contiguous(), k, torch.int64) for k in ("b_ids", "i_ids", "j_ids"))
    M = b.shape[0]
    out = torch.empty(M, 2, device=w_pt0.device)
    s1 = _need(data["scale1"].to(torch.float32).contiguous(), "scale1") if "scale0" in data else None
    _lib.check(_lib.load().loftr_spvs_fine(_ptr(_need(w_pt0, "spv_w_pt0_i")), _ptr(_need(pt1, "spv_pt1_i")), w_pt0.shape[1], pt1.shape[1],
                                           _ptr(b), _ptr(i), _ptr(j), M, scale, radius, _ptr(s1), _ptr(out), _stream()), "loftr_spvs_fine")
    data.update({"expec_f_gt": out})


def compute_supervision_fine(data, config):
    data_source = data["dataset_name"][0]
    if data_source.lower() in ["scannet", "megadepth"]:
        spvs_fine(data, config)
    else:
        raise NotImplementedError


def trainval_inference(matcher, loss, batch, config):
    """PL_LoFTR._trainval_inference (src/lightning/lightning_loftr.py:76-91) without the Lightning module around it: coarse
    supervision -> matcher -> fine supervision -> loss, all on the batch dict; returns ``batch['loss']`` (with its graph when the
    matcher is in ``.train()`` mode with ``full_grads`` / ``head_grads``: ``training_step`` is this + ``.backward()`` + the optimiser)."""
    compute_supervision_coarse(batch, config)
    matcher(batch)
    compute_supervision_fine(batch, config)
    loss(batch)
    return batch["loss"]


class LoFTRLoss(torch.nn.Module):
    """loftr_loss.py:7-192.  Differentiable with respect to conf_matrix / expec_f when those carry a graph."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.loss_config = config["loftr"]["loss"]
        self.match_type = config["loftr"]["match_coarse"]["match_type"]
        self.sparse_spvs = config["loftr"]["match_coarse"]["sparse_spvs"]
        self.correct_thr = self.loss_config["fine_correct_thr"]
        self.c_pos_w = self.loss_config["pos_weight"]
        self.c_neg_w = self.loss_config["neg_weight"]
        self.fine_type = self.loss_config["fine_type"]

    def _coarse_terms(self, conf, data):
        """Everything of compute_coarse_loss that does not depend on conf's VALUES: kind, ids, masks, the loss weights
        after the corner cases (:31-42)."""
        ctype = self.loss_config["coarse_type"]
        if ctype == "cross_entropy":
            assert not self.sparse_spvs, "Sparse Supervision for cross-entropy not implemented!"
            kind = 3
        elif ctype == "focal":
            kind = (1 if self.match_type == "sinkhorn" else 0) if self.sparse_spvs else 2
        else:
            raise ValueError("Unknown coarse loss: {type}".format(type=ctype))
        bins = kind == 1
        N, L, S = conf.shape[0], conf.shape[1] - bins, conf.shape[2] - bins
        M = int(data.get("_spv_count", data["spv_b_ids"].shape[0]))
        if "_spv_count" not in data and M == 1 and int(data["spv_i_ids"][0]) == 0:
            M = 0                     # the reference's placeholder (0, 0, 0) of a pair without ground truth (cell 0 is never supervised)
        b, i, j = (_need(data[k].contiguous(), k, torch.int64) for k in ("spv_b_ids", "spv_i_ids", "spv_j_ids"))
        m0 = _mask_u8(data["mask0"].flatten(-2), "mask0") if "mask0" in data else None
        m1 = _mask_u8(data["mask1"].flatten(-2), "mask1") if "mask0" in data else None
        c_pos_w, c_neg_w = self.c_pos_w, self.c_neg_w
        if M == 0:                    # :32-36: a wrong gt at (0,0,0) with weight 0 and c_pos_w = 0: the positive mean is 0 * x
            c_pos_w = 0.0
        n_neg = N * L * S - M
        if kind >= 2 and n_neg == 0:
            c_neg_w, n_neg = 0.0, 1
        return dict(kind=kind, N=N, L=L, S=S, M=M, ids=(b, i, j), masks=(m0, m1), c_pos_w=c_pos_w, c_neg_w=c_neg_w, n_neg=n_neg,
                    alpha=float(self.loss_config.get("focal_alpha", 0.25)), gamma=float(self.loss_config.get("focal_gamma", 2.0)))

    def _coarse_value(self, conf, t):
        dev = conf.device
        b, i, j = t["ids"]
        m0, m1 = t["masks"]
        sums = torch.zeros(4, dtype=torch.float64, device=dev)
        lib = _lib.load()
        ws = workspace(lib.loftr_loss_workspace_bytes(t["N"], t["L"], t["S"]), dev)
        _lib.check(lib.loftr_coarse_loss_sums(_ptr(_need(conf.contiguous(), "conf")), t["N"], t["L"], t["S"], t["kind"], _ptr(b), _ptr(i),
                                              _ptr(j), t["M"], _ptr(m0), _ptr(m1), t["alpha"], t["gamma"], _ptr(sums), _ptr(ws), ws.numel(),
                                              _stream()), "loftr_coarse_loss_sums")
        s = sums.cpu()
        t["bin_count"] = float(s[3])                  # kind 1: number of supervised dustbin entries (the backward's normaliser)
        pos_mean = s[0] / max(t["M"], 1)
        if t["kind"] == 0:
            loss = t["c_pos_w"] * pos_mean
        elif t["kind"] == 1:
            loss = t["c_pos_w"] * pos_mean + t["c_neg_w"] * (s[2] / s[3])
        else:
            loss = t["c_pos_w"] * pos_mean + t["c_neg_w"] * ((s[2] - s[3]) / t["n_neg"])
        return loss.to(torch.float32).to(dev)

    @_on_device
    def compute_coarse_loss(self, conf, data):
        """:22-99 from the ground-truth id lists (data['spv_*_ids']) instead of the dense conf_matrix_gt.  When `conf` carries an
        autograd graph (loftr_amd.autograd) the result does too: d loss_c / d conf is loftr_coarse_loss_grad."""
        t = self._coarse_terms(conf, data)
        if torch.is_grad_enabled() and conf.requires_grad:
            return _CoarseLossFn.apply(conf, self, t)
        return self._coarse_value(conf, t)

    def _fine_value(self, expec_f, expec_f_gt, with_std):
        dev = expec_f.device
        M = expec_f.shape[0]
        sums = torch.zeros(3, dtype=torch.float64, device=dev)
        lib = _lib.load()
        ws = workspace(lib.loftr_loss_workspace_bytes(1, 1, 1), dev)
        ef = _need(expec_f.contiguous(), "expec_f")
        _lib.check(lib.loftr_fine_loss_sums(_ptr(ef), ef.shape[1], _ptr(_need(expec_f_gt.contiguous(), "expec_f_gt")), M, int(with_std),
                                            float(self.correct_thr), _ptr(sums), _ptr(ws), ws.numel(), _stream()), "loftr_fine_loss_sums")
        s = sums.cpu()
        if s[1] == 0:                 # no correct coarse match
            if not self.training:
                return None, sums
            if with_std:                              # training: correct_mask[0] = True with weight[0] = 0 (:138-143)
                return torch.zeros((), device=dev), sums
            return ((expec_f_gt[0] - expec_f[0, :2]) ** 2).sum().detach(), sums      # plain l2: the false supervision of entry 0 (:113-117)
        norm = (M / s[2]) if with_std else 1.0        # weight = inverse_std / mean(inverse_std)
        return (s[0] * norm / s[1]).to(torch.float32).to(dev), sums

    @_on_device
    def compute_fine_loss(self, expec_f, expec_f_gt):
        """:101-157.  None where the reference returns None (eval mode without a correct coarse match).  With a graph on
        `expec_f` (loftr_amd.autograd.fine_match) the result is differentiable: loftr_fine_loss_grad."""
        if self.fine_type not in ("l2_with_std", "l2"):
            raise NotImplementedError()
        with_std = self.fine_type == "l2_with_std"
        if torch.is_grad_enabled() and expec_f.requires_grad:
            value, sums = self._fine_value(expec_f.detach(), expec_f_gt, with_std)
            if value is None:
                return None
            return _FineLossFn.apply(expec_f, expec_f_gt, self, with_std, value, sums)
        return self._fine_value(expec_f, expec_f_gt, with_std)[0]

    def forward(self, data):
        """:165-192.  Update: data['loss'], data['loss_scalars']."""
        loss_scalars = {}
        conf = data["conf_matrix_with_bin"] if self.sparse_spvs and self.match_type == "sinkhorn" else data["conf_matrix"]
        loss_c = self.compute_coarse_loss(conf, data)
        loss = loss_c * self.loss_config["coarse_weight"]
        loss_scalars.update({"loss_c": loss_c.clone().detach().cpu()})
        loss_f = self.compute_fine_loss(data["expec_f"], data["expec_f_gt"])
        if loss_f is not None:
            loss = loss + loss_f * self.loss_config["fine_weight"]
            loss_scalars.update({"loss_f": loss_f.clone().detach().cpu()})
        else:
            assert self.training is False
            loss_scalars.update({"loss_f": torch.tensor(1.)})
        loss_scalars.update({"loss": loss.clone().detach().cpu()})
        data.update({"loss": loss, "loss_scalars": loss_scalars})


class _CoarseLossFn(torch.autograd.Function):
    """loss_c with d loss_c / d conf from loftr_coarse_loss_grad (csrc/train_bwd.hip)."""

    @staticmethod
    def forward(ctx, conf, module, t):
        ctx.save_for_backward(conf)
        value = module._coarse_value(conf.detach(), t)
        ctx.t = dict(t)                                  # this node's own copy (bin_count belongs to THIS forward)
        return value

    @staticmethod
    @once_differentiable
    def backward(ctx, up):
        (conf,), t = ctx.saved_tensors, ctx.t
        c = _need(conf.detach().contiguous(), "conf")
        g = torch.empty_like(c)
        b, i, j = t["ids"]
        m0, m1 = t["masks"]
        u = float(up.sum())                              # upstream factor of the scalar loss (any shape broadcast to it)
        n_neg = t["bin_count"] if t["kind"] == 1 else t["n_neg"]
        with torch.cuda.device(c.device):
            lib = _lib.load()
            ws = workspace(lib.loftr_loss_workspace_bytes(t["N"], t["L"], t["S"]), c.device)
            _lib.check(lib.loftr_coarse_loss_grad(_ptr(c), t["N"], t["L"], t["S"], t["kind"], _ptr(b), _ptr(i), _ptr(j), t["M"],
                                                  _ptr(m0), _ptr(m1), t["alpha"], t["gamma"], u * t["c_pos_w"] / max(t["M"], 1),
                                                  u * t["c_neg_w"] / n_neg if n_neg else 0.0, _ptr(g), _ptr(ws), ws.numel(), _stream()),
                       "loftr_coarse_loss_grad")
        return g, None, None


class _FineLossFn(torch.autograd.Function):
    """loss_f with d loss_f / d expec_f from loftr_fine_loss_grad (csrc/train_bwd.hip)."""

    @staticmethod
    def forward(ctx, expec_f, expec_f_gt, module, with_std, value, sums):
        ctx.save_for_backward(expec_f, expec_f_gt, sums)
        ctx.args = (with_std, float(module.correct_thr), bool(module.training))
        return value.clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, up):
        expec_f, gt, sums = ctx.saved_tensors
        with_std, thr, training = ctx.args
        e = _need(expec_f.detach().contiguous(), "expec_f")
        g = torch.empty_like(e)
        with torch.cuda.device(e.device):
            _lib.check(_lib.load().loftr_fine_loss_grad(_ptr(e), e.shape[1], _ptr(_need(gt.contiguous(), "expec_f_gt")), e.shape[0],
                                                        int(with_std), thr, int(training), _ptr(sums), float(up.sum()), _ptr(g), _stream()),
                       "loftr_fine_loss_grad")
        return g, None, None, None, None, None
