"""TEST INFRASTRUCTURE ONLY -- numpy (float64) restatement of what torch.autograd derives for the reference between
`loss` and the inputs of its two matching heads:

  * d loss_c / d conf_matrix          LoFTRLoss.compute_coarse_loss     src/losses/loftr_loss.py:22-99  (dual-softmax forms)
  * d conf_matrix / d feat_c0, feat_c1   CoarseMatching.forward          src/loftr/utils/coarse_matching.py:105-119
  * d loss_f / d expec_f              LoFTRLoss._compute_fine_loss_*    src/losses/loftr_loss.py:108-157
  * d expec_f / d feat_f0, feat_f1    FineMatching.forward              src/loftr/utils/fine_matching.py:43-57

The derivatives are written out by hand (chain rule through softmax / clamp / sqrt as torch defines their backward).
Pinned by tests/golden/grad_*.npz: gradients of the reference's OWN modules under torch.autograd
(tests/golden/make_golden_grad.py).  Only tests/ may import this module."""
import numpy as np

F = np.float32


def _open(c, lo, hi):
    """Gradient mask of torch.clamp(c, lo, hi): 1 inside [lo, hi] (bounds included)."""
    c = c.astype(F)
    return ((c >= F(lo)) & (c <= F(hi))).astype(np.float64)


def coarse_loss_grad(conf, conf_gt, weight=None, coarse_type="focal", sparse_spvs=True, alpha=0.25, gamma=2.0, pos_w=1.0,
                     neg_w=1.0):
    """d compute_coarse_loss / d conf for match_type = 'dual_softmax' (loftr_loss.py:22-99)."""
    pos, neg = conf_gt == 1, conf_gt == 0
    weight = None if weight is None else weight.astype(np.float64).copy()
    if not pos.any():                                                     # :31-36
        pos = pos.copy(); pos[0, 0, 0] = True
        if weight is not None:
            weight[0, 0, 0] = 0.
        pos_w = 0.
    if not neg.any():
        neg = neg.copy(); neg[0, 0, 0] = True
        if weight is not None:
            weight[0, 0, 0] = 0.
        neg_w = 0.
    thru = _open(conf, 1e-6, 1 - 1e-6)
    p = np.clip(conf.astype(F), F(1e-6), F(1 - 1e-6)).astype(np.float64)
    w = np.ones_like(p) if weight is None else weight
    if coarse_type == "cross_entropy":
        dpos, dneg = -1.0 / p, 1.0 / (1.0 - p)
    else:
        dpos = alpha * (gamma * np.power(1 - p, gamma - 1) * np.log(p) - np.power(1 - p, gamma) / p)
        dneg = -alpha * (gamma * np.power(p, gamma - 1) * np.log(1 - p) - np.power(p, gamma) / (1 - p))
    g = np.zeros_like(p)
    g[pos] = pos_w * (dpos * w)[pos] / pos.sum()
    if coarse_type == "cross_entropy" or not sparse_spvs:                 # dense: the negatives are supervised too
        g[neg] += neg_w * (dneg * w)[neg] / neg.sum()
    return g * thru


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def dual_softmax_conf_grad(feat_c0, feat_c1, grad_conf, temperature=0.1, mask_c0=None, mask_c1=None):
    """(d L / d feat_c0, d L / d feat_c1) for conf = softmax(sim, 1) * softmax(sim, 2) (coarse_matching.py:105-119)."""
    C = feat_c0.shape[-1]
    f0, f1 = feat_c0.astype(np.float64), feat_c1.astype(np.float64)
    sim = np.einsum("nlc,nsc->nls", f0, f1) / (C * temperature)
    keep = None
    if mask_c0 is not None:
        keep = mask_c0[:, :, None].astype(bool) & mask_c1[:, None, :].astype(bool)
        sim = np.where(keep, sim, -1e9)
    A, B = _softmax(sim, 1), _softmax(sim, 2)
    G = grad_conf.astype(np.float64)
    t = G * A * B
    dsim = 2 * t - B * t.sum(2, keepdims=True) - A * t.sum(1, keepdims=True)
    if keep is not None:
        dsim = np.where(keep, dsim, 0.0)                                   # masked_fill_ cuts the graph
    k = 1.0 / (C * temperature)
    return np.einsum("nls,nsc->nlc", dsim, f1) * k, np.einsum("nls,nlc->nsc", dsim, f0) * k


def fine_loss_grad(expec_f, expec_f_gt, fine_type="l2_with_std", correct_thr=1.0, training=True):
    """d compute_fine_loss / d expec_f [M, 3] (loftr_loss.py:108-157); the std column has no gradient (.detach(), :131)."""
    e, g = expec_f.astype(np.float64), expec_f_gt.astype(np.float64)
    correct = np.abs(expec_f_gt).max(1) < correct_thr
    out = np.zeros_like(e)
    if fine_type == "l2":
        if correct.sum() == 0:
            if not training:
                return None
            correct = correct.copy(); correct[0] = True
        out[correct, :2] = -2 * (g[correct] - e[correct, :2]) / correct.sum()
        return out
    inv = 1.0 / np.clip(e[:, 2], 1e-10, None)
    weight = inv / inv.mean()
    if not correct.any():
        if not training:
            return None
        correct = correct.copy(); correct[0] = True
        weight[0] = 0.
    out[correct, :2] = -2 * (g[correct] - e[correct, :2]) * weight[correct, None] / correct.sum()
    return out


def fine_matching_grad(feat_f0, feat_f1, grad_expec):
    """(d L / d feat_f0, d L / d feat_f1) [M, WW, C] from d L / d expec_f [M, 3] (fine_matching.py:43-57)."""
    M, WW, C = feat_f0.shape
    W = int(round(np.sqrt(WW)))
    f0, f1, g = feat_f0.astype(np.float64), feat_f1.astype(np.float64), grad_expec.astype(np.float64)
    pick = f0[:, WW // 2, :]
    temp = 1.0 / np.sqrt(C)
    heat = _softmax(temp * np.einsum("mc,mrc->mr", pick, f1), 1)
    lin = np.linspace(-1, 1, W)
    gx, gy = np.tile(lin, W), np.repeat(lin, W)                            # create_meshgrid(W, W, True): x fastest
    grid = np.stack([gx, gy], -1)                                          # [WW, 2]
    c = heat @ grid                                                        # [M, 2]
    var = heat @ grid ** 2 - c ** 2
    h = np.where(var >= 1e-10, 0.5 / np.sqrt(np.clip(var, 1e-10, None)), 0.0) * g[:, 2:3]     # d L / d var
    dc = g[:, :2] - 2 * c * h                                              # d L / d coords (direct + through var)
    dheat = dc @ grid.T + h @ (grid ** 2).T                                # [M, WW]
    dsim = temp * heat * (dheat - (heat * dheat).sum(1, keepdims=True))
    d1 = dsim[:, :, None] * pick[:, None, :]
    d0 = np.zeros_like(f0)
    d0[:, WW // 2, :] = np.einsum("mr,mrc->mc", dsim, f1)
    return d0, d1


# ---- Sinkhorn head (coarse_matching.py:121-143 + SuperGlue log_optimal_transport) -------------------------------------------
def sparse_sinkhorn_loss_grad(assign, conf_gt, weight=None, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0):
    """d compute_coarse_loss / d conf_matrix_with_bin [N, L+1, S+1] for sparse_spvs + match_type 'sinkhorn' (loftr_loss.py:56-81):
    focal positives at the ground-truth cells of the interior, focal 'negatives' at the dustbin entries of the rows / columns
    without a ground-truth match (kept iff the row / column has some loss weight)."""
    pos = conf_gt == 1
    weight = None if weight is None else weight.astype(np.float64).copy()
    if not pos.any():
        pos = pos.copy(); pos[0, 0, 0] = True
        if weight is not None:
            weight[0, 0, 0] = 0.
        pos_w = 0.
    thru = _open(assign, 1e-6, 1 - 1e-6)
    p = np.clip(assign.astype(F), F(1e-6), F(1 - 1e-6)).astype(np.float64)
    dfocal = alpha * (gamma * np.power(1 - p, gamma - 1) * np.log(p) - np.power(1 - p, gamma) / p)
    g = np.zeros_like(p)
    w = np.ones(conf_gt.shape) if weight is None else weight
    gi = g[:, :-1, :-1]
    gi[pos] = pos_w * (dfocal[:, :-1, :-1] * w)[pos] / pos.sum()
    neg0, neg1 = conf_gt.sum(-1) == 0, conf_gt.sum(1) == 0          # [N, L], [N, S]
    if weight is not None:
        neg0 = neg0 & (weight.sum(-1) != 0)
        neg1 = neg1 & (weight.sum(1) != 0)
    n_neg = neg0.sum() + neg1.sum()
    g[:, :-1, -1][neg0] = neg_w * dfocal[:, :-1, -1][neg0] / n_neg
    g[:, -1, :-1][neg1] = neg_w * dfocal[:, -1, :-1][neg1] / n_neg
    return g * thru


def _lse(x, axis):
    m = x.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def sinkhorn_conf_grad(feat_c0, feat_c1, bin_score, grad_assign, iters=3, mask_c0=None, mask_c1=None):
    """(d L / d feat_c0, d L / d feat_c1, d L / d bin_score) for assign = exp(log_optimal_transport(sim, bin_score, iters)),
    sim = <feat_c0, feat_c1> / C with -1e9 on padding (coarse_matching.py:121-132).  Reverse mode through the unrolled
    iterations u_t = log_mu - LSE_j(Z + v_{t-1}), v_t = log_nu - LSE_i(Z + u_t), written out by hand."""
    N, L, C = feat_c0.shape
    S = feat_c1.shape[1]
    f0, f1 = feat_c0.astype(np.float64), feat_c1.astype(np.float64)
    sim = np.einsum("nlc,nsc->nls", f0, f1) / C
    keep = None
    if mask_c0 is not None:
        keep = mask_c0[:, :, None].astype(bool) & mask_c1[:, None, :].astype(bool)
        sim = np.where(keep, sim, -1e9)
    Z = np.full((N, L + 1, S + 1), float(bin_score))
    Z[:, :L, :S] = sim
    norm = -np.log(L + S)
    log_mu = np.concatenate([np.full(L, norm), [np.log(S) + norm]])[None].repeat(N, 0)
    log_nu = np.concatenate([np.full(S, norm), [np.log(L) + norm]])[None].repeat(N, 0)
    us, vs = [], [np.zeros((N, S + 1))]
    for _ in range(iters):
        us.append(log_mu - _lse(Z + vs[-1][:, None, :], 2))
        vs.append(log_nu - _lse(Z + us[-1][:, :, None], 1))
    if iters == 0:
        us.append(np.zeros((N, L + 1)))
    out = Z + us[-1][:, :, None] + vs[-1][:, None, :] - norm
    D = grad_assign.astype(np.float64) * np.exp(out)                 # d L / d out
    dZ = D.copy()
    du, dv = D.sum(2), D.sum(1)
    for t in range(iters, 0, -1):
        Pc = np.exp(Z + us[t - 1][:, :, None] + vs[t][:, None, :] - log_nu[:, None, :])         # softmax over i of Z + u_t
        dZ -= dv[:, None, :] * Pc
        du = du - (dv[:, None, :] * Pc).sum(2)
        Pr = np.exp(Z + vs[t - 1][:, None, :] + us[t - 1][:, :, None] - log_mu[:, :, None])     # softmax over j of Z + v_{t-1}
        dZ -= du[:, :, None] * Pr
        dv = -(du[:, :, None] * Pr).sum(1)
        du = np.zeros_like(du)
    dbin = dZ[:, L, :].sum() + dZ[:, :L, S].sum()
    dsim = dZ[:, :L, :S]
    if keep is not None:
        dsim = np.where(keep, dsim, 0.0)
    return np.einsum("nls,nsc->nlc", dsim, f1) / C, np.einsum("nls,nlc->nsc", dsim, f0) / C, dbin
