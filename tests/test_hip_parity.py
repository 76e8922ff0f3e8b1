"""GPU parity tests: the HIP hot path (through the C-ABI) against
  (1) the committed golden vectors produced by the REAL reference (tests/golden/*.npz), and
  (2) the numpy oracle run on the same seeded inputs (small sizes only).

Tolerances are BASELINE.json's: |mconf| <= 1e-4, |mkpts*_f| <= 1e-3 px.  Index outputs
(b_ids, i_ids, j_ids) must be identical; a match may flip only if it is provably borderline
(see _cases.compare_to_golden).
"""
import numpy as np
import pytest

from _cases import (SMALL_CASES, MID_CASES, FULL_CASES, TOL_CONF, load_case, compare_to_golden,
                    check_conf_digest, run_hip)

pytestmark = pytest.mark.gpu


def _ref_noise_px(g):
    """|mkpts1_f| distance between the reference's float32 run (the golden) and the SAME reference module run in float64, stored by
    make_golden.py for the cases that ask for it ('ref64': the "trained-like" peaked cases): the reference's own rounding noise."""
    if "ref64/mkpts1_f" not in g:
        return 0.0
    k64 = {k: n for n, k in enumerate(zip(g["ref64/b_ids"].tolist(), g["ref64/i_ids"].tolist(), g["ref64/j_ids"].tolist()))}
    com = [(n, k64[k]) for n, k in enumerate(zip(g["b_ids"].tolist(), g["i_ids"].tolist(), g["j_ids"].tolist())) if k in k64]
    ia, ib = [c[0] for c in com], [c[1] for c in com]
    return float(np.abs(g["mkpts1_f"][ia].astype(np.float64) - g["ref64/mkpts1_f"][ib]).max())


def _check_against_golden(name, out, inp, g, max_flips=0):
    # two independent fp32 roundings of the same exact result, each within n of it, are within 2n of each other: the bar cannot be below
    # twice the reference's own distance from exact arithmetic (peaked_ds: the reference's fp32 forward is 9.2e-4 px from its fp64 forward; same rule as tests/test_e2e_golden.py)
    tol_px = max(1e-3, 2.0 * _ref_noise_px(g))
    rep = compare_to_golden(out, g, inp["cfg"]["match_coarse"]["thr"], tol_px=tol_px, max_flips=max_flips)
    if "ref64/mkpts1_f" in g:
        # the margin guard of tests/test_e2e_golden.py on the feature-level cases that carry the reference's float64 run: OUR distance to
        # it against the reference's own float32 distance -- RMS over the common matches <= 1.5 x, maximum <= 1.6 x (+ one fp32 ulp of a coordinate)
        def dist(side):
            k64 = {k: n for n, k in enumerate(zip(g["ref64/b_ids"].tolist(), g["ref64/i_ids"].tolist(), g["ref64/j_ids"].tolist()))}
            com = [(n, k64[k]) for n, k in enumerate(zip(side["b_ids"].tolist(), side["i_ids"].tolist(), side["j_ids"].tolist())) if k in k64]
            ia, ib = [c[0] for c in com], [c[1] for c in com]
            d = np.asarray(side["mkpts1_f"])[ia].astype(np.float64) - g["ref64/mkpts1_f"][ib]
            c = np.asarray(side["mconf"])[ia].astype(np.float64) - g["ref64/mconf"][ib]
            return float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())), float(np.abs(c).max()), float(np.sqrt((c ** 2).mean()))
        pm, pr, cm, cr = dist(out)
        rpm, rpr, rcm, rcr = dist(g)
        import os
        rep_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(rep_dir, exist_ok=True)
        with open(os.path.join(rep_dir, "parity_features.txt"), "a") as fh:
            fh.write(f"{name:28s} vs ref-fp64: ours px max {pm:.2e} rms {pr:.2e} conf max {cm:.2e} rms {cr:.2e} | ref-fp32 itself px max {rpm:.2e} rms {rpr:.2e} "
                     f"conf max {rcm:.2e} rms {rcr:.2e} | ratios px {pm / max(rpm, 1e-12):.2f} / {pr / max(rpr, 1e-12):.2f}\n")
        # (key points only: the confidences of this case are saturated -- the reference's fp32 and fp64 runs agree to 4e-16 on them --
        #  while pass B of the sweep evaluates conf = exp2(2 v log2e - LSE_row - LSE_col) with O(100) fp32 biases: ~1e-5 relative at
        #  conf ~ 1 by design, csrc/score_sweep.h; measured 4.2e-5, held to the north-star 1e-4 by compare_to_golden above)
        assert pr <= 1.5 * rpr + 6e-5, (name, "rms", pr, rpr)
        assert pm <= 1.6 * rpm + 6e-5, (name, "max", pm, rpm)
    check_conf_digest(out["conf_matrix"], g)
    if "conf_matrix" in g:
        assert np.abs(out["conf_matrix"] - g["conf_matrix"]).max() <= TOL_CONF
    if "conf_matrix_with_bin" in g:
        # dustbin entries of masked problems are O(100) (mass of the padded rows): relative bound there
        ref = g["conf_matrix_with_bin"]
        assert (np.abs(out["conf_matrix_with_bin"] - ref) <= TOL_CONF * np.maximum(1.0, np.abs(ref))).all()
    if "assign_bin_col" in g and "conf_matrix_with_bin" in out:
        a = out["conf_matrix_with_bin"]
        for got, ref in ((a[:, :, -1], g["assign_bin_col"]), (a[:, -1, :], g["assign_bin_row"])):
            assert (np.abs(got - ref) <= TOL_CONF * np.maximum(1.0, np.abs(ref))).all()
    assert out["expec_f"].shape == g["expec_f"].shape or max_flips
    for k in ("b_ids", "i_ids", "j_ids", "m_bids"):
        assert out[k].dtype == np.int64, k
    for k in ("mkpts0_c", "mkpts1_c", "mkpts0_f", "mkpts1_f", "mconf", "expec_f"):
        assert out[k].dtype == np.float32, k
    assert out["gt_mask"].dtype == np.bool_
    return rep


@pytest.mark.parametrize("name", SMALL_CASES + MID_CASES)
def test_hip_vs_reference_golden(name):
    rc, inp, g = load_case(name)
    out = run_hip(inp)
    rep = _check_against_golden(name, out, inp, g)
    assert rep["M_out"] == rep["M_ref"]
    if len(g["expec_f"]):
        # the std column is ill-conditioned (sqrt(clamp(var,1e-10))): loose bound
        assert np.abs(out["expec_f"][:, 2] - g["expec_f"][:, 2]).max() <= 5e-3


@pytest.mark.parametrize("name", FULL_CASES)
def test_hip_vs_reference_golden_full(name):
    """BASELINE geometries: 640x480 (L=S=4800) dual-softmax / sinkhorn, 840x840 masked."""
    rc, inp, g = load_case(name)
    out = run_hip(inp)
    _check_against_golden(name, out, inp, g, max_flips=2)


@pytest.mark.parametrize("name", ["small_ds", "small_mask", "small_ot", "small_ds_corr"])
def test_hip_vs_oracle_stages(name):
    """Stage-by-stage against the numpy oracle: coarse transformer output, conf volume,
    fine windows before/after the fine transformer."""
    import torch
    from oracle import loftr_oracle as O
    from _cases import build_hip_matcher
    rc, inp, g = load_case(name)
    ref = O.loftr_hot_path(inp["feat_c0"], inp["feat_c1"], inp["feat_f0"], inp["feat_f1"], inp["w"], inp["cfg"],
                           inp["hw0_i"], inp["hw1_i"], inp["mask0"], inp["mask1"], inp["scale0"], inp["scale1"],
                           keep_intermediates=True)
    model = build_hip_matcher(inp["cfg"], inp["w"])
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fc0 = model.pos_encoding(t(inp["feat_c0"]))
    fc1 = model.pos_encoding(t(inp["feat_c1"]))
    tbf = inp["cfg"]["coarse"]["temp_bug_fix"]
    assert np.abs(fc0.cpu().numpy() - O.add_pos_flatten(inp["feat_c0"], tbf)).max() <= 1e-5
    # channels-last storage (what the backbone hands over) must give the same bytes
    fc0_cl = model.pos_encoding(t(inp["feat_c0"]).contiguous(memory_format=torch.channels_last))
    assert torch.equal(fc0_cl, fc0)
    m0 = m1 = None
    if inp["mask0"] is not None:
        m0, m1 = t(inp["mask0"]).flatten(-2), t(inp["mask1"]).flatten(-2)
    fc0, fc1 = model.loftr_coarse(fc0, fc1, m0, m1)
    assert np.abs(fc0.cpu().numpy() - ref["feat_c0"]).max() <= 2e-4
    assert np.abs(fc1.cpu().numpy() - ref["feat_c1"]).max() <= 2e-4
    out = run_hip(inp, model=model)
    assert np.abs(out["conf_matrix"] - ref["conf_matrix"]).max() <= TOL_CONF
    assert np.array_equal(out["b_ids"], ref["b_ids"])
    assert np.array_equal(out["i_ids"], ref["i_ids"])
    assert np.array_equal(out["j_ids"], ref["j_ids"])
    assert np.abs(out["mkpts1_f"] - ref["mkpts1_f"]).max() <= 1e-3


def test_conf_matrix_elided_same_matches():
    """materialize_conf=False (conf_matrix never written) must select the same matches."""
    rc, inp, g = load_case("mid_ds")
    a = run_hip(inp, materialize_conf=True)
    b = run_hip(inp, materialize_conf=False)
    assert "conf_matrix" not in b
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts1_f"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("name", ["small_ds", "mid_ds"])
def test_channels_last_features_identical(name):
    """NCHW and channels-last backbone outputs go through different load paths (transposing
    flatten / strided gather vs coalesced) and must produce bitwise identical results."""
    rc, inp, g = load_case(name)
    a = run_hip(inp, channels_last=False)
    b = run_hip(inp, channels_last=True)
    for k in ("conf_matrix", "b_ids", "i_ids", "j_ids", "mconf", "mkpts1_f", "expec_f"):
        assert np.array_equal(a[k], b[k]), k


def test_linear_building_block():
    import torch
    from loftr_amd import ops
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in [(1, 128, 256), (100, 256, 256), (4800, 512, 512), (333, 128, 128), (130, 384, 16)]:
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g)
        out = ops.linear(a.cuda(), w.cuda()).cpu()
        ref = (a.double() @ w.double().T)
        err = (out.double() - ref).abs().max().item()
        assert err <= 1e-4 * K ** 0.5, (M, N, K, err)


def test_determinism():
    """Two runs give bitwise identical results (fixed-order reductions, no float atomics)."""
    rc, inp, g = load_case("mid_ds")
    a = run_hip(inp)
    b = run_hip(inp)
    for k in ("conf_matrix", "mconf", "mkpts1_f", "expec_f", "j_ids"):
        assert np.array_equal(a[k], b[k]), k


def test_properties_full_size():
    """Size-independent properties at the BASELINE size (N=2, L=S=4800), no oracle needed:
    rows/cols of the dual-softmax conf are bounded by the softmax simplex, matches are mutual
    argmaxes of the returned conf_matrix, ordered, inside the border, above the threshold."""
    rc, inp, g = load_case("full_ds_thr02")
    out = run_hip(inp)
    conf = out["conf_matrix"]
    assert conf.shape == (2, 4800, 4800)
    # conf = exp2(2 v log2e - LSE_row - LSE_col) (csrc/coarse_match.hip, sweep pass B): the two log-sum-exp biases are
    # O(100) fp32 numbers, so a confidence of 1 carries ~1e-5 of rounding (the matching tolerance is 1e-4)
    assert conf.min() >= 0 and conf.max() <= 1 + 3e-5
    assert conf.sum(2).max() <= 1 + 1e-4 and conf.sum(1).max() <= 1 + 1e-4
    b, i, j = out["b_ids"], out["i_ids"], out["j_ids"]
    assert np.array_equal(out["mconf"], conf[b, i, j])
    assert np.array_equal(conf[b, i, j], conf[b, i].max(-1))
    assert np.array_equal(conf[b, i, j], conf[b, :, j].max(-2)) if len(b) < 64 else True
    assert (out["mconf"] > rc["mc"]["thr"]).all()
    key = b * 4800 + i
    assert (np.diff(key) > 0).all()
    br = rc["mc"]["border_rm"]
    for ids, (h, w) in ((i, rc["hw0_c"]), (j, rc["hw1_c"])):
        y, x = ids // w, ids % w
        assert (y >= br).all() and (y < h - br).all() and (x >= br).all() and (x < w - br).all()
    assert np.array_equal(out["_match_counts"][1:], np.bincount(b, minlength=2))
    assert out["_match_counts"][0] == len(b)


def test_batch_consistency_full_size():
    """Pairs are independent: a batch of 3 pairs at the BASELINE size must give bitwise the same
    per-pair results as the three pairs run one at a time -- tiles of the batched launches straddle pair
    boundaries (3 * 4800 rows is not a multiple of any tile height), the arithmetic per output element may not."""
    import copy
    from loftr_amd.synth import make_features
    from _cases import build_hip_matcher
    rc, inp, g = load_case("full_ds_thr0")
    c0, c1, f0, f1 = make_features(123, 3, (60, 80), (60, 80), corr=0.4)
    model = build_hip_matcher(inp["cfg"], inp["w"])
    base = dict(inp, feat_c0=c0, feat_c1=c1, feat_f0=f0, feat_f1=f1)
    full = run_hip(base, model=model)
    off = 0
    for b in range(3):
        one = run_hip(dict(base, feat_c0=c0[b:b + 1], feat_c1=c1[b:b + 1], feat_f0=f0[b:b + 1], feat_f1=f1[b:b + 1]),
                      model=model)
        m = int(full["_match_counts"][1 + b])
        assert m == len(one["mconf"]) and m > 100
        sl = slice(off, off + m)
        assert np.array_equal(full["conf_matrix"][b], one["conf_matrix"][0])
        for k in ("i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f", "expec_f"):
            assert np.array_equal(full[k][sl], one[k]), (b, k)
        assert (full["b_ids"][sl] == b).all()
        off += m
    assert off == len(full["mconf"])


@pytest.mark.parametrize("C,L,S,masked", [(256, 300, 300, False), (256, 130, 75, True), (128, 25, 25, False)])
def test_single_encoder_layer_vs_oracle(C, L, S, masked):
    """loftr_encoder_layer_fwd (the per-layer entry point, transformer.py:35-58) against the numpy oracle:
    self- and cross-attention, unequal lengths, padding masks, both widths."""
    import torch
    from oracle import loftr_oracle as O
    from loftr_amd.loftr import LoFTREncoderLayer
    from loftr_amd.synth import _encoder_layer
    rng = np.random.default_rng(C + L)
    w = {}
    _encoder_layer(rng, "l.", C, w)
    layer = LoFTREncoderLayer(C, 8).eval()
    layer.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in w.items()})
    layer = layer.cuda()
    nb = 3
    x = rng.standard_normal((nb, L, C)).astype(np.float32)
    src = rng.standard_normal((nb, S, C)).astype(np.float32)
    xm = sm = None
    if masked:
        xm = np.ones((nb, L), bool); xm[1, L // 2:] = False
        sm = np.ones((nb, S), bool); sm[2, S // 3:] = False
    t = lambda a: None if a is None else torch.from_numpy(a).cuda()
    for self_attn in (False, True):
        if self_attn:
            got = layer(t(x), t(x), t(xm), t(xm)).cpu().numpy()
            ref = O.encoder_layer(x, x, w, "l.", 8, xm, xm)
        else:
            got = layer(t(x), t(src), t(xm), t(sm)).cpu().numpy()
            ref = O.encoder_layer(x, src, w, "l.", 8, xm, sm)
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (self_attn, np.abs(got - ref).max())


@pytest.mark.parametrize("hw0,hw1,C,masked,prefilter", [((7, 9), (9, 7), 256, False, False), ((7, 9), (5, 7), 256, True, True),
                                                         ((6, 8), (8, 6), 128, False, False), ((9, 12), (12, 9), 256, True, False),
                                                         ((5, 7), (75, 91), 256, True, False), ((6, 6), (72, 96), 256, False, True),
                                                         ((4, 5), (105, 105), 256, False, False)])
def test_sinkhorn_paths_vs_oracle(hw0, hw1, C, masked, prefilter):
    """Sinkhorn coarse matching against the numpy oracle on the shapes the goldens do not reach: rows that are not
    16-byte aligned (S % 4 != 0: unaligned 16-byte groups and a ragged last group in the row-streaming passes), another
    descriptor width (tiled score store), masks on unequal grids, and rows wider than 5119 columns (the 512-thread variant
    of the passes: the outdoor 105 x 105 grid and an aligned one)."""
    import torch
    from oracle import loftr_oracle as O
    from loftr_amd import ops
    rng = np.random.default_rng(hw0[0] * 100 + hw1[1] + C)
    N, L, S = 3, hw0[0] * hw0[1], hw1[0] * hw1[1]
    f0 = rng.standard_normal((N, L, C)).astype(np.float32) * 2
    f1 = rng.standard_normal((N, S, C)).astype(np.float32) * 2
    k = min(L, S)
    cols = np.arange(k) if S < 1000 else np.sort(rng.permutation(S)[:k])       # (wide grids: the first k cells are all border cells)
    f1[:, cols] += 1.5 * f0[:, rng.permutation(L)[:k]]
    m0 = m1 = None
    if masked:
        m0 = np.ones((N,) + hw0, bool); m0[1, hw0[0] - 2:] = False
        m1 = np.ones((N,) + hw1, bool); m1[2, :, hw1[1] - 3:] = False
    conf_ref, assign_ref = O.sinkhorn_conf(f0, f1, np.float32(1.0), 3, None if m0 is None else m0.reshape(N, -1),
                                           None if m1 is None else m1.reshape(N, -1), prefilter=prefilter)
    sel = O.coarse_match_select(conf_ref, 0.0, 1, hw0, hw1, (hw0[0] * 8, hw0[1] * 8), m0, m1)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    r = ops.coarse_match(t(f0), t(f1), hw0, hw1, thr=0.0, border_rm=1, scale=8.0, match_type="sinkhorn", bin_score=1.0, skh_iters=3,
                         skh_prefilter=prefilter, mask0=None if m0 is None else t(m0).flatten(-2),
                         mask1=None if m1 is None else t(m1).flatten(-2), want_assign=True)
    conf, assign = r["conf_matrix"].cpu().numpy(), r["conf_matrix_with_bin"].cpu().numpy()
    valid = np.ones((N, L, S), bool) if m0 is None else (m0.reshape(N, -1)[:, :, None] & m1.reshape(N, -1)[:, None, :])
    # masked entries are rounding noise of ((Z + u) + v) - norm with |u|, |v| ~ 1e9: compared where valid
    assert np.abs(conf - conf_ref)[valid].max() <= TOL_CONF
    assert np.abs(assign[:, :-1, :-1] - assign_ref[:, :-1, :-1])[valid].max() <= TOL_CONF
    if m0 is None:
        assert np.abs(assign[:, -1, :] - assign_ref[:, -1, :]).max() <= TOL_CONF and np.abs(assign[:, :, -1] - assign_ref[:, :, -1]).max() <= TOL_CONF
    got = set(zip(r["b_ids"].tolist(), r["i_ids"].tolist(), r["j_ids"].tolist()))
    want = set(zip(sel["b_ids"].tolist(), sel["i_ids"].tolist(), sel["j_ids"].tolist()))
    assert len(got ^ want) <= 1, sorted(got ^ want)            # (a near-tie between two fp32 roundings may flip one)
    assert len(want) > 0 or prefilter


@pytest.mark.parametrize("hw0,hw1,C,masked", [((7, 9), (9, 7), 128, False), ((6, 8), (5, 7), 64, True), ((9, 11), (7, 5), 256, True)])
def test_dual_softmax_paths_vs_oracle(hw0, hw1, C, masked):
    """Dual-softmax coarse matching against the numpy oracle off the goldens' shapes: descriptor widths other than 256
    (the tiled two-pass kernels) and odd, unequal grids with masks on the sweep (partial panels / row blocks)."""
    import torch
    from oracle import loftr_oracle as O
    from loftr_amd import ops
    rng = np.random.default_rng(hw0[0] * 10 + hw1[1] + C)
    N, L, S = 3, hw0[0] * hw0[1], hw1[0] * hw1[1]
    f0 = rng.standard_normal((N, L, C)).astype(np.float32)
    f1 = rng.standard_normal((N, S, C)).astype(np.float32)
    k = min(L, S)
    f1[:, :k] += 1.5 * f0[:, rng.permutation(L)[:k]]
    m0 = m1 = None
    if masked:
        m0 = np.ones((N,) + hw0, bool); m0[1, hw0[0] - 2:] = False
        m1 = np.ones((N,) + hw1, bool); m1[2, :, hw1[1] - 2:] = False
    conf_ref = O.dual_softmax_conf(f0, f1, 0.1, None if m0 is None else m0.reshape(N, -1), None if m1 is None else m1.reshape(N, -1))
    sel = O.coarse_match_select(conf_ref, 0.0, 1, hw0, hw1, (hw0[0] * 8, hw0[1] * 8), m0, m1)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    r = ops.coarse_match(t(f0), t(f1), hw0, hw1, thr=0.0, border_rm=1, scale=8.0, temperature=0.1,
                         mask0=None if m0 is None else t(m0).flatten(-2), mask1=None if m1 is None else t(m1).flatten(-2))
    assert np.abs(r["conf_matrix"].cpu().numpy() - conf_ref).max() <= TOL_CONF
    got = set(zip(r["b_ids"].tolist(), r["i_ids"].tolist(), r["j_ids"].tolist()))
    want = set(zip(sel["b_ids"].tolist(), sel["i_ids"].tolist(), sel["j_ids"].tolist()))
    assert len(got ^ want) <= 1 and len(want) > 10, sorted(got ^ want)


def _coarse_transformer_case(N, L0, L1, masked):
    import torch
    from loftr_amd import LoFTR, get_cfg
    from loftr_amd.synth import make_weights
    cfg = get_cfg(thr=0.0)
    model = LoFTR(cfg).eval()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(0, cfg).items()}, strict=False)
    tr = model.cuda().loftr_coarse
    g = torch.Generator(device="cpu").manual_seed(N * 7 + L0 + L1)
    f0 = torch.randn(N, L0, 256, generator=g).cuda()
    f1 = torch.randn(N, L1, 256, generator=g).cuda()
    m0 = m1 = None
    if masked:
        m0 = torch.ones(N, L0, dtype=torch.bool); m0[:, L0 - L0 // 5:] = False
        m1 = torch.ones(N, L1, dtype=torch.bool); m1[0, L1 - L1 // 3:] = False
        m0, m1 = m0.cuda(), m1.cuda()
    structs = [layer.weight_struct() for layer in tr.layers]
    prepared = tr._prepared(structs, f0.device)

    def run(mode, diag=None):
        from loftr_amd import ops
        with torch.no_grad():
            o = ops.transformer(f0, f1, structs, tr.layer_names, tr.nhead, m0, m1, inplace=False, prepared=prepared, mode=mode, diag=diag)
        torch.cuda.synchronize()
        return o[0].clone(), o[1].clone()
    return run, tr


COARSE_SHAPES = [(8, 4800, 4800, False), (3, 300, 300, False), (2, 700, 500, True)]


@pytest.mark.parametrize("shape", COARSE_SHAPES)
def test_scheduled_transformer_is_bit_identical_to_call_order(shape):
    """The coarse transformer as launches runs a schedule of two-job launches (csrc/transformer.hip: coarse_transformer_scheduled -- the
    next self-attention call on image 0 rides in the idle slots of the cross call on image 1); debug switch encoder_schedule = 0 keeps
    the reference's call order (transformer.py:92-97).  Both must give the same bits: full batch-8 size (a split self call), fewer
    sequences than XCDs, unequal masked grids."""
    import torch
    from loftr_amd import ops
    run, _ = _coarse_transformer_case(*shape)
    with ops.debug_switch(encoder_schedule=0):
        a = run("launches")
    assert ops.debug_get("encoder_schedule") == (1, 1)
    b = run("launches")
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("shape", COARSE_SHAPES + [(1, 4800, 4800, False), (2, 11025, 11025, True)])
def test_persistent_transformer_order_independent_and_matches_launches(shape):
    """Round 6: the coarse transformer as ONE persistent launch (csrc/encoder_fused.hip: coarse_persistent_kernel) whose 256 resident
    workgroups pull K / F / X work items from a host-planned queue and wait on per-pair / per-tile counters (transformer.py:96-97: the
    dependency `feat1 attends to the updated feat0` is per pair).  The queue order must not change a bit: the dependency-driven plan
    (critical path first) against the plan in the reference's call order, repeated runs against the first (a difference is a race);
    against the per-call launches the results agree to float32 noise (the K items sum the four waves' K^T V blocks in another order than
    proj_kv_kernel).  The status word stays 0."""
    import torch
    run, _ = _coarse_transformer_case(*shape)
    ref = run("launches")
    scale = max(ref[0].abs().max().item(), ref[1].abs().max().item())
    outs = {}
    for mode in ("persistent_call_order", "persistent"):
        diag = torch.zeros(16, dtype=torch.uint8, device="cuda")
        outs[mode] = run(mode, diag)
        assert int(diag.view(torch.int32)[0].item()) == 0, mode
        for _ in range(2):
            diag.zero_()
            again = run(mode, diag)
            assert int(diag.view(torch.int32)[0].item()) == 0
            assert torch.equal(again[0], outs[mode][0]) and torch.equal(again[1], outs[mode][1]), mode
    a, b = outs["persistent_call_order"], outs["persistent"]
    assert torch.isfinite(b[0]).all() and torch.isfinite(b[1]).all()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in range(2):
        assert (b[k] - ref[k]).abs().max().item() <= 2e-6 * scale, (k, (b[k] - ref[k]).abs().max().item(), scale)


@pytest.mark.parametrize("shape", [(2, 700, 500), (2, 1200, 1200), (3, 11025, 11025)])
def test_padded_tiles_are_skipped_without_touching_any_other_token(shape):
    """ABI 24, loftr_transformer_fwd_padded (MegaDepth padding masks): a 128-token tile without a valid token is not computed -- its rows
    come back as they went in -- and EVERY other token is bit-identical to loftr_transformer_fwd: the tile's K^T V / Ksum partial is
    exactly +0 either way (linear_attention.py:37-40).  Masks: a padded tail (bottom padding), a fully padded tile in the middle of
    sequence 1 of image 1, a pair without padding next to padded ones; unequal grids; the 105 x 105 outdoor grid."""
    import torch
    from loftr_amd import ops
    N, L0, L1 = shape
    run, tr = _coarse_transformer_case(N, L0, L1, False)
    structs = [layer.weight_struct() for layer in tr.layers]
    prepared = tr._prepared(structs, torch.device("cuda", 0))
    g = torch.Generator(device="cpu").manual_seed(5 * N + L0)
    f0 = torch.randn(N, L0, 256, generator=g).cuda()
    f1 = torch.randn(N, L1, 256, generator=g).cuda()
    m0 = torch.ones(N, L0, dtype=torch.bool); m1 = torch.ones(N, L1, dtype=torch.bool)
    m0[0, L0 - (2 * L0) // 5:] = False                                   # padded tail: >= one whole tile and a partly padded one
    m1[0, L1 - L1 // 3:] = False
    if L1 >= 4 * 128:
        m1[-1, 128:256] = False                                          # a whole tile in the middle
        m1[-1, 300:310] = False                                          # and a few tokens of a valid tile
    m0, m1 = m0.cuda(), m1.cuda()
    with torch.no_grad():
        ref = ops.transformer(f0, f1, structs, tr.layer_names, tr.nhead, m0, m1, inplace=False, prepared=prepared, mode="launches")
        ref = (ref[0].clone(), ref[1].clone())
        out = ops.transformer(f0, f1, structs, tr.layer_names, tr.nhead, m0, m1, inplace=False, prepared=prepared, mode="persistent", skip_padded=True)
    torch.cuda.synchronize()
    skipped = 0
    for k, (x, m, o, r) in enumerate(((f0, m0, out[0], ref[0]), (f1, m1, out[1], ref[1]))):
        L = x.shape[1]
        assert torch.isfinite(o).all()
        for n in range(N):
            for t0 in range(0, L, 128):
                sl = slice(t0, min(t0 + 128, L))
                if m[n, sl].any():
                    assert torch.equal(o[n, sl], r[n, sl]), (k, n, t0)
                else:
                    skipped += 1
                    assert torch.equal(o[n, sl], x[n, sl]), (k, n, t0)
                    assert not torch.equal(r[n, sl], x[n, sl])            # (the reference does compute something there)
    assert skipped >= 3


def test_persistent_transformer_refuses_a_plan_of_another_shape():
    """loftr_transformer_fwd_planned with a plan built for other sizes: nothing is computed, the status word reports 2, and the C-ABI
    call returns an error for a plan buffer that is too small (include/loftr_hip.h)."""
    import ctypes as C
    import torch
    from loftr_amd import ops, _lib
    lib = _lib.load()
    run, tr = _coarse_transformer_case(2, 300, 300, False)
    kinds = [{"self": 0, "cross": 1}[n] for n in tr.layer_names]
    arr = (C.c_int * len(kinds))(*kinds)
    assert lib.loftr_coarse_plan_bytes(arr, len(kinds), 2, 300, 300) > 0
    assert lib.loftr_coarse_plan_bytes(arr, 3, 2, 300, 300) == 0                  # an odd number of layers has no persistent form
    assert lib.loftr_coarse_plan_signature(len(kinds), 2, 300, 300, 0) != lib.loftr_coarse_plan_signature(len(kinds), 2, 300, 300, 1)
    # a plan for (2, 300, 260) has the same number of items as (2, 300, 300) (3 + 3 tiles): hand it to the (2, 300, 300) call
    other = ops.coarse_plan(kinds, 2, 300, 260, torch.device("cuda:0"))
    key = ("cuda:0", tuple(kinds), 2, 300, 300, 0)
    saved = ops._COARSE_PLANS.get(key)
    ops._COARSE_PLANS[key] = other
    try:
        diag = torch.zeros(16, dtype=torch.uint8, device="cuda")
        run("persistent", diag)
        assert int(diag.view(torch.int32)[0].item()) == 2
    finally:
        if saved is None:
            ops._COARSE_PLANS.pop(key, None)
        else:
            ops._COARSE_PLANS[key] = saved
    diag = torch.zeros(16, dtype=torch.uint8, device="cuda")
    a = run("persistent", diag)
    b = run("launches")
    assert int(diag.view(torch.int32)[0].item()) == 0 and (a[0] - b[0]).abs().max().item() <= 2e-6 * b[0].abs().max().item()
