"""Import the *real* reference (zju3dv/LoFTR at /root/reference) on a box without kornia/yacs.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py (to generate the committed golden
vectors), by tests that compare against the reference (skipped when it is absent) and by bench.py's
`cpu_baseline` leg.  /root/reference exists only in the authoring container; on the GPU box `import_reference()`
falls back to the bytecode bundle that oracle/stage_ref.py compiled from it (oracle/_ref/, git-ignored, travels
with the snapshot like the built .so).  Nothing under loftr_amd/ imports this module.

The reference imports three things that are not installed here:
  * yacs.config.CfgNode                       (src/loftr/utils/cvpr_ds_config.py:1)
  * kornia.geometry.subpix.dsnt.spatial_expectation2d, kornia.utils.grid.create_meshgrid
                                               (src/loftr/utils/fine_matching.py:5-6,49-50)
  * src.loftr.utils.superglue.log_optimal_transport (git-ignored third-party file,
                                               src/loftr/utils/coarse_matching.py:75-79)
They are replaced by in-memory stub modules that restate the published semantics
(kornia 0.4.1; SuperGlue master).  No reference source is copied.
"""
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("LOFTR_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "loftr", "loftr.py"))


class _CfgNode(dict):
    """Enough of yacs.config.CfgNode for cvpr_ds_config.py: attribute access on a dict."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _create_meshgrid(height, width, normalized_coordinates=True, device=None):
    xs = torch.linspace(0, width - 1, width, device=device)
    ys = torch.linspace(0, height - 1, height, device=device)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], -1)[None]          # [1,H,W,2], (x,y)


def _spatial_expectation2d(inp, normalized_coordinates=True):
    b, c, h, w = inp.shape
    grid = _create_meshgrid(h, w, normalized_coordinates, inp.device).to(inp.dtype)
    px = grid[..., 0].reshape(1, 1, -1)
    py = grid[..., 1].reshape(1, 1, -1)
    flat = inp.reshape(b, c, -1)
    return torch.stack([(flat * px).sum(-1), (flat * py).sum(-1)], -1)


def _log_optimal_transport(scores, alpha, iters):
    """SuperGlue's log_optimal_transport, restated (see oracle/loftr_oracle.py)."""
    import math
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    alpha_ = alpha.expand(b, 1, 1)
    couplings = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, alpha_], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])[None].expand(b, -1)
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])[None].expand(b, -1)
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(couplings + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(couplings + u.unsqueeze(2), dim=1)
    return couplings + u.unsqueeze(2) + v.unsqueeze(1) - norm


def _install_stubs():
    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    mod("yacs")
    mod("yacs.config", CfgNode=_CfgNode)
    dsnt = mod("kornia.geometry.subpix.dsnt", spatial_expectation2d=_spatial_expectation2d)
    mod("kornia")
    mod("kornia.geometry")
    mod("kornia.geometry.subpix", dsnt=dsnt)
    mod("kornia.utils")
    mod("kornia.utils.grid", create_meshgrid=_create_meshgrid)
    # The installed `transformers` package carries a line-equivalent log_optimal_transport;
    # prefer it when importable so the OT oracle is pinned against independent code.
    lot = _log_optimal_transport
    try:
        from transformers.models.superglue.modeling_superglue import log_optimal_transport as _hf
        lot = _hf
    except Exception:
        pass
    mod("src.loftr.utils.superglue", log_optimal_transport=lot)


def reference_mode():
    """'source' (the checkout at REFERENCE_ROOT), 'bundle' (oracle/_ref/loftr_reference.bundle, made from it by
    oracle/stage_ref.py) or None."""
    if reference_available():
        return "source"
    from oracle import stage_ref
    return "bundle" if stage_ref.bundle_available() else None


def import_reference():
    """Returns (LoFTR class, default_cfg) of the real reference: from the checkout when it is on this machine,
    otherwise from the staged bytecode bundle (same code objects, compiled from those very files)."""
    mode = reference_mode()
    if mode is None:
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT} and no staged bundle under oracle/_ref/ "
                           "(run `python -m oracle.stage_ref` where the reference exists)")
    _install_stubs()
    if mode == "source":
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
    else:
        from oracle import stage_ref
        stage_ref.install_finder()
    from src.loftr import LoFTR, default_cfg  # noqa
    return LoFTR, default_cfg


# ---- evaluation caller (src/utils/metrics.py): SURVEY.md §8(f) rank 2 ---------------------------
def _cross_product_matrix(x):
    """kornia 0.4.1 geometry.epipolar.numeric.cross_product_matrix: [*,3] -> [*,3,3] skew matrices."""
    x0, x1, x2 = x[..., 0], x[..., 1], x[..., 2]
    z = torch.zeros_like(x0)
    return torch.stack([z, -x2, x1, x2, z, -x0, -x1, x0, z], dim=-1).view(*x.shape[:-1], 3, 3)


def _convert_points_to_homogeneous(p):
    """kornia 0.4.1 geometry.conversions.convert_points_to_homogeneous: pad the last dim with 1."""
    return torch.nn.functional.pad(p, [0, 1], "constant", 1.0)


def import_reference_metrics():
    """The real `src.utils.metrics` module.  cv2 / loguru / kornia are absent here: loguru and cv2 are only
    touched by the RANSAC pose path and the logger (stubbed with modules that raise on use / swallow logs),
    the two kornia helpers used by the epipolar error are restated above (published semantics)."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")

    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    if "cv2" not in sys.modules:
        mod("cv2")                                   # any attribute access (findEssentialMat ...) raises AttributeError
    mod("loguru", logger=_Logger())
    mod("kornia")
    mod("kornia.geometry")
    numeric = mod("kornia.geometry.epipolar.numeric", cross_product_matrix=_cross_product_matrix)
    mod("kornia.geometry.epipolar", numeric=numeric)
    mod("kornia.geometry.conversions", convert_points_to_homogeneous=_convert_points_to_homogeneous)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    return importlib.import_module("src.utils.metrics")


# ---- input wire format (src/utils/dataset.py): SURVEY.md §8(f) rank 3 ---------------------------
def import_reference_dataset_utils():
    """The real `src.utils.dataset` module.  cv2 / h5py / loguru are absent: they are only touched by the file
    readers (imread / imdecode / resize, h5py.File), which are NOT part of what is pinned here -- the stubs raise
    on use.  What is pinned: get_resized_wh, get_divisible_wh, pad_bottom_right (pure numpy)."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    for name in ("cv2", "h5py"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "loguru" not in sys.modules:
        m = types.ModuleType("loguru")
        m.logger = _Logger()
        sys.modules["loguru"] = m
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    return importlib.import_module("src.utils.dataset")


# ---- training-side consumers (src/loftr/utils/supervision.py, src/losses/loftr_loss.py): SURVEY.md §8(f) rank 4 ----
def import_reference_training():
    """(supervision module, LoFTRLoss class) of the real reference.  loguru is stubbed (logger only); kornia's
    create_meshgrid is the restatement above (published semantics)."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    _install_stubs()

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    if "loguru" not in sys.modules:
        m = types.ModuleType("loguru")
        m.logger = _Logger()
        sys.modules["loguru"] = m
    sys.modules["kornia.utils"].create_meshgrid = _create_meshgrid
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    sup = importlib.import_module("src.loftr.utils.supervision")
    loss = importlib.import_module("src.losses.loftr_loss")
    return sup, loss.LoFTRLoss
