"""Matcher configuration: a plain nested dict with the reference's lower-cased keys.

Mirrors ``src/loftr/utils/cvpr_ds_config.py:10-50`` (``default_cfg``) of zju3dv/LoFTR, without
yacs.  ``full_default_cfg`` mirrors the LOFTR sub-tree of ``src/config/default.py:1-45`` which
differs in three defaults: temp_bug_fix True, skh_prefilter False, and the
extra key sparse_spvs.
"""
import copy

default_cfg = {
    "backbone_type": "ResNetFPN",
    "resolution": (8, 2),
    "fine_window_size": 5,
    "fine_concat_coarse_feat": True,
    "resnetfpn": {"initial_dim": 128, "block_dims": [128, 196, 256]},
    "coarse": {
        "d_model": 256, "d_ffn": 256, "nhead": 8,
        "layer_names": ["self", "cross"] * 4,
        "attention": "linear", "temp_bug_fix": False,
    },
    "match_coarse": {
        "thr": 0.2, "border_rm": 2, "match_type": "dual_softmax",
        "dsmax_temperature": 0.1, "skh_iters": 3, "skh_init_bin_score": 1.0,
        "skh_prefilter": True, "train_coarse_percent": 0.4, "train_pad_num_gt_min": 200,
    },
    "fine": {
        "d_model": 128, "d_ffn": 128, "nhead": 8,
        "layer_names": ["self", "cross"] * 1, "attention": "linear",
    },
}


def get_cfg(**match_coarse_overrides):
    """Deep copy of ``default_cfg`` with ``match_coarse`` keys overridden."""
    cfg = copy.deepcopy(default_cfg)
    cfg["match_coarse"].update(match_coarse_overrides)
    return cfg


def full_default_cfg():
    """LOFTR defaults of src/config/default.py (what test.py/train.py hand to the matcher)."""
    cfg = copy.deepcopy(default_cfg)
    cfg["coarse"]["temp_bug_fix"] = True
    cfg["match_coarse"].update(skh_prefilter=False, train_coarse_percent=0.2, sparse_spvs=True)
    return cfg
