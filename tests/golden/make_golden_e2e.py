"""Image-level end-to-end goldens: the REAL reference's ``LoFTR.forward`` from images to ``mkpts*_f``.

Run in the authoring container only (needs /root/reference and PIL):

    python tests/golden/make_golden_e2e.py            # all cases
    python tests/golden/make_golden_e2e.py scannet    # one case

Unlike ``make_golden.py`` (which swaps the backbone for synthetic feature maps) nothing is replaced here:
the reference's own ResNet-FPN, transformers and matchers run on CPU / fp32 / eval / no_grad
(/root/reference/src/loftr/loftr.py:29-75), exactly what notebooks/demo_single_pair.ipynb cells 3-4 do.

Cases (BASELINE.json configs[0] / configs[1]):
  * ``e2e_scannet``   -- assets/scannet_sample_images/scene0711_00_frame-001680.jpg / -001995.jpg, grayscale,
                         resized to 640x480 (PIL bilinear; the notebook uses cv2 -- irrelevant, both sides of the
                         parity test see the SAME uint8 bytes, which are stored in the .npz), ``/ 255``.
  * ``e2e_synth``     -- pair 0 of the bench's batch: ``loftr_amd.synth.make_images(1234, 8, 480, 640)``
                         (regenerated from the seed on the GPU box; checksums stored).
  * ``e2e_synth_bn06`` -- the same pair through a backbone with more strongly randomised BatchNorm statistics:
                         fewer (~400) but 10x more confident matches (conf up to 0.2) and larger feature magnitudes.
                         (bn_strength 1.0 gives conf up to 0.94 but there the reference's OWN fp32 forward already sits
                         1.25e-3 px from its fp64 forward -- sim values ~1e3 in the fine correlation -- so no fp32
                         implementation can be asked for 1e-3 px on it.)
Weights: there is no checkpoint on the box, so every parameter is seeded -- the matcher from
``synth.make_weights(0)`` and the backbone (incl. non-trivial BatchNorm statistics, spread ``bn_strength``)
from ``synth.make_backbone_weights(7)``; ``temp_bug_fix=True`` (indoor_ds_new / notebook cell 3).
Each case is run at thr 0.0 (M ~ 1e2..1e3) and at the stock thr 0.2 (M = 0 with random weights: the empty path).

Stored per case: the input images (uint8 for the JPEG pair), matches / key points / confidences for both
thresholds, digests of conf_matrix, checksums of the backbone outputs, the reference's CPU wall time, and
(``ref64/*``) the matches of the SAME reference module run in float64 (``model.double()``): the distance between
the two is the reference's own rounding noise on that input, the context for reading the parity margins.
"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from loftr_amd.config import get_cfg                                           # noqa: E402
from loftr_amd.synth import make_weights, make_backbone_weights, make_images, checksum   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
H_IMG, W_IMG = 480, 640
BACKBONE_SEED, MATCHER_SEED = 7, 0
SCANNET = ("assets/scannet_sample_images/scene0711_00_frame-001680.jpg",
           "assets/scannet_sample_images/scene0711_00_frame-001995.jpg")
KEEP = ("b_ids", "i_ids", "j_ids", "mkpts0_c", "mkpts1_c", "mconf", "expec_f", "mkpts0_f", "mkpts1_f")


def e2e_cfg(thr, rc=None):
    cfg = get_cfg(thr=thr)
    cfg["coarse"]["temp_bug_fix"] = bool((rc or {}).get("temp_bug_fix", True))     # outdoor_ds.ckpt: False (configs/loftr/outdoor/buggy_pos_enc/loftr_ds.py:3-4)
    if rc and "border_rm" in rc:
        cfg["match_coarse"]["border_rm"] = int(rc["border_rm"])
    if rc and rc.get("match_type") == "sinkhorn":        # configs/loftr/indoor/loftr_ot.py + default.py:29-36 (BASELINE configs[4])
        cfg["match_coarse"].update(match_type="sinkhorn", skh_prefilter=False, sparse_spvs=True)
    if rc and rc.get("resolution"):                      # ResNetFPN_16_4: coarse map at 1/16, fine at 1/4 (resnet_fpn.py:121-199)
        cfg["resolution"] = tuple(rc["resolution"])
        cfg["resnetfpn"] = {"initial_dim": 128, "block_dims": list(rc["block_dims"])}
    return cfg


CASES = {"e2e_scannet": dict(images="scannet", bn_strength=0.3),
         "e2e_synth": dict(images="synth", bn_strength=0.3),
         "e2e_synth_bn06": dict(images="synth", bn_strength=0.6),
         # images of DIFFERENT sizes: the reference runs its backbone once per image (loftr.py:48-49) and L != S everywhere after
         "e2e_unequal": dict(images="synth", bn_strength=0.3, crop0=(384, 512), crop1=(320, 448)),
         # MegaDepth-style batch: zero-padded bottom / right, coarse padding masks, scale0 / scale1 (dataset.py:72-118; exercises
         # coarse_matching.py:28-43,115-118,243-244, linear_attention.py:35-39, fine_matching.py:68 from images)
         "e2e_masked": dict(images="synth", bn_strength=0.3, crop0=(384, 512), crop1=(384, 512), valid0=(320, 512), valid1=(384, 400),
                            scale0=(1.9, 1.9), scale1=(1.25, 1.5)),
         # the other backbone the reference ships: ResNetFPN_16_4 (coarse 1/16 = 24 x 32 cells, fine 1/4)
         # Sinkhorn matching from images (conf_matrix_with_bin is produced as well: sparse_spvs)
         "e2e_ot": dict(images="synth", bn_strength=0.3, crop0=(384, 512), crop1=(384, 512), match_type="sinkhorn"),
         "e2e_r16_4": dict(images="synth", bn_strength=0.3, crop0=(384, 512), crop1=(384, 512), resolution=(16, 4),
                           block_dims=(128, 128, 196, 256)),
         # round 3, "trained-like" statistics from IMAGES: the coarse head of the backbone (layer3_outconv) is scaled so that the
         # residual stream of the random-weight transformer dominates its updates; image1 is image0 shifted by whole coarse cells, so
         # corresponding cells keep near-identical descriptors: conf close to 1 and hundreds of matches at the STOCK threshold 0.2
         "e2e_peaked": dict(images="synth", bn_strength=0.3, crop0=(384, 512), crop1=(384, 512), coarse_gain=6.0),
         # round 5 (round-4 verdict, missing #2).  A BATCH from images: pairs 0..2 of the bench's batch in one forward -- catches
         # batch-position mistakes in the backbone and the image-level plumbing that the feature-level invariance tests cannot see.
         "e2e_batch": dict(images="synth", bn_strength=0.3, n=3),
         # BASELINE configs[3] from images, exactly bench.py:other_configs' workload: 2 pairs of 840 x 840, valid region 560 rows x 840
         # columns zero-padded at the bottom, mask0 / mask1 [2, 105, 105], scale 1.9, temp_bug_fix=False (outdoor_ds.ckpt), border_rm 2
         # (configs/loftr/outdoor/loftr_ds.py:1-5, coarse_matching.py:28-43,115-118, fine_matching.py:68); L = S = 11 025
         "e2e_outdoor_840": dict(images="synth", bn_strength=0.3, n=2, size=(840, 840), valid0=(560, 840), valid1=(560, 840),
                                 scale0=(1.9, 1.9), scale1=(1.9, 1.9), temp_bug_fix=False, border_rm=2),
         # round 6 (round-5 verdict, next #2): the "trained-like" peaked regime (the e2e_peaked recipe: coarse_gain 6) on the OTHER
         # heads and shapes -- the masked 840 x 840 outdoor batch, indoor Sinkhorn, and a 3-pair batch: real checkpoints produce
         # exactly this regime (conf close to 1, hundreds of matches at the stock threshold) and one 640 x 480 dual-softmax pair was
         # the only image-level case that covered it
         "e2e_peaked_outdoor": dict(images="synth", bn_strength=0.3, n=2, size=(840, 840), valid0=(560, 840), valid1=(560, 840),
                                    scale0=(1.9, 1.9), scale1=(1.9, 1.9), temp_bug_fix=False, border_rm=2, coarse_gain=6.0),
         "e2e_peaked_ot": dict(images="synth", bn_strength=0.3, crop0=(384, 512), crop1=(384, 512), match_type="sinkhorn", coarse_gain=19.0),     # (Sinkhorn scores carry no 1 / temperature: 19^2 = 6^2 * 10 gives the spread of the dual-softmax cases)
         "e2e_peaked_batch": dict(images="synth", bn_strength=0.3, n=3, coarse_gain=6.0)}


def e2e_state_dict(module_with_backbone, cfg, bn_strength, coarse_gain=1.0, fine_gain=1.0):
    """Seeded full state_dict (torch tensors): matcher weights + backbone weights / BN statistics.  coarse_gain multiplies the
    1x1 convolution that produces the coarse map (layer3_outconv, resnet_fpn.py:64), i.e. feat_c itself; fine_gain the last
    convolution of the fine branch (layer1_outconv2[3], resnet_fpn.py:84), i.e. feat_f."""
    sd = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v))) for k, v in make_weights(MATCHER_SEED, cfg).items()}
    gains = {"layer3_outconv.weight": coarse_gain, "layer1_outconv2.3.weight": fine_gain}
    for k, v in make_backbone_weights(BACKBONE_SEED, module_with_backbone.backbone, bn_strength).items():
        sd["backbone." + k] = v * gains[k] if gains.get(k, 1.0) != 1.0 else v
    return sd


def load_images(name):
    """(image0, image1) float32 [1,1,480,640] in [0,1] + what has to be stored to regenerate them on the GPU box."""
    if CASES[name]["images"] == "scannet":
        from PIL import Image
        from oracle.ref_shim import REFERENCE_ROOT
        u8 = [np.asarray(Image.open(os.path.join(REFERENCE_ROOT, p)).convert("L").resize((W_IMG, H_IMG), Image.BILINEAR))
              for p in SCANNET]
        imgs = [(a.astype(np.float32) / np.float32(255.0))[None, None] for a in u8]
        return imgs[0], imgs[1], dict(image0_u8=u8[0], image1_u8=u8[1])
    i0, i1 = synth_images(CASES[name])
    return i0, i1, dict(image_checksums=np.array([checksum(i0), checksum(i1)]))


def synth_images(rc):
    """The first rc['n'] (default 1) pairs of the bench's rank-0 batch (480 x 640), cropped; or of make_images(1234, n, *rc['size'])
    (the bench's outdoor workload)."""
    n = int(rc.get("n", 1))
    if rc.get("size"):                                   # (the recipes' "hw" is the constant 480 x 640 of the older cases)
        i0, i1 = make_images(1234, n, int(rc["size"][0]), int(rc["size"][1]))
    else:
        i0, i1 = make_images(1234, 8, H_IMG, W_IMG)
    return _crop(i0[:n], rc.get("crop0")), _crop(i1[:n], rc.get("crop1"))


def _crop(img, hw):
    return np.ascontiguousarray(img if hw is None else img[:, :, :hw[0], :hw[1]])


def extras(rc, img0, img1):
    """mask0 / mask1 [1, H/8, W/8] bool and scale0 / scale1 [1, 2] of a masked case (numpy), images zeroed outside the valid
    rectangle like pad_bottom_right does (dataset.py:72-89); {} for the other cases."""
    if "valid0" not in rc:
        return {}
    out = {}
    for tag, img in (("0", img0), ("1", img1)):
        vh, vw = rc["valid" + tag]
        img[:, :, vh:, :] = 0
        img[:, :, :, vw:] = 0
        m = np.zeros((img.shape[0], img.shape[2] // 8, img.shape[3] // 8), bool)
        m[:, :vh // 8, :vw // 8] = True
        out["mask" + tag] = m
        out["scale" + tag] = np.asarray([rc["scale" + tag]] * img.shape[0], np.float32)
    return out


def images_from_golden(g):
    """Inverse of the storage above (used by the tests on boxes without the reference / the JPEGs)."""
    if "image0_u8" in g:
        return tuple((np.asarray(g[k]).astype(np.float32) / np.float32(255.0))[None, None] for k in ("image0_u8", "image1_u8"))
    rc = json.loads(str(g["recipe"]))
    i0, i1 = synth_images(rc)
    want = np.asarray(g["image_checksums"])
    assert np.allclose([checksum(i0), checksum(i1)], want, rtol=1e-12), "synthetic images drifted"
    return i0, i1


def run_reference(img0, img1, thr, bn_strength, timing=False, dtype=torch.float32, extra=None, rc=None):
    from oracle.ref_shim import import_reference
    from tests.golden.make_golden import conf_digest
    RefLoFTR, _ = import_reference()
    cfg = e2e_cfg(thr, rc)
    model = RefLoFTR(copy.deepcopy(cfg)).eval()
    model.load_state_dict(e2e_state_dict(model, cfg, bn_strength, (rc or {}).get("coarse_gain", 1.0)), strict=True)
    model = model.to(dtype)
    grabbed = {}
    model.backbone.register_forward_hook(lambda m, a, out: grabbed.update(feat_c=out[0].numpy().copy(), feat_f=out[1].numpy().copy()))
    secs = []
    for _ in range(3 if timing else 1):
        data = {"image0": torch.from_numpy(img0).to(dtype), "image1": torch.from_numpy(img1).to(dtype)}
        for k, v in (extra or {}).items():
            data[k] = torch.from_numpy(v) if v.dtype == bool else torch.from_numpy(v).to(dtype)
        t0 = time.perf_counter()
        with torch.no_grad():
            model(data)
        secs.append(time.perf_counter() - t0)
    out = {k: data[k].numpy() for k in KEEP}
    out.update(conf_digest(data["conf_matrix"].numpy()))
    out["feat_c_checksum"] = np.float64(checksum(grabbed["feat_c"]))
    out["feat_f_checksum"] = np.float64(checksum(grabbed["feat_f"]))
    out["feat_c_absmax"] = np.float32(np.abs(grabbed["feat_c"]).max())
    out["feat_f_absmax"] = np.float32(np.abs(grabbed["feat_f"]).max())
    out["ref_cpu_seconds"] = np.asarray(secs, np.float64)
    return out


def make(name):
    img0, img1, store = load_images(name)
    ex = extras(CASES[name], img0, img1)
    for tag, thr in (("thr0", 0.0), ("thr02", 0.2)):
        out = run_reference(img0, img1, thr, CASES[name]["bn_strength"], timing=(tag == "thr0"), extra=ex, rc=CASES[name])
        for k, v in out.items():
            if tag == "thr02" and (k.startswith("conf_") or k.startswith("feat_") or k == "ref_cpu_seconds"):
                continue          # conf_matrix / features do not depend on the threshold
            store[f"{tag}/{k}" if k in KEEP else k] = v
        print(f"{name} {tag}: M={len(out['mconf'])} conf.max={out['conf_row_max'].max():.4f} "
              f"|feat_c|max={out['feat_c_absmax']:.2f} ref CPU {np.median(out['ref_cpu_seconds']):.1f}s on {os.cpu_count()} vCPU")
    out64 = run_reference(img0, img1, 0.0, CASES[name]["bn_strength"], dtype=torch.float64, extra=ex, rc=CASES[name])
    for k in KEEP:
        store[f"ref64/{k}"] = out64[k]
    k32 = list(zip(store["thr0/b_ids"].tolist(), store["thr0/i_ids"].tolist(), store["thr0/j_ids"].tolist()))
    k64 = {k: n for n, k in enumerate(zip(out64["b_ids"].tolist(), out64["i_ids"].tolist(), out64["j_ids"].tolist()))}
    com = [(n, k64[k]) for n, k in enumerate(k32) if k in k64]
    ia, ib = [c[0] for c in com], [c[1] for c in com]
    print(f"{name} ref fp32 vs ref fp64: common {len(com)}/{len(k32)}/{len(k64)} "
          f"d_mconf={np.abs(store['thr0/mconf'][ia] - out64['mconf'][ib]).max():.2e} "
          f"d_mkpts1_f={np.abs(store['thr0/mkpts1_f'][ia] - out64['mkpts1_f'][ib]).max():.2e}px")
    store["recipe"] = np.array(json.dumps({**dict(name=name, hw=[H_IMG, W_IMG], backbone_seed=BACKBONE_SEED, matcher_seed=MATCHER_SEED,
                                                  temp_bug_fix=True, thr=[0.0, 0.2], ref_cpu_count=os.cpu_count()), **CASES[name]}))
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **store)
    print(f"-> {path} {os.path.getsize(path) / 1e3:.1f} kB")


if __name__ == "__main__":
    names = [("e2e_" + a if not a.startswith("e2e_") else a) for a in sys.argv[1:]] or list(CASES)
    for nm in names:
        make(nm)
