"""Golden vectors for the ResNet-FPN backbone produced by the REAL reference modules
(src/loftr/backbone/resnet_fpn.py: ResNetFPN_8_2 / ResNetFPN_16_4), CPU / fp32 / eval.  Authoring container only:

    python tests/golden/make_golden_backbone.py        ->  tests/golden/backbone.npz

Weights are not stored: both sides fill their state_dict from `loftr_amd.synth.make_backbone_weights(seed, module)`
(one numpy Generator walked in state_dict order, loaded with strict=True, so the parameter layouts must agree too).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_shim import import_reference              # noqa: E402
from loftr_amd.synth import make_backbone_weights          # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {   # name: (resolution, block_dims, weight seed, input seed, input [N,1,H,W])
    "r8_2": ((8, 2), [128, 196, 256], 21, 22, (2, 1, 32, 48)),
    "r16_4": ((16, 4), [128, 196, 256, 512], 23, 24, (1, 1, 32, 64)),
}


def backbone_cfg(resolution, dims):
    return {"backbone_type": "ResNetFPN", "resolution": resolution, "resnetfpn": {"initial_dim": 128, "block_dims": dims}}


def main():
    import_reference()                                      # installs the stubs + sys.path
    from src.loftr.backbone import build_backbone
    out = {}
    for name, (res, dims, wseed, xseed, shape) in CASES.items():
        m = build_backbone(backbone_cfg(res, dims)).eval()
        m.load_state_dict(make_backbone_weights(wseed, m), strict=True)
        x = torch.rand(*shape, generator=torch.Generator().manual_seed(xseed))
        with torch.no_grad():
            c, f = m(x)
        out[f"{name}_coarse"] = c.numpy()
        out[f"{name}_fine"] = f.numpy()
        out[f"{name}_nparams"] = np.int64(sum(p.numel() for p in m.parameters()))
        print(name, tuple(c.shape), tuple(f.shape), float(c.abs().max()), float(f.abs().max()))
    np.savez_compressed(os.path.join(HERE, "backbone.npz"), **out)


if __name__ == "__main__":
    main()
