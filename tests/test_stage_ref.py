"""oracle/stage_ref.py: the bytecode bundle of the reference (what bench.py's cpu_baseline leg imports on the GPU box)
behaves exactly like the checkout it was compiled from."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, stage_ref                     # noqa: E402

PROBE = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
from oracle import ref_shim
assert ref_shim.reference_mode() == %r, ref_shim.reference_mode()
LoFTR, cfg = ref_shim.import_reference()
torch.manual_seed(0)
net = LoFTR(cfg).eval()
g = torch.Generator().manual_seed(1)
d = {"image0": torch.rand(1, 1, 64, 96, generator=g), "image1": torch.rand(1, 1, 64, 96, generator=g)}
cfg["match_coarse"]["thr"] = 0.0
with torch.no_grad():
    net(d)
h = hashlib.sha256()
for k in ("conf_matrix", "mkpts0_f", "mkpts1_f", "mconf"):
    h.update(d[k].numpy().tobytes())
print(len(net.state_dict()), h.hexdigest())
"""


def _run(mode, env_root):
    env = dict(os.environ, LOFTR_REFERENCE_ROOT=env_root)
    r = subprocess.run([sys.executable, "-c", PROBE % (ROOT, mode)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


@pytest.mark.skipif(not ref_shim.reference_available(), reason="needs the reference checkout (authoring container)")
def test_bundle_matches_checkout(tmp_path):
    assert stage_ref.stage(verbose=False) == stage_ref.BUNDLE
    a = _run("source", ref_shim.REFERENCE_ROOT)
    b = _run("bundle", str(tmp_path / "no_reference_here"))
    assert a == b and a.split()[0] == "211"


def test_bundle_loads_when_present():
    if not stage_ref.bundle_available():
        pytest.skip("no staged bundle on this machine")
    b = stage_ref.load_bundle()
    assert "src.loftr.loftr" in b["modules"] and "src.loftr.utils" in b["modules"]       # incl. the namespace package


def test_tampered_bundle_is_refused(monkeypatch, tmp_path):
    """The bundle's bytes are checked against the committed oracle/ref_bundle.sha256 BEFORE anything is unmarshalled or exec'd."""
    if not stage_ref.bundle_available():
        pytest.skip("no staged bundle on this machine")
    raw = open(stage_ref.BUNDLE, "rb").read()
    bad = tmp_path / "loftr_reference.bundle"
    bad.write_bytes(raw[:-1] + bytes([raw[-1] ^ 1]))
    monkeypatch.setattr(stage_ref, "BUNDLE", str(bad))
    with pytest.raises(ImportError, match="sha256"):
        stage_ref.load_bundle()
