"""Recipe that makes the REAL reference (zju3dv/LoFTR, `src/loftr/**`) available on the GPU box.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Nothing under loftr_amd/ imports this module or what it builds; the
only consumers are oracle/ref_shim.py (tests) and bench.py's `cpu_baseline` leg (the reference's own CPU
`forward()` timed on the GPU box's host cores, BASELINE.json north_star).

The reference is pure Python, so "building" it means compiling it: every module of the package `src.loftr`
(the path `LoFTR.forward` imports: /root/reference/src/loftr/loftr.py:1-10) is compiled by CPython from the
sources WHERE THEY LIE under /root/reference into code objects, and the marshalled code objects are written
to ONE binary bundle, `oracle/_ref/loftr_reference.bundle`.  No reference source text enters this repository:
`oracle/_ref/` is git-ignored (history stays source-only) but not gpurun-ignored, so the bundle travels to the
GPU box with the snapshot exactly like the built `libloftr_hip.so` does.  The bundle is only valid for the
interpreter that made it (same image on both sides; the magic number is checked on load).

    python -m oracle.stage_ref            # build()  in __graft_entry__ calls stage() when /root/reference exists

`BundleFinder` is the import hook that serves `src`, `src.loftr`, ... from the bundle; ref_shim installs it
when /root/reference itself is absent (the third-party stubs -- yacs, kornia, superglue -- are ref_shim's).
"""
import hashlib
import importlib.abc
import importlib.machinery
import importlib.util
import marshal
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
BUNDLE = os.path.join(REF_DIR, "loftr_reference.bundle")
# sha256 of the bundle FILE this repository expects (tracked; the staging is deterministic for a given reference checkout and CPython):
# load_bundle() refuses a bundle whose bytes differ -- what gets exec'd on the GPU box is exactly what was compiled from /root/reference here
EXPECTED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_bundle.sha256")
REFERENCE_ROOT = os.environ.get("LOFTR_REFERENCE_ROOT", "/root/reference")
# `src/__init__.py` (empty package marker) + the whole `src.loftr` package: loftr.py, backbone/, loftr_module/, utils/
PACKAGE_DIRS = ("src/loftr",)
EXTRA_FILES = ("src/__init__.py",)


def _module_name(rel):
    parts = rel[:-3].split("/")
    is_pkg = parts[-1] == "__init__"
    if is_pkg:
        parts = parts[:-1]
    return ".".join(parts), is_pkg


def _sources(root):
    rels = [f for f in EXTRA_FILES if os.path.isfile(os.path.join(root, f))]
    for d in PACKAGE_DIRS:
        for dp, _, files in sorted(os.walk(os.path.join(root, d))):
            for f in sorted(files):
                if f.endswith(".py"):
                    rels.append(os.path.relpath(os.path.join(dp, f), root).replace(os.sep, "/"))
    return rels


def stage(root=REFERENCE_ROOT, verbose=True, update_hash=False):
    """Compile the reference's `src.loftr` package into oracle/_ref/loftr_reference.bundle.  Returns the path,
    or None when the reference is not on this machine (the GPU box: it then uses the bundle that travelled)."""
    if not os.path.isfile(os.path.join(root, "src", "loftr", "loftr.py")):
        return None
    modules, digest = {}, hashlib.sha256()
    for rel in _sources(root):
        with open(os.path.join(root, rel), "rb") as fh:
            text = fh.read()
        digest.update(rel.encode() + b"\0" + text)
        name, is_pkg = _module_name(rel)
        # the recorded file name points back at the reference checkout (tracebacks), not into this repo
        code = compile(text, f"<reference>/{rel}", "exec", dont_inherit=True, optimize=0)
        modules[name] = (is_pkg, marshal.dumps(code))
    for name in list(modules):                              # directories without __init__.py (src/loftr/utils) are namespace packages
        parts = name.split(".")
        for n in range(1, len(parts)):
            modules.setdefault(".".join(parts[:n]), (True, None))
    blob = marshal.dumps({"magic": importlib.util.MAGIC_NUMBER, "python": sys.version, "sha256": digest.hexdigest(),
                          "modules": modules})
    # The tracked oracle/ref_bundle.sha256 names the bundle's bytes AND the digest of the reference sources they were compiled from.
    # A re-stage of the SAME sources may produce other bytes (marshal is not byte-stable across processes): then the record follows
    # the new file.  Sources that differ from the recorded ones are a different reference: refuse unless told to (--update-hash) --
    # until round 5 any re-stage silently blessed whatever it had built (advisor, round 5).
    rec = open(EXPECTED).read().split() if os.path.isfile(EXPECTED) else []
    rec_sources = rec[rec.index("sha256") + 1].rstrip(",") if "sha256" in rec else None
    if rec and rec_sources is not None and rec_sources != digest.hexdigest() and not update_hash:
        raise RuntimeError(f"the reference under {root} (sources sha256 {digest.hexdigest()[:16]}) is not the one oracle/ref_bundle.sha256 records "
                           f"({rec_sources[:16]}): run `python -m oracle.stage_ref --update-hash` if that is intended")
    os.makedirs(REF_DIR, exist_ok=True)
    tmp = BUNDLE + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(blob)
    os.replace(tmp, BUNDLE)
    file_hash = hashlib.sha256(blob).hexdigest()
    if not rec or rec[0] != file_hash:
        with open(EXPECTED, "w") as fh:
            fh.write(f"{file_hash}  loftr_reference.bundle  (sources sha256 {digest.hexdigest()}, {sys.version.split()[0]})\n")
    if verbose:
        print(f"[stage_ref] {len(modules)} reference modules -> {BUNDLE} ({len(blob) / 1e3:.1f} kB, sha256 {digest.hexdigest()[:16]})")
    return BUNDLE


def bundle_available():
    return os.path.isfile(BUNDLE)


def load_bundle():
    with open(BUNDLE, "rb") as fh:
        raw = fh.read()
    want = open(EXPECTED).read().split()[0] if os.path.isfile(EXPECTED) else None
    got = hashlib.sha256(raw).hexdigest()
    if want is None or got != want:                         # verified BEFORE anything is unmarshalled or exec'd (advisor, round 4)
        raise ImportError(f"{BUNDLE}: sha256 {got[:16]} does not match the committed oracle/ref_bundle.sha256 "
                          f"({(want or 'missing')[:16]}); re-run `python -m oracle.stage_ref` where /root/reference exists")
    b = marshal.loads(raw)
    if b["magic"] != importlib.util.MAGIC_NUMBER:
        raise ImportError(f"{BUNDLE} was made by a different CPython ({b['python']}); re-run `python -m oracle.stage_ref` "
                          "where /root/reference exists")
    return b


class BundleFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Serves the modules of the bundle (`src`, `src.loftr`, `src.loftr.loftr`, ...) from their code objects."""

    def __init__(self, bundle):
        self.modules = bundle["modules"]
        self.sha256 = bundle["sha256"]

    def find_spec(self, fullname, path=None, target=None):
        if fullname not in self.modules:
            return None
        return importlib.machinery.ModuleSpec(fullname, self, is_package=self.modules[fullname][0], origin=f"{BUNDLE}:{fullname}")

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        is_pkg, raw = self.modules[module.__name__]
        if is_pkg:
            module.__path__ = []                           # sub-modules resolve through this finder, not the file system
        if raw is not None:
            exec(marshal.loads(raw), module.__dict__)


def install_finder():
    """Put the bundle's finder on sys.meta_path (idempotent).  Returns it."""
    for f in sys.meta_path:
        if isinstance(f, BundleFinder):
            return f
    f = BundleFinder(load_bundle())
    sys.meta_path.insert(0, f)
    return f


if __name__ == "__main__":
    if stage(update_hash="--update-hash" in sys.argv) is None:
        print(f"[stage_ref] no reference at {REFERENCE_ROOT}; nothing staged"
              + (f" (existing bundle kept: {BUNDLE})" if bundle_available() else ""))
