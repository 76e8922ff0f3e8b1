"""Build libloftr_hip.so (the C-ABI of the HIP kernels) in-tree for gfx950.

    python -m loftr_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared object lands next to this file so that it
travels with the repo snapshot to the GPU box (a JIT cache would not).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libloftr_hip.so")
SOURCES = ["linear.hip", "coarse_plan.hip", "attention.hip", "transformer.hip", "coarse_match.hip", "fine.hip", "misc.hip",
           "sp_convert.hip", "conv.hip", "eval.hip", "input.hip", "comm.hip", "pose.hip", "train.hip", "train_bwd.hip", "encoder_fused.hip", "fine_fused.hip", "head_grads.hip", "encoder_bwd.hip", "fine_bwd.hip", "train_glue.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-parameter"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps_mtime():
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    paths.append(os.path.join(os.path.dirname(HERE), "include", "loftr_hip.h"))
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True, extra_flags=(), lib_path=None, obj_dir=None):
    """Compile every HIP source for gfx950 and link libloftr_hip.so.  Returns its path.
    extra_flags / lib_path / obj_dir build a variant next to the product library (A/B experiments)."""
    global LIB_PATH, OBJ_DIR
    lib_default, obj_default = LIB_PATH, OBJ_DIR
    if lib_path:
        LIB_PATH, OBJ_DIR, force = lib_path, obj_dir or (OBJ_DIR + "_variant"), True
    try:
        return _build(force, verbose, tuple(extra_flags))
    finally:
        LIB_PATH, OBJ_DIR = lib_default, obj_default


def _build(force, verbose, extra_flags):
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= _deps_mtime():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # export the C ABI only (include/loftr_hip.h: loftr_*); the C++ launch helpers shared between translation units stay internal
    vs = os.path.join(OBJ_DIR, "exports.map")
    with open(vs, "w") as fh:
        fh.write("{ global: loftr_*; local: *; };\n")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={vs}", "-o", LIB_PATH, *objs, "-ldl"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"built {LIB_PATH} ({os.path.getsize(LIB_PATH) / 1e6:.2f} MB)")
    return LIB_PATH


if __name__ == "__main__":
    # python -m loftr_amd.build [--force] [--variant NAME -DFLAG ...]  ->  libloftr_hip_NAME.so
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        name = sys.argv[i + 1]
        build(extra_flags=[a for a in sys.argv[i + 2:] if a.startswith("-")],
              lib_path=os.path.join(HERE, f"libloftr_hip_{name}.so"), obj_dir=os.path.join(CSRC, "build", f"variant_{name}"))
    else:
        build(force="--force" in sys.argv)
