"""bench.py's algorithmic work table and roofline classification (host logic, CPU): the numbers behind
`roofline.achieved` must be the ones DESIGN.md §4 / SURVEY.md §8(d) state."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["bench_mod"] = m
    spec.loader.exec_module(m)
    return m


def test_backbone_conv_table(bench):
    convs = bench.backbone_convs(480, 640)
    assert len(convs) == 21                                              # ResNetFPN_8_2 minus the stem (resnet_fpn.py:43-118)
    assert sum(1 for c in convs if c[2] == 3 and c[3] == 1) == 14        # 3x3 stride-1 layers: patch kernels
    wide = [c for c in convs if c[2] == 3 and c[3] == 1 and (c[1] + 31) // 32 == 7]
    assert len(wide) == 5                                                # Cout = 196 -> 224 columns
    flops_per_image = sum(2 * (h // s) * (w // s) * cout * cin * k * k for cin, cout, k, s, h, w in convs)
    stem = 2 * 240 * 320 * 128 * 49
    # the backbone is ~595 GFLOP per PAIR (two images)
    assert abs(2 * (flops_per_image + stem) / 1e9 - 595) < 15


def test_algorithmic_work_matches_design_table(bench):
    B, L, M = 8, 4800, 7663
    w = bench.algorithmic_work(B, L, L, M)
    C = 256
    rows = 2 * B * L
    # dual-softmax GEMM passes: 2 B L S C flops each
    assert w["score_sweep_kernel<0>"][0] == 2 * B * L * L * C == w["score_sweep_kernel<1>"][0]
    assert w["score_sweep_kernel<1>"][1] == 4 * B * ((L + L) * C + L * L)     # conf_matrix written once
    # k/v projection with the fused KV reduction, 8 layer passes over both images
    assert w["proj_kv_kernel"][0] == 8 * (2 * rows * C * 2 * C + 2 * rows * C * 32)
    # encoder per layer pass and pair (6.45 GFLOP per call incl. attention): projections + merge + MLP
    for fused in (False, True):                                            # four kernels per layer call, or encoder_x_kernel (round 3)
        w0 = bench.algorithmic_work(B, L, L, 0, fused=fused)               # no matches: the coarse level alone
        per_call = sum(w0[k][0] for k in ("proj_kernel", "proj_kv_kernel", "linear_kernel", "linear_ln_kernel", "encoder_x_kernel")
                       if k in w0) / 8 / (2 * B)
        assert abs(per_call / 1e9 - 6.45) < 0.2                            # 6.45 GFLOP per encoder call
    # fused x side: x (SP + fp32) in, out (fp32 + SP) in place -- 4 tensor streams per call
    wf = bench.algorithmic_work(B, L, L, 0, fused=True)
    assert wf["encoder_x_kernel"][1] == 8 * 4 * (4 * rows * C + 8 * C * C) and wf["proj_kernel"][0] == 0
    assert w["linear_kernel"][0] > w0["linear_kernel"][0]                  # the fine level rides on the same kernels
    # the three conv entries partition the 21 convolutions of 2B images
    convs = bench.backbone_convs(480, 640)
    total = sum(2 * 2 * B * (h // s) * (ww // s) * cout * cin * k * k for cin, cout, k, s, h, ww in convs)
    assert w["conv_kernel"][0] + w["conv3x3_duo_kernel<Cfg<4,2,4,4,1>>"][0] + w["conv3x3_duo_kernel<Cfg<7,2,4,8,2>>"][0] == total
    assert w["conv3x3_duo_kernel<Cfg<7,2,4,8,2>>"][0] > 0.3 * total and w["conv3x3_duo_kernel<Cfg<4,2,4,4,1>>"][0] > 0.4 * total


def test_roofline_entry_classification(bench):
    # a GEMM kernel far from the HBM roof: matrix-pipe bound, executed rate = 3 x algorithmic
    e = bench.roofline_entry("conv3x3_duo_kernel<Cfg<4,2,4,4,1>>", total_ms=8.0, launches=9, flops=2.5e12, nbytes=8e9, steps=1)
    assert e["bound"] == "mfma" and e["unit"] == "TFLOP/s" and e["peak"] == 2500.0
    assert abs(e["executed_fp16_TFLOP_s"] - 3 * 2.5e12 / 8e-3 / 1e12) < 1 and abs(e["frac"] - e["mfma_frac"]) < 1e-9
    assert abs(e["avg_launch_us"] - 8000 / 9) < 0.01
    # a streaming kernel: HBM bound against 8 TB/s
    e = bench.roofline_entry("gather_windows_kernel", total_ms=0.1, launches=1, flops=0, nbytes=4e8, steps=1)
    assert e["bound"] == "hbm" and e["peak"] == 8000.0 and abs(e["achieved"] - 4000.0) < 1 and abs(e["frac"] - 0.5) < 1e-3
    # a GEMM kernel whose bytes dominate is reported against the HBM roof
    e = bench.roofline_entry("linear_ln_kernel", total_ms=0.1, launches=1, flops=2e10, nbytes=3.2e8, steps=1)
    assert e["bound"] == "hbm" and e["hbm_frac"] > e["mfma_frac"]
    assert bench.roofline_entry("linear_kernel", 0.0, 0, 1, 1, 1) is None


def test_single_gpu_launch_retries_once_after_a_signal(monkeypatch, capsys):
    """bench.run_with_retry: a child killed by a signal is run once more in a fresh process; an ordinary failure is not."""
    import subprocess
    import bench

    class P:
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out

    calls = []

    def fake(seq):
        it = iter(seq)

        def run(cmd, env=None, stdout=None):
            calls.append((cmd[-2:], env["LOFTR_BENCH_CHILD"], env["LOFTR_BENCH_ATTEMPT"]))
            return next(it)
        return run

    monkeypatch.setattr(subprocess, "run", fake([P(-6, b""), P(0, b'{"value": 1}\\n')]))
    assert bench.run_with_retry(["--steps", "2"]) == 0
    assert capsys.readouterr().out == '{"value": 1}\\n'
    assert [c[1:] for c in calls] == [("1", "1"), ("1", "2")] and calls[0][0] == ["--steps", "2"]
    calls.clear()
    monkeypatch.setattr(subprocess, "run", fake([P(1, b"")]))
    assert bench.run_with_retry([]) == 1 and len(calls) == 1          # Python exception: deterministic, no retry
    calls.clear()
    monkeypatch.setattr(subprocess, "run", fake([P(-11, b""), P(-6, b"")]))
    assert bench.run_with_retry([]) == 134 and len(calls) == 2         # out of attempts: 128 + signal


def test_pmc_kernel_families_follow_the_timing_table():
    """tools/rocpd_pmc.py pools rocprofv3's kernel names into the families bench.py's timing table knows: the two-job launches of the
    encoder kernel (round 4) and the round-4 3x3 convolution templates must land on the ids they are timed under, or the encoder
    group's traffic / mfma_busy silently lose half their launches."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("rocpd_pmc", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rocpd_pmc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    f = mod.bench_name
    assert f("(anonymous namespace)::efx::encoder_x2_kernel((anonymous namespace)::efx::Args2)") == "encoder_x_kernel"
    assert f("(anonymous namespace)::efx::encoder_x_kernel((anonymous namespace)::efx::Args)") == "encoder_x_kernel"
    assert f("void conv3x3_duo_kernel<c3d::Cfg<7, 2, 4, 8, 2> >(Conv3Args)") == "conv3x3_duo_kernel<Cfg<7,2,4,8,2>>"
    assert f("void conv3x3_duo_kernel<c3d::Cfg<4, 2, 4, 4, 1> >(Conv3Args)") == "conv3x3_duo_kernel<Cfg<4,2,4,4,1>>"
    assert f("void (anonymous namespace)::sweep::score_sweep_kernel<1, false, false, false>((anonymous namespace)::sweep::Args)") == "score_sweep_kernel<1>"


def test_timing_slot_names_are_the_kernels_that_run(bench):
    """Round-4 verdict (#6): the bench reported the round-4 kernels under the names of the kernels they replaced.  The library's timing slots
    (csrc/misc.hip: T_NAMES) now carry the names of what the default path launches, bench.py's tables use exactly those names, and every
    pooled family lists the rocprofv3 names it stands for."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "loftr_amd", "csrc", "misc.hip")).read()
    block = src[src.index("T_NAMES[LOFTR_T_COUNT]"):]
    slots = re.findall(r'"([^"]+)"', block[:block.index("};")])
    assert len(slots) == 15 and len(set(slots)) == 15
    for dead in ("conv3x3_kernel", "conv3x3_wide_kernel", "score_conf_kernel", "score_stats_kernel", "score_store_kernel"):
        assert dead not in slots
    for name in bench.GEMM_KERNELS + bench.BACKBONE_KERNELS[:3] + bench.ENCODER_KERNELS + bench.NORTH_STAR_TIMED:
        assert name in slots, name
    assert set(bench.POOLED_FROM) <= set(slots)
    w = bench.algorithmic_work(B=8, L=4800, S=4800, M=7600)
    assert set(w) <= set(slots) | {"attn_small_kernel"}, set(w) - set(slots)


def test_pmc_table_describes_the_kernels_of_this_build(bench):
    """profiles/pmc_traffic.json is quoted by bench.py only for the build it was collected on.  Its identity stamp is the hash of the kernel
    sources OR (round 5) of the machine code of the kernels it lists, dug out of the library's gfx950 code objects by tools/kernel_code_hash.py:
    a change to any profiled kernel's code -- and only that -- must fail here until the PMC passes are re-collected."""
    import json
    import os
    from loftr_amd import build as build_mod
    build_mod.build(verbose=False)
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    meta = t["_meta"]
    ok_source = meta.get("source_hash") == bench.source_hash()
    ok_code = meta.get("kernel_code_hash") == bench.pmc_kernel_code_hash(t.keys())
    assert ok_source or ok_code, ("profiles/pmc_traffic.json belongs to another build of the profiled kernels: re-run tools/gpu/r5_profile.sh", meta)
    assert bench.pmc_table() and bench.pmc_traffic("encoder_x_kernel") > 1e8 and bench.PMC_IDENTITY in ("source", "kernel_code")
    # the extractor sees the library's kernels: every family of the table has device functions behind it
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_code_hash", os.path.join(ROOT, "tools", "kernel_code_hash.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from loftr_amd import _lib
    funcs = m.kernel_functions(_lib.LIB_PATH)
    assert len(funcs) >= 150
    for key in t:
        if not key.startswith("_"):
            base = key.split("<")[0]
            assert any(base in n for n in funcs), key
