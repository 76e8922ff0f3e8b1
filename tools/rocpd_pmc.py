#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, rocpd sqlite).

    python tools/rocpd_pmc.py <fetch.db> <write.db> [--sq <sq.db>] [--json profiles/pmc_traffic.json]

--sq: a third pass with `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`: per kernel
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)  (GUI_ACTIVE is reported summed over the 8 XCDs; the counter
advances 32 cycles per v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md).  The JSON is stamped with the hash of the kernel
sources (bench.py:source_hash) so that bench.py only quotes PMC numbers collected on the build it is running.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  gfx950 correction (MI355X_MICROARCH.md
§HBM): FETCH_SIZE counts the 128-B requests of a wide coalesced streaming read at 64 B, i.e. it
reports HALF the bytes -> doubled here ("fetch_x2").  WRITE_SIZE is uncalibrated in the guide; it
is calibrated below against a kernel with a known write volume (score_conf_kernel writes exactly
N*L*S*4 bytes of conf_matrix) and the factor printed.
"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
                      "where counter_name=? group by kernel_name", (counter,)).fetchall()
    return {r[0]: dict(n=r[1], avg_kib=r[2], min_kib=r[3], max_kib=r[4]) for r in rows}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.split(r"[<(]", name)[0]


def bench_name(full):
    """Kernel name as bench.py's timing table knows it (one entry per kernel FAMILY: template variants are pooled)."""
    if "score_sweep_kernel<1" in full:
        return "score_sweep_kernel<1>"          # dual-softmax pass B (writes conf_matrix)
    if "score_sweep_kernel<0" in full:
        return "score_sweep_kernel<0>"         # pass A: shared-reference variant + exact variant
    if "score_sweep_kernel<2" in full:
        return "score_sweep_kernel<2>"         # Sinkhorn: score store on the sweep
    if "conv3x3_duo_kernel" in full:        # round 4: the 3x3 stride-1 convolutions; timed under the ids of the kernels they replaced
        return "conv3x3_duo_kernel<Cfg<7,2,4,8,2>>" if "Cfg<7" in full else "conv3x3_duo_kernel<Cfg<4,2,4,4,1>>"
    if "encoder_x2_kernel" in full:         # round 4: two-job launches of the same kernel body, timed under LOFTR_T_ENCODER_X
        return "encoder_x_kernel"
    if "coarse_persistent_kernel" in full:  # round 6: the whole coarse transformer as one persistent launch (X, K and F work items), timed under LOFTR_T_ENCODER_X
        return "encoder_x_kernel"
    return short(full).split("::")[-1]    # efx::encoder_x_kernel, ffx::fine_pair_kernel -> the timing table's names


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    busy, active = {}, {}
    if "--sq" in sys.argv:
        sq = sys.argv[sys.argv.index("--sq") + 1]
        busy, active = per_kernel(sq, "SQ_VALU_MFMA_BUSY_CYCLES"), per_kernel(sq, "GRBM_GUI_ACTIVE")
    pool = {}
    for k in set(fetch) | set(write):
        if not ("kernel" in k and ("loftr" in k or "Geometry" in k or "Args" in k or "anonymous" in k or "attn" in k or "kv_" in k
                                  or "gather" in k or "fine_match" in k or "pos_encode" in k)):
            continue
        n = fetch.get(k, write.get(k))["n"]
        e = pool.setdefault(bench_name(k), dict(launches=0, fetch=0.0, write=0.0, busy=0.0, active=0.0, variants=[]))
        e["launches"] += n
        e["fetch"] += fetch.get(k, {}).get("avg_kib", 0.0) * 1024 * 2 * n          # gfx950: FETCH_SIZE counts 128-B requests at 64 B
        e["write"] += write.get(k, {}).get("avg_kib", 0.0) * 1024 * n
        if k in busy and k in active:
            e["busy"] += busy[k]["avg_kib"] * busy[k]["n"]                          # (per_kernel's "avg_kib" is the plain average)
            e["active"] += active[k]["avg_kib"] * active[k]["n"]
        e["variants"].append(k[:140])
    out = {}
    print(f"{'kernel (template variants pooled)':44s} {'calls':>6} {'fetch_x2 MB':>12} {'write MB':>10} {'hbm MB/launch':>14} {'mfma_busy':>10}")
    for name, e in sorted(pool.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
        n = e["launches"]
        f, w = e["fetch"] / n, e["write"] / n
        out[name] = dict(launches=n, fetch_bytes_x2=f, write_bytes=w, hbm_bytes_per_launch=f + w, variants=e["variants"])
        mb = ""
        if e["active"] > 0:
            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (checked against kernel durations x clock), the SQ counter over
            # all 1024 SIMDs: busy fraction of a SIMD's matrix pipe = busy / (active / 8 * 1024)
            out[name]["mfma_busy"] = e["busy"] / (e["active"] / 8.0 * 1024.0)
            mb = f"{out[name]['mfma_busy']:10.3f}"
        print(f"{name:44s} {n:6d} {f / 1e6:12.2f} {w / 1e6:10.2f} {(f + w) / 1e6:14.2f} {mb}")
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    try:
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        out["_meta"] = dict(source_hash=b.source_hash(), kernel_code_hash=b.pmc_kernel_code_hash(out.keys()),
                            fetch_correction="FETCH_SIZE x2 (gfx950)", write_correction="none")
        print("# source hash", out["_meta"]["source_hash"], "kernel code hash", out["_meta"]["kernel_code_hash"])
    except Exception as e:      # noqa: BLE001
        print("# could not stamp the source hash:", e)
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
