#!/usr/bin/env python
"""gpurun_out/parity_e2e.txt (+ parity_features.txt), as written by the GPU tests, -> the parity table of a round:

    python tools/margins_table.py gpurun_out/parity_e2e.txt [gpurun_out/parity_features.txt] > profiles/r06_parity_margins.txt

Per image-level case and backbone: |d mconf| and |d mkpts1_f| against the reference's fp32 golden next to the ABSOLUTE north-star bars
(1e-4 / 1e-3 px), the reference's own fp32-vs-fp64 distance on the same input, and whether the case passes on the absolute bar or only
through the rule `tol = max(bar, 2 x the reference's own noise)` (tests/test_e2e_golden.py)."""
import re
import sys

BAR_C, BAR_P = 1e-4, 1e-3
rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+backbone=(\S+)\s+M_ref=\s*(\d+) M_out=\s*(\d+) flips=(\d+) d_mconf=(\S+) d_mkpts1_f=(\S+)px d_conf_rowmax=(\S+) \| vs ref-fp64: ours (\S+) / (\S+)px, "
                 r"ref-fp32 itself (\S+) / (\S+)px \| rms ours (\S+) / (\S+)px, ref-fp32 (\S+) / (\S+)px", ln)
    if m:
        g = m.groups()
        rows.append(dict(case=g[0], bb=g[1], M=int(g[2]), flips=int(g[4]), dc=float(g[5]), dp=float(g[6]), oc=float(g[8]), op=float(g[9]),
                         rc=float(g[10]), rp=float(g[11]), orc=float(g[12]), orp=float(g[13]), rrc=float(g[14]), rrp=float(g[15])))
latest = {}
for r in rows:                                   # a file appended to by several test runs: keep the last line per (case, backbone)
    latest[(r["case"], r["bb"])] = r
print("# Image-level parity against the reference's forward from images (tests/test_e2e_golden.py), absolute numbers.")
print("# d_* = ours vs the reference's fp32 golden; ref noise = the reference's own fp32 run vs its fp64 run on the same input;")
print("# ours->fp64 = our distance to the reference's fp64 run (max, and the RMS ratio ours / reference over the common matches).")
print(f"# bars: |d mconf| <= {BAR_C:g}, |d mkpts1_f| <= {BAR_P:g} px.  'abs' = inside the absolute bars; '2x' = passes only through tol = max(bar, 2 x ref noise)")
print(f"{'case':18s} {'backbone':8s} {'M':>5s} {'flips':>5s} | {'d_mconf':>9s} {'d_px':>9s} {'pass':>4s} | {'ref noise conf':>14s} {'px':>9s} | {'d_px / noise':>12s} | {'ours->fp64 px':>13s} {'rms ratio px':>12s} {'conf':>6s}")
worst_torch = 0.0
for (case, bb), r in latest.items():
    ok_abs = r["dc"] <= BAR_C and r["dp"] <= BAR_P
    ratio = r["dp"] / r["rp"] if r["rp"] > 0 else 0.0
    if bb == "torch" and not ok_abs:
        worst_torch = max(worst_torch, ratio)
    print(f"{case:18s} {bb:8s} {r['M']:5d} {r['flips']:5d} | {r['dc']:9.2e} {r['dp']:9.2e} {'abs' if ok_abs else '2x':>4s} | {r['rc']:14.2e} {r['rp']:9.2e} | {ratio:12.2f} | "
          f"{r['op']:13.2e} {r['orp'] / r['rrp'] if r['rrp'] else 0:12.2f} {r['orc'] / r['rrc'] if r['rrc'] else 0:6.2f}")
n2 = sum(1 for r in latest.values() if not (r["dc"] <= BAR_C and r["dp"] <= BAR_P))
print(f"# {len(latest)} (case, backbone) rows; {n2} pass only through the 2 x rule; worst d_px / ref noise of a torch-backbone (MIOpen) row outside the absolute bar: {worst_torch:.2f}")
for f in sys.argv[2:]:
    print("# ---- feature-level cases with a float64 leg (tests/test_hip_parity.py)")
    for ln in dict((l.split()[0], l) for l in open(f) if l.strip()).values():
        print(ln.rstrip())
