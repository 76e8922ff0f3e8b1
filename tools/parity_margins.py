"""Print the parity margins of the HIP path against every reference golden (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _cases import SMALL_CASES, MID_CASES, FULL_CASES, load_case, compare_to_golden, run_hip
print(f"{'case':22s} {'M':>6s} {'flips':>5s} {'d_mconf':>10s} {'d_mkpts1_f':>11s} {'d_expec_xy':>11s} {'d_conf_rowmax':>13s}")
for name in SMALL_CASES + MID_CASES + FULL_CASES:
    rc, inp, g = load_case(name)
    out = run_hip(inp)
    rep = compare_to_golden(out, g, inp["cfg"]["match_coarse"]["thr"], max_flips=4, tol_conf=1.0, tol_px=10.0)
    rowmax = np.abs(out["conf_matrix"].max(2) - g["conf_row_max"]).max()
    print(f"{name:22s} {rep['M_ref']:6d} {len(rep['only_out']) + len(rep['only_ref']):5d} {rep.get('d_mconf', 0):10.2e} "
          f"{rep.get('d_mkpts1_f', 0):11.2e} {rep.get('d_expec_xy', 0):11.2e} {rowmax:13.2e}")
