#!/usr/bin/env python
"""Per-call durations of the kernels matching a pattern, in dispatch order, for the LAST `n` dispatches
(one benchmark step), from a rocprofv3 rocpd sqlite trace.

    python tools/rocpd_calls.py gpurun_out/prof/p_results.db conv3x3 14
"""
import sqlite3
import sys


def main(path, pattern, n):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {namecol}, start, end from kernels where {namecol} like ? order by start", (f"%{pattern}%",)).fetchall()
    rows = rows[-n:]
    print(" ".join(f"{'W' if 'wide' in r[0] else ''}{(r[2] - r[1]) / 1e3:.0f}" for r in rows), " | total", f"{sum(r[2] - r[1] for r in rows) / 1e3:.0f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
