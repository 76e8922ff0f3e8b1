#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max / %.

    python tools/rocpd_summary.py gpurun_out/prof/r01_results.db [naive_conv] > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main(path, skip_until=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    where = ""
    note = ""
    if skip_until:
        # drop everything up to the last dispatch of a kernel whose name contains `skip_until`
        # (MIOpen's find-mode trial kernels during warm-up would otherwise drown the summary)
        t = db.execute(f"select max(end) from kernels where {namecol} like ?", (f"%{skip_until}%",)).fetchone()[0]
        if t:
            where = f"where start > {t}"
            note = f"   [dispatches after the last '{skip_until}' kernel only]"
    rows = db.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels {where} group by {namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# source: {path}   (rocprofv3 --kernel-trace --stats; durations in microseconds){note}")
    print(f"# total GPU kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
    for n, c, t, a, mn, mx in rows:
        print(f"{c:7d} {t / 1e3:12.1f} {a / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * t / total:6.2f}  {short(n)}")
    coverage(db, where)


def coverage(db, where):
    """GPU coverage over the LAST 40 % of the trace (the timed steps; warm-up and instrumented steps come first): union of the kernel
    intervals / wall span, and the idle gaps by size -- the launch-gap evidence DESIGN.md §6b quotes."""
    iv = sorted(db.execute(f"select start, end from kernels {where}").fetchall())
    if len(iv) < 10:
        return
    t_lo = iv[0][0] + 0.6 * (iv[-1][1] - iv[0][0])
    iv = [(a, b) for a, b in iv if a >= t_lo]
    busy, gaps, cur_a, cur_b = 0, [], iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_b:
            busy += cur_b - cur_a
            gaps.append(a - cur_b)
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    span = iv[-1][1] - iv[0][0]
    big = sorted(gaps, reverse=True)[:8]
    print(f"# coverage over the last 40 % of the trace ({span / 1e6:.1f} ms, {len(iv)} dispatches): kernels cover {100 * busy / span:.2f} % of the wall time; "
          f"{len(gaps)} gaps, total {sum(gaps) / 1e6:.3f} ms, > 20 us: {sum(1 for g in gaps if g > 20000)} (sum {sum(g for g in gaps if g > 20000) / 1e6:.3f} ms); "
          f"largest [us]: {[round(g / 1e3, 1) for g in big]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
