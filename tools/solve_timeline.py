#!/usr/bin/env python3
"""How much of a batched solve's wall time the GPU is busy: from a rocprofv3 --kernel-trace database, the dispatches from the
first ilqr_loop_init_kernel on, grouped into solves (a gap > 1 ms starts a new one): span, busy time (kernels incl. the
runtime's fill / copy kernels), idle time, and the idle time split by what follows the gap.

    python tools/solve_timeline.py trace.db
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    first = next((i for i, r in enumerate(rows) if "ilqr_loop_init" in r[0]), None)
    if first is None:
        print("no solve in this trace")
        return
    rows = rows[first:]
    solves, cur = [], [rows[0]]
    for r in rows[1:]:
        if "ilqr_loop_init" in r[0] and (r[1] - cur[-1][2]) > 2e5:
            solves.append(cur); cur = [r]
        else:
            cur.append(r)
    solves.append(cur)
    print("%5s %8s %10s %10s %10s %7s %9s %9s" % ("solve", "kernels", "span_us", "busy_us", "idle_us", "idle%", "fill+copy", "their_us"))
    for i, s in enumerate(solves):
        span = (s[-1][2] - s[0][1]) / 1e3
        busy = sum(r[2] - r[1] for r in s) / 1e3
        rt = [r for r in s if "rocclr" in r[0]]
        print("%5d %8d %10.1f %10.1f %10.1f %6.1f%% %9d %9.1f" % (i, len(s), span, busy, span - busy, 100 * (span - busy) / span, len(rt),
                                                               sum(r[2] - r[1] for r in rt) / 1e3))
    s = solves[-1]
    gaps = {}
    for a, b in zip(s, s[1:]):
        g = (b[1] - a[2]) / 1e3
        key = b[0].split("(")[0].replace("void altro_hip::", "")[:48]
        gaps.setdefault(key, []).append(g)
    print("idle time of the last solve by the kernel that follows the gap:")
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:10]:
        print("   %-50s n %4d  total %8.1f us  median %6.2f us  max %7.2f us" % (k, len(v), sum(v), sorted(v)[len(v) // 2], max(v)))


if __name__ == "__main__":
    main(sys.argv[1])
