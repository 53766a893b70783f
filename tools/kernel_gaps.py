#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd .db): end of one dispatch to start of the next.

    python tools/kernel_gaps.py trace.db [name-substring ...]
"""
import sqlite3
import sys


def main(path, subs):
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    rows = [r for r in rows if not subs or any(s in r[0] for s in subs)]
    gaps = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        key = (n0.split("(")[0][-40:], n1.split("(")[0][-40:])
        gaps.setdefault(key, []).append((s1 - e0) / 1e3)
    print("%-42s -> %-42s %6s %10s %10s %10s" % ("after", "before", "n", "median_us", "min_us", "max_us"))
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
        v = sorted(v)
        print("%-42s -> %-42s %6d %10.2f %10.2f %10.2f" % (k[0], k[1], len(v), v[len(v) // 2], v[0], v[-1]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
