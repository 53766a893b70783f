#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel durations and per-kernel PMC counter means.

    python tools/rocpd_summary.py gpurun_out/prof_r01/trace/bench_results.db [more.db ...] > profiles/x.txt
"""
import sqlite3
import sys


def main(paths):
    for p in paths:
        con = sqlite3.connect(p)
        cur = con.cursor()
        print("=== %s" % p)
        print("%-72s %6s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
        rows = cur.execute(
            "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
            "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1.0
        for r in rows:
            print("%-72s %6d %12.1f %12.2f %12.2f %12.2f %6.1f%%" % (r[0][:72], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where name='counters_collection'")]
        if tabs:
            crow = cur.execute(
                "select kernel_name, counter_name, count(*), avg(v), min(v), max(v), max(vg), max(sg) from "
                "(select kernel_name, counter_name, dispatch_id, sum(value) as v, max(vgpr_count) as vg, "
                " max(sgpr_count) as sg from counters_collection group by kernel_name, counter_name, dispatch_id) "
                "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
            if crow:
                print("%-60s %-28s %6s %18s %18s %18s %5s %5s" % ("kernel", "counter", "disp", "mean/dispatch", "min", "max", "vgpr", "sgpr"))
                for r in crow:
                    print("%-60s %-28s %6d %18.3f %18.3f %18.3f %5s %5s" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
