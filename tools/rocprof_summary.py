#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`
writes DIR/NAME_results.db) into the per-kernel table committed under profiles/.

usage: tools/rocprof_summary.py gpurun_out/prof/r_results.db [--pmc] > profiles/rNN_xxx.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    print("# source: %s" % db)
    print("# %-86s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        if len(name) > 86:
            name = name[:83] + "..."
        print("  %-86s %8d %14.1f %12.3f %7.2f" % (name, calls, total, avg, pct))
    if "--pmc" in sys.argv:
        try:
            rows = c.execute(
                "select k.name, p.counter_name, sum(p.value), count(*) from pmc_events p "
                "join kernels k on k.dispatch_id = p.dispatch_id group by 1, 2 order by 1, 2").fetchall()
        except sqlite3.Error as e:
            print("# pmc query failed: %s" % e)
            rows = []
        print("# %-70s %-28s %18s %8s" % ("kernel", "counter", "sum", "n"))
        for name, cn, v, n in rows:
            print("  %-70s %-28s %18.1f %8d" % (name[:70], cn, v, n))


if __name__ == "__main__":
    main()
