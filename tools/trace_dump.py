#!/usr/bin/env python3
"""Launch-by-launch timeline of the LAST repetition in a rocprofv3 kernel trace (rocpd database).

usage: tools/trace_dump.py r_results.db <marker-substring> [max_rows]
A repetition = the launches from one occurrence of the marker kernel (a kernel that runs exactly once per step,
e.g. `charbonnier_partial`) to the next.  Prints, per launch: stream, start offset (us), duration (us), the idle
gap to the previous launch on the same stream, workgroups, and the kernel name -- the view that shows whether a
step is bound by launch gaps, by under-filled grids or by the side stream.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("dvsr::", "")
    return re.sub(r"\(.*$", "", name)[:60]


def main():
    c = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2]
    max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    gcols = [x for x in ("grid_x", "grid_size_x", "grid_size") if x in cols]
    wcols = [x for x in ("workgroup_x", "workgroup_size_x", "workgroup_size") if x in cols]
    sel = "name, start, end, %s, %s, %s" % (qcol, gcols[0] if gcols else "0", wcols[0] if wcols else "1")
    rows = c.execute("select %s from kernels order by start" % sel).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        print("# marker '%s' seen %d times; need >= 2" % (marker, len(marks)))
        return
    a, b = marks[-2], marks[-1]
    rows = rows[a:b]
    t0 = rows[0][1]
    print("# repetition of %d launches, span %.1f us" % (len(rows), (max(r[2] for r in rows) - t0) / 1e3))
    last_end = {}
    busy = {}
    print("# %-6s %10s %9s %8s %7s  %s" % ("stream", "start_us", "dur_us", "gap_us", "wgs", "kernel"))
    for n, (name, s, e, q, g, w) in enumerate(rows):
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
        if n < max_rows:
            print("  %-6s %10.1f %9.1f %8.1f %7d  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, (g // w) if w else 0, short(name)))
    for q, v in busy.items():
        print("# stream %s busy %.1f us" % (q, v))


if __name__ == "__main__":
    main()
