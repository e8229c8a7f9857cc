#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 --pmc database (summed over instances, averaged over dispatches).
usage: tools/pmc_dump.py results.db [kernel-substring]"""
import collections
import sqlite3
import sys

db = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "dvsr"
c = sqlite3.connect(db)
rows = c.execute("select k.name, p.counter_name, p.dispatch_id, sum(p.counter_value), max(k.duration) from pmc_events p join kernels k "
                 "on k.dispatch_id = p.dispatch_id group by 1, 2, 3").fetchall()
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for n, cn, _, v, dur in rows:
    if sub not in n:
        continue
    a = agg[(n, cn)]
    a[0] += v; a[1] += dur; a[2] += 1
for (n, cn), (v, dur, cnt) in sorted(agg.items()):
    print("%-60s %-28s n=%4d mean=%16.1f avg_us=%9.1f" % (n[:60], cn, cnt, v / cnt, dur / cnt / 1e3))
