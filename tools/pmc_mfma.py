#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from two rocprofv3 PMC passes (separate runs, --kernel-trace only):
SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA occupies a SIMD's matrix pipe, summed over the chip) and
GRBM_GUI_ACTIVE (shader-clock cycles the GPU was busy during the dispatch).
GRBM_GUI_ACTIVE is reported once per XCD and rocprofv3 sums the 8 instances, hence the / 8:
    utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs);  effective clock = GUI_ACTIVE / 8 / duration
usage: tools/pmc_mfma.py mfma.db gui.db > profiles/rNN_pmc_mfma_util.txt"""
import collections
import sqlite3
import sys

mfma_db, gui_db = sys.argv[1:3]


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select k.name, p.dispatch_id, sum(p.counter_value), max(k.duration) from pmc_events p join kernels k "
                     "on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by 1, 2", (counter,)).fetchall()
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for n, _, v, dur in rows:
        a = agg[n]
        a[0] += v; a[1] += dur; a[2] += 1
    return agg


m = per_kernel(mfma_db, "SQ_VALU_MFMA_BUSY_CYCLES")
g = per_kernel(gui_db, "GRBM_GUI_ACTIVE")
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / --pmc GRBM_GUI_ACTIVE (separate passes) over")
print("# `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inner-step`; MI355X = 256 CUs x 4 SIMDs.")
print("# %-62s %6s %12s %12s %10s" % ("kernel", "n", "mfma_util", "clock_GHz", "avg_us"))
for n, (busy, _, cnt) in sorted(m.items(), key=lambda kv: -kv[1][0]):
    if n not in g or "dvsr" not in n or busy == 0:
        continue
    gui, dur, gc = g[n]
    cyc = (gui / gc) / 8.0  # shader cycles of one XCD during the dispatch
    util = (busy / cnt) / (cyc * 1024.0)
    print("  %-62s %6d %11.1f%% %12.2f %10.1f" % (n[:62], cnt, 100 * util, cyc / (dur / gc), dur / gc / 1e3))
