#!/usr/bin/env python3
"""Per-stream busy time and critical-path view of a rocprofv3 kernel trace (rocpd database).

usage: tools/stream_timeline.py r_results.db [skip_fraction]
Takes the last (1 - skip_fraction) of the trace (steady state), splits kernels by HIP stream/queue
and prints, per stream, the busy time, the number of launches and the gap time between launches,
plus the wall span.  Shows whether the weight-gradient side stream or the main stream is the
critical path of the backward pass.
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    c = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    print("# kernels columns:", ", ".join(cols))
    rows = c.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
    if not rows:
        return
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[1] >= cut]
    span = (rows[-1][2] - rows[0][1]) / 1e3
    per = defaultdict(list)
    for name, s, e, q in rows:
        per[q].append((s, e, name))
    print("# window span %.1f us, %d launches" % (span, len(rows)))
    for q, lst in sorted(per.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in lst) / 1e3
        gaps = sum(max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)) / 1e3
        print("stream/queue %-6s launches %6d  busy %10.1f us (%5.1f%% of span)  idle between launches %10.1f us" %
              (q, len(lst), busy, 100.0 * busy / span, gaps))
        agg = defaultdict(lambda: [0, 0.0])
        for s, e, name in lst:
            agg[name][0] += 1
            agg[name][1] += (e - s) / 1e3
        for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print("      %-70s %6d %10.1f us  avg %8.1f" % (name[:70], n, tot, tot / n))


if __name__ == "__main__":
    main()
