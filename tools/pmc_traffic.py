#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE rocprofv3 databases (separate --pmc passes) -> per-kernel KB-per-launch table
and the pmc_traffic.json that bench.py reads for roofline.traffic.
usage: pmc_traffic.py fetch.db write.db out.txt out.json"""
import collections
import hashlib
import json
import os
import sqlite3
import sys

fetch_db, write_db, out_txt, out_json = sys.argv[1:5]
res = {}
for db, cn in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
    c = sqlite3.connect(db)
    rows = c.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name=? "
                     "group by 1,2", (cn,)).fetchall()
    agg = collections.defaultdict(list)
    for n, _, v in rows:
        agg[n].append(v)
    for n, vs in agg.items():
        res.setdefault(n, {})[cn] = (sum(vs) / len(vs), len(vs))
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over",
         "# `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inner-step`; KB per launch, mean over launches.",
         "# Calibration on this workload (r01): WRITE_SIZE equals the algorithmic writes (tsa_gate 73.1 MB vs 74.0, conv_last",
         "# 10.8 vs 11.06).  FETCH_SIZE of these 4-byte-per-lane loads matches algorithmic reads x halo overlap UNcorrected",
         "# (conv_last 359 MB vs 236 MB x 340/256 + 11 MB).  Kernels whose inputs arrive as 16-B/lane LDS-DMA streams",
         "# (conv2d_dma_kernel, mdcn_fwd_dma_kernel: window, offsets and weights by global_load_lds_dwordx4) get the gfx950",
         "# correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts their 128-B requests at 64 B -> x2 (column fetch_x2).",
         "# %-64s %6s %14s %14s %14s" % ("kernel", "n", "fetch_KB", "fetch_x2_KB", "write_KB")]
tot = {"f": 0.0, "w": 0.0, "n": 0}
for n, d in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0] * kv[1].get("FETCH_SIZE", (0, 1))[1]):
    if "dvsr" not in n:
        continue
    f, w = d.get("FETCH_SIZE", (0, 0)), d.get("WRITE_SIZE", (0, 0))
    wide = "conv2d_dma_kernel" in n or "mdcn_fwd_dma_kernel" in n or "conv2d_wino_kernel" in n or "conv2d_wino4_kernel" in n or "conv2d_wino5_kernel" in n or "mdcn_fwd_split_kernel" in n
    fc = f[0] * (2.0 if wide else 1.0)
    lines.append("  %-64s %6d %14.1f %14s %14.1f" % (n[:64], f[1], f[0], ("%.1f" % fc) if wide else "-", w[0]))
    # every kernel bench.py files under "conv3x3s1": DMA-halo, register-staged (conv_first) and K-split 3x3 stride-1 convs
    if "conv2d_dma_kernel" in n or "conv2d_wino_kernel" in n or "conv2d_wino4_kernel" in n or "conv2d_wino5_kernel" in n or "conv2d_pipe_kernel<3, 1" in n or "conv2d_ksplit_kernel" in n:
        tot["f"] += fc * f[1]; tot["w"] += w[0] * f[1]; tot["n"] += f[1]
per = (tot["f"] + tot["w"]) / max(tot["n"], 1) * 1024
lines.append("# dominant kernel (3x3/s1 conv, all kernels and geometries, corrected fetch): %.1f MB HBM traffic per launch (fetch %.1f + write %.1f) "
             "vs 102.5 MB algorithmic" % (per / 1e6, tot["f"] / max(tot["n"], 1) * 1024 / 1e6,
                                          tot["w"] / max(tot["n"], 1) * 1024 / 1e6))
open(out_txt, "w").write("\n".join(lines) + "\n")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dig = hashlib.sha256()
for f in ("conv2d_v2.hip", "conv2d_wino.hip", "conv2d_wino3.hip", "conv2d_wino4.hip", "conv2d_wino5.hip", "small_grid.h", "common.h"):   # bench.py refuses the figure when these have changed since
    dig.update(open(os.path.join(ROOT, "dynavsr_amd", "csrc", f), "rb").read())
json.dump({"sources_sha256_16": dig.hexdigest()[:16], "conv3x3s1": {"bytes_per_launch": per, "fetch_bytes": tot["f"] / max(tot["n"], 1) * 1024,
                         "write_bytes": tot["w"] / max(tot["n"], 1) * 1024,
                         "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
                                   % out_txt.split("/")[-1]}}, open(out_json, "w"), indent=1)
print(lines[-1])
