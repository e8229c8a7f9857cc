#!/usr/bin/env python3
"""Total HBM-side traffic of a profiled loop: FETCH_SIZE / WRITE_SIZE databases of two rocprofv3 --pmc passes (separate,
--kernel-trace only) over the SAME command -> bytes per step, summed over all dvsr kernels (FETCH_SIZE doubled for the
kernels that read 16 B per lane through LDS-DMA, as MI355X_MICROARCH.md prescribes for gfx950).
usage: pmc_total.py fetch.db write.db <steps run by the command> [marker-substring: kernel that runs once per step]"""
import sqlite3
import sys

fetch_db, write_db, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
tot = {}
for db, cn in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
    c = sqlite3.connect(db)
    rows = c.execute("select name, sum(counter_value), count(*) from pmc_events where counter_name=? group by 1", (cn,)).fetchall()
    s = 0.0
    for n, v, k in rows:
        if "dvsr" not in n:
            continue
        wide = cn == "FETCH_SIZE" and ("conv2d_dma_kernel" in n or "mdcn_fwd_dma_kernel" in n or "conv2d_dmarow" in n or
                                        "conv2d_wino_kernel" in n or "conv2d_wino3_kernel" in n)
        s += v * (2.0 if wide else 1.0)
    tot[cn] = s * 1024.0 / steps
print("fetch %.1f MB + write %.1f MB = %.1f MB per step" % (tot["FETCH_SIZE"] / 1e6, tot["WRITE_SIZE"] / 1e6,
                                                          (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1e6))
print("BYTES_PER_STEP %.0f" % (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]))
if len(sys.argv) > 6 and sys.argv[4] == "--json":   # merge into profiles' pmc_traffic.json: --json <file> <key>
    import json
    path, key = sys.argv[5], sys.argv[6]
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        d = {}
    d[key] = {"bytes_per_step": tot["FETCH_SIZE"] + tot["WRITE_SIZE"], "fetch_bytes": tot["FETCH_SIZE"], "write_bytes": tot["WRITE_SIZE"],
              "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), all kernels, per step"}
    # extra "name=value" arguments (numbers where they parse) are stored beside the figures: e.g. the batched inner step's
    # frames_per_batch=16 h=176 w=320 -> bench.py divides by the frames of a batch
    for kv in sys.argv[7:]:
        k_, v_ = kv.split("=", 1)
        try:
            v_ = float(v_) if "." in v_ else int(v_)
        except ValueError:
            pass
        d[key][k_] = v_
    if "frames_per_batch" in d[key]:
        d[key]["bytes_per_frame_step"] = d[key]["bytes_per_step"] / d[key]["frames_per_batch"]
    json.dump(d, open(path, "w"), indent=1)
