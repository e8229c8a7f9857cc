#!/usr/bin/env python3
"""Per-kind forward breakdown of other EDVR configurations (x2 family, EDVR-L): looks for launches that fell off
the fast paths.  usage (GPU box): python tools/op_profile_cfg.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import engine, synth  # noqa: E402
from dynavsr_amd.models.archs.EDVR_arch import EDVR  # noqa: E402

for tag, cfg, (h, w) in (("EDVR-M x2 1x5x3x180x320", dict(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=2), (180, 320)),
                         ("EDVR-L x4 1x7x3x64x64", dict(nf=128, nframes=7, groups=8, front_RBs=5, back_RBs=40, scale=4), (64, 64)),
                         ("EDVR-L x4 1x7x3x180x320", dict(nf=128, nframes=7, groups=8, front_RBs=5, back_RBs=40, scale=4), (180, 320))):
    net = EDVR(**cfg)
    net.load_state_dict(synth.edvr_state_dict(0, **cfg))
    net = net.cuda()
    x = synth.clip(1, 1, cfg["nframes"], h, w, smooth=False).cuda()
    plan = engine.get_plan(net._cfg(), 1, h, w)
    params = [p.detach().contiguous() for p in net.ordered_parameters()]
    ws = torch.empty(plan.workspace_bytes(False), dtype=torch.uint8, device="cuda")
    out = torch.empty(1, 3, cfg["scale"] * h, cfg["scale"] * w, device="cuda")
    info = plan.op_info()
    tot = [0.0] * len(info)
    reps = 3
    for r in range(reps + 1):
        ms = plan.forward_timed(params, x, out, ws)
        if r:
            tot = [a + b for a, b in zip(tot, ms)]
    kinds = {}
    for (kind, name, fl, by), t in zip(info, tot):
        k = kinds.setdefault(kind, [0.0, 0.0, 0])
        k[0] += t / reps; k[1] += fl; k[2] += 1
    print("%s: %.2f ms" % (tag, sum(tot) / reps))
    for kind, (t, fl, n) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
        print("   %-10s %3d launches %8.3f ms %8.1f TFLOP/s" % (kind, n, t, fl / max(t, 1e-9) / 1e9))
    slow = sorted(((t / reps * 1e3, name, kind, fl / max(t / reps, 1e-9) / 1e9) for (kind, name, fl, by), t in zip(info, tot)), reverse=True)[:4]
    print("   slowest: " + "; ".join("%s %.0f us (%.0f TF)" % (n, us, tf) for us, n, k, tf in slow))
