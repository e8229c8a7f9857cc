#!/usr/bin/env python3
"""fp32-MFMA vs bf16-MFMA conv path: forward time of EDVR-M (1x5x3x180x320) and EDVR-L (1x7x3x64x64, the LR size
of BASELINE configs[4]'s 256x256 HR tiles) and the output difference.  usage (GPU box): python tools/bf16_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.models.archs.EDVR_arch import EDVR  # noqa: E402


def run(cfg, shape, steps=10):
    outs = []
    for bf in (0, 1, 2):
        net = EDVR(bf16_mfma=bf, **cfg)
        net.load_state_dict(synth.edvr_state_dict(0, **cfg))
        net = net.cuda()
        x = synth.clip(1, *shape, smooth=False).cuda()
        with torch.no_grad():
            for _ in range(3):
                y = net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                y = net(x)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        outs.append((ms, y))
    (m0, y0) = outs[0]
    print("%-28s fp32 MFMA %8.2f ms" % (str(cfg or "EDVR-M") + " " + "x".join(map(str, shape)), m0))
    for name, (m1, y1) in zip(("bf16 operands", "3-way bf16 split"), outs[1:]):
        rel = float((y1 - y0).norm() / y0.norm())
        psnr = float(10 * torch.log10(1.0 / ((y1 - y0) ** 2).mean()))
        print("    %-18s %8.2f ms (x%.2f) | rel-L2 vs fp32 MFMA %.2e, PSNR %.1f dB" % (name, m1, m0 / m1, rel, psnr))


run({}, (1, 5, 180, 320))
run(dict(nf=128, nframes=7, back_RBs=40), (1, 7, 64, 64))
