#!/usr/bin/env python3
"""Degradation.apply: dvsr_degrade_apply on the GPU vs the reference's CPU tensor path
(ReflectionPad2d + conv2d(groups=3, stride=scale), restated with torch on the host cores).
usage (GPU box): python tools/degrade_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd.data.random_kernel_generator import Degradation  # noqa: E402

for shape, what in (((5, 3, 256, 256), "training sample, HR patch 5x3x256x256 -> LR 64x64 -> SLR 16x16"),
                    ((5, 3, 720, 1280), "full frames 5x3x720x1280 -> 180x320 -> 45x80")):
    img = torch.rand(*shape)
    d = Degradation(21, 4, theta=0.4, sigma=[1.7, 2.9])
    ks = torch.from_numpy(d.kernel_shift(d.kernel)).float()
    w = ks.repeat(3, 1, 1, 1)
    pad = torch.nn.ReflectionPad2d(ks.shape[0] // 2)

    def cpu_chain(x):
        lr = torch.nn.functional.conv2d(pad(x), w, groups=3, stride=4).mul(255).clamp(0, 255).round().div(255)
        return lr, torch.nn.functional.conv2d(pad(lr), w, groups=3, stride=4)
    cpu_chain(img)
    t0 = time.perf_counter()
    for _ in range(3):
        lr_c, slr_c = cpu_chain(img)
    cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
    x = img.cuda()
    for _ in range(3):
        lr = d.apply(x, quantise=True); slr = d.apply(lr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        lr = d.apply(x, quantise=True); slr = d.apply(lr)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    dev = e0.elapsed_time(e1) / n * 1e3
    algo = 4 * (x.numel() + 2 * lr.numel() + slr.numel())
    print("%s" % what)
    print("  GPU %.1f us per HR->LR->SLR chain on the stream (%.0f GB/s of %.1f MB algorithmic traffic), %.2f ms per call pair "
          "incl. the host-side kernel shift; CPU (torch, %d threads) %.1f ms; max |diff| LR %.1e SLR %.1e"
          % (dev, algo / dev / 1e3, algo / 1e6, wall, torch.get_num_threads(), cpu_ms,
             float((lr.cpu() - lr_c).abs().max()), float((slr.cpu() - slr_c).abs().max())))
