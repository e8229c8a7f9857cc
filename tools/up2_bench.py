#!/usr/bin/env python3
"""2x bilinear up-sampling at the PCD pyramid's sizes: microseconds and GB/s per launch (hipEvents).
usage (GPU box): python tools/up2_bench.py   (DVSR_UP2=4: the four-columns-per-thread kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynavsr_amd import hipops  # noqa: E402

for (n, c, h, w) in [(5, 64, 90, 160), (5, 64, 45, 80), (1, 64, 90, 160), (1, 3, 180, 320)]:
    x = torch.randn(n, c, h, w, device="cuda")
    for _ in range(3):
        y = hipops.upsample_bilinear(x, 2, 2.0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    s.record()
    for _ in range(reps):
        y = hipops.upsample_bilinear(x, 2, 2.0)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    print("upsample 2x %dx%dx%dx%d: %7.1f us  %7.1f GB/s (incl. the output allocation)" % (n, c, h, w, us, x.numel() * 5 * 4 / us / 1e3))
