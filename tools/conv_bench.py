#!/usr/bin/env python3
"""Micro-benchmark of single launches of the hot kernels at the 180x320 workload sizes.
usage (GPU box): python tools/conv_bench.py [reps]   -> one line per kernel: us, TFLOP/s or GB/s"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"


def timeit(fn, reps=reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3  # us


def report(name, us, flops=0.0, nbytes=0.0):
    print("%-34s %9.1f us  %7.2f TFLOP/s  %8.1f GB/s" % (name, us, flops / us / 1e6, nbytes / us / 1e3), flush=True)


torch.manual_seed(0)
N, C, H, W = 5, 64, 180, 320
x = torch.randn(N, C, H, W, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.04
b = torch.randn(C, device=dev)
fl = 2.0 * N * H * W * C * C * 9
report("conv3x3 64->64 5x180x320", timeit(lambda: hipops.conv2d_forward(x, w, b, act=1)), fl, 4.0 * 2 * x.numel())
x2 = torch.randn(N, C, H, W, device=dev)
w2 = torch.randn(C, 2 * C, 3, 3, device=dev) * 0.03
report("conv3x3 128->64 (2 ptr) 5x180x320", timeit(lambda: hipops.conv2d_forward(x, w2, b, act=1, x1=x2)), 2 * fl,
       4.0 * 3 * x.numel())
w216 = torch.randn(216, C, 3, 3, device=dev) * 0.04
b216 = torch.randn(216, device=dev)
report("conv3x3 64->216 5x180x320", timeit(lambda: hipops.conv2d_forward(x, w216, b216)), fl * 216 / 64,
       4.0 * x.numel() * (1 + 216 / 64))
xh = torch.randn(1, 64, 720, 1280, device=dev)
report("conv3x3 64->64 1x720x1280 (HRconv)", timeit(lambda: hipops.conv2d_forward(xh, w, b, act=1)),
       2.0 * 720 * 1280 * 64 * 64 * 9, 4.0 * 2 * xh.numel())
w3 = torch.randn(3, 64, 3, 3, device=dev) * 0.04
b3 = torch.randn(3, device=dev)
report("conv3x3 64->3 1x720x1280 (last)", timeit(lambda: hipops.conv2d_forward(xh, w3, b3)),
       2.0 * 720 * 1280 * 64 * 3 * 9, 4.0 * xh.numel())
w1 = torch.randn(64, 320, 1, 1, device=dev) * 0.05
g = torch.randn(1, 320, H, W, device=dev)
report("conv1x1 320->64 1x180x320", timeit(lambda: hipops.conv2d_forward(g, w1, b, act=1)),
       2.0 * H * W * 320 * 64, 4.0 * (g.numel() + 64 * H * W))
om = torch.randn(N, 216, H, W, device=dev)
report("mdcn pack fwd 64ch 5x180x320", timeit(lambda: hipops.mdcn_pack_forward(x, om, w, b, 8, act=1)),
       fl, 4.0 * (2 * x.numel() + om.numel()))
gy = torch.randn(N, C, H, W, device=dev)
report("conv3x3 bwd (dgrad+wgrad) 64->64", timeit(lambda: hipops.conv2d_backward(gy, x, w), max(3, reps // 4)),
       2 * fl, 4.0 * 4 * x.numel())
xs, oms = x[:, :, :44, :80].contiguous(), om[:, :, :44, :80].contiguous()
off, msk = oms[:, :144].contiguous(), torch.sigmoid(oms[:, 144:]).contiguous()
gys = gy[:, :, :44, :80].contiguous()
report("mdcn bwd 64ch 5x44x80", timeit(lambda: hipops.mdcn_backward(xs, off, msk, w, gys, 1, 1, 1, 1, 8),
                                        max(3, reps // 4)), 3 * fl * 44 * 80 / (H * W), 0)
