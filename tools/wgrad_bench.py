#!/usr/bin/env python3
"""Weight-gradient kernel alone (dvsr_conv2d_backward with gx = NULL) at the inner-step and bench sizes.
usage (GPU box): python tools/wgrad_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


torch.manual_seed(0)
shapes = [(5, 64, 64, 44, 80, 3), (5, 128, 64, 44, 80, 3), (5, 64, 216, 44, 80, 3),
          (1, 64, 64, 176, 320, 3), (1, 320, 64, 44, 80, 1), (5, 64, 64, 22, 40, 3),
          (5, 64, 64, 180, 320, 3), (1, 64, 64, 720, 1280, 3)]
if os.environ.get("WGRAD_BENCH_BATCHED"):   # the layers of the batched inner step (8 frames as one batch)
    shapes = [(8, 64, 64, 44, 80, 3), (40, 64, 64, 44, 80, 3), (40, 64, 216, 44, 80, 3), (8, 64, 64, 176, 320, 3),
              (40, 64, 64, 176, 320, 3), (40, 64, 64, 22, 40, 3), (8, 320, 64, 44, 80, 1)]
from dynavsr_amd import _lib as L  # noqa: E402


def split3(x, gy, cout):
    """The exact 3-way bf16 split kernel (conv2d_wgrad_split3_kernel, what the plans run for 3x3 stride-1 layers)."""
    n, cin, h, w = x.shape
    gw, gb = torch.empty(cout, cin, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
    d = L.Conv2dDesc(L.ptr(x), None, None, None, None, None, n, cin, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_backward_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    return lambda: L.check(L.lib().dvsr_conv2d_wgrad_split3(d, L.ptr(gy), L.ptr(gw), L.ptr(gb), ws.data_ptr(), ws.numel(),
                                                           L.stream()), "dvsr_conv2d_wgrad_split3")


for (n, cin, cout, h, w, ks) in shapes:
    x = torch.randn(n, cin, h, w, device="cuda")
    gy = torch.randn(n, cout, h, w, device="cuda")
    wt = torch.randn(cout, cin, ks, ks, device="cuda")
    us = timeit(lambda: hipops.conv2d_backward(gy, x, wt, need_gx=False))
    fl = 2.0 * n * h * w * cin * cout * ks * ks
    extra = ""
    if ks == 3:
        us3 = timeit(split3(x, gy, cout))
        extra = "   | bf16 3-way split %9.1f us  %7.2f TFLOP/s" % (us3, fl / us3 / 1e6)
    print("wgrad %dx%d %3d->%3d k%d %4dx%-4d fp32 MFMA %9.1f us  %7.2f TFLOP/s%s" % (n, cin, cin, cout, ks, h, w, us, fl / us / 1e6, extra),
          flush=True)
