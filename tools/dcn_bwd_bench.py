#!/usr/bin/env python3
"""DCN backward alone (dvsr_mdcn_backward) at the inner-step and bench sizes; kernel times come from
rocprofv3 --kernel-trace --stats around this script.  usage (GPU box): python tools/dcn_bwd_bench.py [reps [big|small]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
which = sys.argv[2] if len(sys.argv) > 2 else "both"
for (n, h, w) in [s_ for s_ in [(5, 44, 80), (5, 180, 320)] if which == "both" or (which == "big") == (s_[1] == 180)]:
    x = torch.randn(n, 64, h, w, device="cuda")
    off = torch.randn(n, 144, h, w, device="cuda") * 1.3
    msk = torch.sigmoid(torch.randn(n, 72, h, w, device="cuda"))
    wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.04
    gy = torch.randn(n, 64, h, w, device="cuda")
    for _ in range(2):
        hipops.mdcn_backward(x, off, msk, wt, gy, 1, 1, 1, 1, 8)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        hipops.mdcn_backward(x, off, msk, wt, gy, 1, 1, 1, 1, 8)
    e.record()
    torch.cuda.synchronize()
    print("mdcn backward %dx64x%dx%d  %9.1f us per call (op level: zero-fill + kernels + wgrad)" % (n, h, w, s.elapsed_time(e) / reps * 1e3))
