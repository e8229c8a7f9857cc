#!/usr/bin/env python3
"""Cycle-stamp phases of one mdcn_bwd_fused_kernel launch (debug build: python -m dynavsr_amd.build --trace).
usage (GPU box): python tools/dcn_bwd_trace.py [N H W]
Stamps (thread 0 of every workgroup): 0 start, 1 staging loads issued and written, 2 barrier passed, 3 dcol MFMAs done,
4 scale + window clear done, 5 sampling done, 6 barrier passed, 7 window flushed, 8 weight-gradient MFMAs + exchange done
(last 64-cout block), 9 partials stored."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DVSR_HIP_LIB", os.path.join(HERE, "dynavsr_amd", "libdynavsr_hip_trace.so"))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dynavsr_amd import _lib, hipops  # noqa: E402

n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 44, 80)
torch.manual_seed(0)
x = torch.randn(n, 64, h, w, device="cuda")
off = torch.randn(n, 144, h, w, device="cuda") * 1.3
msk = torch.sigmoid(torch.randn(n, 72, h, w, device="cuda"))
wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.04
gy = torch.randn(n, 64, h, w, device="cuda")
for _ in range(3):
    hipops.mdcn_backward(x, off, msk, wt, gy, 1, 1, 1, 1, 8)
torch.cuda.synchronize()
NB = 1 << 16
buf = torch.zeros(NB * 16, dtype=torch.int64, device="cuda")
fn = _lib.lib().dvsr_debug_dcn_bwd_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn(buf.data_ptr(), 0)
hipops.mdcn_backward(x, off, msk, wt, gy, 1, 1, 1, 1, 8)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NB, 16)
t = t[t[:, 0] != 0]
med = lambda v: float(np.median(v))
print("mdcn_bwd_fused_kernel %dx64x%dx%d: %d workgroups, launch span %.0f cycles" % (n, h, w, len(t), float(t[:, 7].max() - t[:, 0].min())))
names = ["staging (window, W^T, offsets) issued + written", "first barrier", "dcol = W^T gout (192 MFMAs per wave)",
         "scale, barriers, window clear", "sampling (18 x (pixel row, tap))", "barrier", "window flush (global atomics)",
         "weight gradient: staging + 192 MFMAs per wave + exchange", "partials stored"]
for i, nm in enumerate(names):
    print("  %-52s %8.0f cycles (median)" % (nm, med(t[:, i + 1] - t[:, i])))
if t[:, 10].any():
    for nm, i0, i1 in (("  prologue: loads issued", 0, 10), ("  prologue: window landed + written", 10, 11), ("  prologue: W^T written", 11, 12),
                       ("  prologue: offsets landed (spills)", 12, 1)):
        print("  %-52s %8.0f cycles (median)" % (nm, med(t[:, i1] - t[:, i0])))
print("  %-52s %8.0f" % ("lifetime", med(t[:, 9] - t[:, 0])))
