#!/usr/bin/env python3
"""Cycle-stamp timeline of one conv2d_wino5_kernel launch (debug build: python -m dynavsr_amd.build --trace).
    python tools/wino5_trace.py [N C0 C1 COUT H W]
Stamps (thread 0 of every workgroup, a consumer wave): 0 start, 3 + k end of chunk k (k < 16; chunk 0 carries the prologue:
raw(0) landed, V(0) built by the producer waves), 40 loop done, 50 column transform done, 51.. the exchange rounds,
41 stores issued, 42 stores acknowledged.
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DVSR_HIP_LIB", os.path.join(HERE, "dynavsr_amd", "libdynavsr_hip_trace.so"))
os.environ.setdefault("DVSR_CONV_WINO", "2")
os.environ.setdefault("DVSR_CONV_WINO5", "2")
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dynavsr_amd import _lib as L  # noqa: E402

a = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 else [5, 64, 0, 64, 180, 320]
n, c0, c1, cout, h, w = a
dev = "cuda:0"
x0 = torch.rand(n, c0, h, w, device=dev)
x1 = torch.rand(n, c1, h, w, device=dev) if c1 else None
wt = torch.rand(cout, c0 + c1, 3, 3, device=dev) - 0.5
b = torch.rand(cout, device=dev)
y = torch.empty(n, cout, h, w, device=dev)
d = L.Conv2dDesc(L.ptr(x0), L.ptr(x1), L.ptr(wt), L.ptr(b), None, L.ptr(y), n, c0, c1, h, w, cout, 3, 1, 1, 1, 0, 1, 0, 0)
geo = (ctypes.c_int * 4)()
L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "geometry")
print("shape", a, "geometry", list(geo))
ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)) * 2, 1 << 20), dtype=torch.uint8, device=dev)


def call():
    L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "forward_packed")


for _ in range(3):
    call()
torch.cuda.synchronize()
NB = 1 << 14
buf = torch.zeros(NB * 64, dtype=torch.int64, device=dev)
fn = L.lib().dvsr_debug_conv_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn(buf.data_ptr(), 0)
call()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NB, 64)
t = t[t[:, 0] != 0]
print("workgroups traced: %d" % len(t))


def stat(name, v):
    v = np.asarray(v, dtype=np.float64)
    print("%-44s median %8.0f  p10 %8.0f  p90 %8.0f" % (name, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))


rt = (t[:, 61] - t[:, 60]).astype(np.float64)
print("core clock (cycle counter / s_memrealtime): %.0f MHz; kernel wall span %.1f us" %
      (np.median((t[:, 42] - t[:, 0]) / np.maximum(rt, 1)) * 100.0, (t[:, 61].max() - t[:, 60].min()) / 100.0))
stat("workgroup lifetime", t[:, 42] - t[:, 0])
nch = (c0 + c1) // 8
prev = t[:, 0]
for k in range(min(nch, 16)):
    cur = t[:, 3 + k]
    stat("chunk %d%s" % (k, " (+ prologue)" if k == 0 else ""), cur - prev)
    prev = cur
stat("chunks %d.. + loop exit" % min(nch, 16) if nch > 16 else "loop exit", t[:, 40] - prev)
for nm, i0, i1 in (("epilogue: column transform", 40, 50), ("first barrier", 50, 51), ("round 0: LDS writes", 51, 52), ("round 0: barrier", 52, 53),
                   ("round 0: reads + row transform + stores", 53, 54), ("barrier (reads done)", 54, 55), ("round 1: LDS writes", 55, 56),
                   ("round 1: barrier", 56, 57), ("round 1: reads + row transform + stores", 57, 58), ("stores acknowledged", 41, 42)):
    stat(nm, t[:, i1] - t[:, i0])

# workgroups that followed each other on one CU (XCC_ID, HW_ID's se / sh / cu): idle time between the end of one and the start
# of the next (s_memrealtime, 100 MHz)
import collections  # noqa: E402
by_cu = collections.defaultdict(list)
for r in t:
    hw = int(r[63])
    by_cu[(int(r[62]) & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)].append((int(r[60]), int(r[61])))
gaps = []
for v in by_cu.values():
    v.sort()
    gaps += [b[0] - a_[1] for a_, b in zip(v, v[1:])]
if gaps:
    print("CUs seen: %d; workgroups per CU: %s" % (len(by_cu), dict(collections.Counter(len(v) for v in by_cu.values()))))
    stat("idle between two workgroups of a CU (x 10 ns)", gaps)
    first = min(r[60] for r in t)
    stat("a CU's first workgroup starts after the kernel's first (x 10 ns)", [v[0][0] - first for v in by_cu.values()])
