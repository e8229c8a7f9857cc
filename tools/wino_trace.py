#!/usr/bin/env python3
"""Cycle-stamp timeline of one conv2d_wino_kernel launch (debug build: python -m dynavsr_amd.build --trace).
    DVSR_CONV_WINO=2 python tools/wino_trace.py [N C0 C1 COUT H W]
Stamps (s_memtime, thread 0 of every workgroup): 0 start, 1 first chunk landed, 2 first V image built, 3 + k end of chunk
k's block, 40 loop done, 41 epilogue arithmetic + stores issued, 42 stores acknowledged.
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DVSR_HIP_LIB", os.path.join(HERE, "dynavsr_amd", "libdynavsr_hip_trace.so"))
os.environ.setdefault("DVSR_CONV_WINO", "2")
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
print("geometry", list(geo))
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
    print("%-36s median %8.0f  p10 %8.0f  p90 %8.0f" % (name, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))


rt = (t[:, 61] - t[:, 60]).astype(np.float64)
print("core clock (s_memtime / s_memrealtime): %.0f MHz; kernel wall span %.1f us" %
      (np.median((t[:, 42] - t[:, 0]) / np.maximum(rt, 1)) * 100.0, (t[:, 61].max() - t[:, 60].min()) / 100.0))
stat("workgroup lifetime", t[:, 42] - t[:, 0])
stat("prologue: first DMA landed", t[:, 1] - t[:, 0])
if t[:, 44].any():   # persistent workgroups: the stamps are those of the LAST tile of each workgroup
    stat("last tile: loop top before the barrier", t[:, 43] - t[:, 44])
    stat("last tile: loop-top barrier (DMA wait)", t[:, 1] - t[:, 43])
stat("prologue: first transform", t[:, 2] - t[:, 1])
nch = (c0 + c1) // 8
prev = t[:, 2]
for k in range(nch):
    cur = t[:, 3 + k] if k + 1 < nch else t[:, 40]
    stat("chunk %d block" % k, cur - prev)
    prev = cur
stat("epilogue arithmetic + store issue", t[:, 41] - t[:, 40])
if t[:, 50].any():   # the 2 x 2-block kernel: the epilogue's two exchange rounds
    for nm, i0, i1 in (("epilogue: C_i in place", 40, 50), ("epilogue: first barrier", 50, 51), ("round 0: LDS writes (+ residual loads)", 51, 52),
                       ("round 0: barrier", 52, 53), ("round 0: LDS reads + sums", 53, 54), ("barrier (reads done)", 54, 55),
                       ("round 1: LDS writes", 55, 56), ("round 0: activation + stores issued", 56, 57),
                       ("round 1: barrier + LDS reads + sums", 57, 58), ("round 1: activation + stores issued", 58, 41)):
        stat(nm, t[:, i1] - t[:, i0])
stat("store acknowledge", t[:, 42] - t[:, 41])
# workgroups per CU over time: rounds
hw = t[:, 63]
key = ((hw >> 8) & 0xFF).astype(np.int64)
print("start deciles (us, s_memrealtime):", np.percentile((t[:, 60] - t[:, 60].min()) / 100.0, [0, 10, 25, 50, 75, 90, 100]).round(1))
print("end   deciles (us, s_memrealtime):", np.percentile((t[:, 61] - t[:, 60].min()) / 100.0, [0, 10, 25, 50, 75, 90, 100]).round(1))

if t[:, 24].any() and not t[:, 27].any():   # conv2d_wino4_kernel: stamps inside chunk 3 of waves 0 and 4 (one SIMD)
    names = ["vmcnt wait (raw k+1, A of slot 0)", "s_barrier", "DMA issue + half 0 (slot 0 MFMAs | slot 1 built)", "half 1 (slot 1 MFMAs | slot 0 of k+1 built)"]
    for w, base in ((0, 20), (4, 30)):
        for i, nm in enumerate(names):
            stat("wave %d: %s" % (w, nm), t[:, base + i + 1] - t[:, base + i])
        stat("wave %d: chunk total" % w, t[:, base + 4] - t[:, base])
    stat("wave 4 start - wave 0 start", t[:, 30] - t[:, 20])
elif t[:, 20].any():   # conv2d_wino3_kernel: stamps inside phase 6 (chunk 3, rows 0/1) of waves 0 and 4 (one SIMD)
    names = ["wait for operands", "M0 (6 MFMAs + loads of pair 1)", "V (transform + split + writes)", "vmcnt / lgkmcnt wait",
             "s_barrier", "M1 (6 MFMAs + loads of next pair 0)", "tail (raw reads, DMA issue)"]
    for w, base in ((0, 20), (4, 30)):
        for i, nm in enumerate(names):
            stat("wave %d: %s" % (w, nm), t[:, base + i + 1] - t[:, base + i])
        stat("wave %d: phase total" % w, t[:, base + 7] - t[:, base])
    stat("wave 4 start - wave 0 start", t[:, 30] - t[:, 20])
