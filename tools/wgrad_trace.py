#!/usr/bin/env python3
"""Cycle-stamp timeline of one conv2d_wgrad_split3v_kernel launch (debug build: python -m dynavsr_amd.build --trace).
    python tools/wgrad_trace.py [N CIN COUT H W]
Stamps (s_memtime, thread 0 of every workgroup): 0 start, 1 first tile loaded + converted, 2 tile loop done, 3 flush issued;
for the workgroup's THIRD tile: 4 top, 5 LDS stores issued, 6 barrier passed, 10 + i end of MFMA step i, 7 loop done, 8 second
barrier passed."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DVSR_HIP_LIB", os.path.join(HERE, "dynavsr_amd", "libdynavsr_hip_trace.so"))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dynavsr_amd import _lib as L  # noqa: E402

a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [40, 64, 64, 44, 80]
n, cin, cout, h, w = a
dev = "cuda:0"
x = torch.rand(n, cin, h, w, device=dev) - 0.5
gy = torch.rand(n, cout, h, w, device=dev) - 0.5
gw, gb = torch.empty(cout, cin, 3, 3, device=dev), torch.empty(cout, device=dev)
d = L.Conv2dDesc(L.ptr(x), None, None, None, None, None, n, cin, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
ws = torch.empty(max(int(L.lib().dvsr_conv2d_backward_workspace_bytes(d)), 16), dtype=torch.uint8, device=dev)


def call():
    L.check(L.lib().dvsr_conv2d_wgrad_split3(d, L.ptr(gy), L.ptr(gw), L.ptr(gb), ws.data_ptr(), ws.numel(), L.stream()),
            "dvsr_conv2d_wgrad_split3")


for _ in range(3):
    call()
torch.cuda.synchronize()
NB = 1 << 12
buf = torch.zeros(NB * 64, dtype=torch.int64, device=dev)
fn = L.lib().dvsr_debug_wgrad_trace
fn.argtypes = [ctypes.c_void_p]
fn(buf.data_ptr())
call()
torch.cuda.synchronize()
fn(None)
t = buf.cpu().numpy().reshape(NB, 64)
t = t[t[:, 0] != 0]
print("shape %s, workgroups traced: %d" % (a, len(t)))


def stat(name, v):
    v = np.asarray(v, dtype=np.float64)
    print("%-44s median %8.0f  p10 %8.0f  p90 %8.0f" % (name, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))


# (s_memtime counters are per XCD: cluster the workgroups by counter base before comparing start / end times)
order = np.argsort(t[:, 0]); ts = t[order]
cl = np.concatenate([[0], np.cumsum(np.diff(ts[:, 0]) > 10 ** 9)])
for c in range(int(cl.max()) + 1):
    w = ts[cl == c]
    span = float(w[:, 3].max() - w[:, 0].min())
    st0 = np.sort(w[:, 0] - w[:, 0].min())
    print("XCD cluster %d: %3d workgroups, span %9.0f cycles, median lifetime %9.0f, started within 5 %% of the span: %3d, later starts at %s"
          % (c, len(w), span, np.median(w[:, 3] - w[:, 0]), int((st0 < 0.05 * span).sum()),
             np.array2string(st0[st0 >= 0.05 * span][:6].astype(np.int64))))
stat("lifetime (0 -> 3)", t[:, 3] - t[:, 0])
stat("first tile: loads + convert (0 -> 1)", t[:, 1] - t[:, 0])
stat("tile loop (1 -> 2)", t[:, 2] - t[:, 1])
stat("flush issue (2 -> 3)", t[:, 3] - t[:, 2])
m = t[t[:, 8] != 0]
print("workgroups with a third tile: %d" % len(m))
stat("3rd tile: LDS stores (4 -> 5)", m[:, 5] - m[:, 4])
stat("3rd tile: barrier (5 -> 6)", m[:, 6] - m[:, 5])
prev = m[:, 6]
for i in range(12):
    stat("3rd tile: MFMA step %d" % i, m[:, 10 + i] - prev)
    prev = m[:, 10 + i]
stat("3rd tile: MFMA loop (6 -> 7)", m[:, 7] - m[:, 6])
stat("3rd tile: second barrier (7 -> 8)", m[:, 8] - m[:, 7])
stat("3rd tile: whole (4 -> 8)", m[:, 8] - m[:, 4])
