"""Winograd F(2x2, 3x3) kernel (conv2d_wino.hip) against the direct DMA-halo kernel: accuracy vs fp64 torch and time
per launch, through dvsr_conv2d_forward_packed.  The geometry choice is read once per process:
    DVSR_CONV_WINO=2 python tools/wino_bench.py     (Winograd wherever eligible)
    DVSR_CONV_WINO=0 python tools/wino_bench.py     (direct kernels)
"""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynavsr_amd import _lib as L  # noqa: E402

DEV = "cuda:0"
ACT = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.1), 2: F.relu}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def run(name, n, c0, c1, cout, h, w, act=1, res=False, ps=0, iters=20, check=True):
    cin = c0 + c1
    x0 = rnd(n, c0, h, w, seed=1)
    x1 = rnd(n, c1, h, w, seed=6) if c1 else None
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1 / np.sqrt(cin * 9))
    b = rnd(cout, seed=3, scale=0.1)
    r = rnd(n, cout, h, w, seed=4) if res else None
    d0, d1 = x0.to(DEV), (x1.to(DEV) if c1 else None)
    dw, db, dr = wt.to(DEV), b.to(DEV), (r.to(DEV) if res else None)
    y = torch.empty((n, cout // 4, 2 * h, 2 * w) if ps else (n, cout, h, w), device=DEV)
    d = L.Conv2dDesc(L.ptr(d0), L.ptr(d1), L.ptr(dw), L.ptr(db), L.ptr(dr), L.ptr(y), n, c0, c1, h, w, cout, 3, 1, 1,
                     act, ps, 1, 0, 0)
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "geometry")
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)) * 2, 1 << 20), dtype=torch.uint8, device=DEV)

    def call():
        L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "forward_packed")
    call()
    torch.cuda.synchronize()
    err = float("nan")
    if check:
        x = torch.cat([x0, x1], 1) if c1 else x0
        ref = ACT[act](F.conv2d(x.double(), wt.double(), b.double(), 1, 1))
        if res:
            ref = ref + r.double()
        if ps:
            ref = F.pixel_shuffle(ref, 2)
        got = y.cpu().double()
        err = float((got - ref).norm() / ref.norm())
        mx = float((got - ref).abs().max())
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = 2.0 * n * cout * cin * 9 * h * w
    print("%-28s geo=%s  %8.1f us  %7.1f TFLOP/s(alg)  rel-L2 %.2e  max-abs %.2e" %
          (name, list(geo), us, flops / us / 1e6, err, mx if check else float("nan")), flush=True)


if __name__ == "__main__":
    print("DVSR_CONV_WINO =", os.environ.get("DVSR_CONV_WINO"), "DVSR_CONV_ABLATE =", os.environ.get("DVSR_CONV_ABLATE"))
    if "--quick" in sys.argv:
        run("fe_rb_a 5x64->64 180x320", 5, 64, 0, 64, 180, 320, act=2, check=False)
        run("L1_offset_conv1 cat 128->64", 5, 64, 64, 64, 180, 320, act=1, check=False)
        run("rc_rb 1x64->64 180x320", 1, 64, 0, 64, 180, 320, act=2, check=False)
        sys.exit(0)
    run("fe_rb_a 5x64->64 180x320", 5, 64, 0, 64, 180, 320, act=2)
    run("fe_rb_b +res", 5, 64, 0, 64, 180, 320, act=0, res=True)
    run("rc_rb 1x64->64 180x320", 1, 64, 0, 64, 180, 320, act=2)
    run("L1_offset_conv1 cat 128->64", 5, 64, 64, 64, 180, 320, act=1)
    run("L1_om 64->216", 5, 64, 0, 216, 180, 320, act=0)
    run("upconv1 64->256 ps", 1, 64, 0, 256, 180, 320, act=1, ps=2)
    run("upconv2 64->256 ps 360x640", 1, 64, 0, 256, 360, 640, act=1, ps=2, check=False)
    run("HRconv 64->64 720x1280", 1, 64, 0, 64, 720, 1280, act=1, check=False)
    run("L2 5x64->64 90x160", 5, 64, 0, 64, 90, 160, act=1)
    run("odd 3x72->40 90x200", 3, 72, 0, 40, 90, 200, act=0)
    run("small 16x64->64 44x80", 16, 64, 0, 64, 44, 80, act=1)
    run("one chunk 2x8->64 128x128", 2, 8, 8, 64, 128, 128, act=1)
