"""Determinism soak of the Winograd kernel: the same launch repeated must give bit-identical outputs (a race between the
chunk pipeline's DMA / transform / MFMA stages or in the output exchange would show up as run-to-run differences).
    python tools/wino_soak.py [repeats]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DVSR_CONV_WINO", "2")
from dynavsr_amd import _lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
shapes = [(5, 64, 0, 64, 180, 320, 1, True, 0), (5, 64, 64, 64, 180, 320, 1, False, 0), (2, 64, 0, 216, 90, 160, 0, False, 0),
          (1, 64, 0, 256, 180, 320, 1, False, 2), (16, 64, 0, 64, 44, 80, 2, True, 0), (3, 72, 0, 40, 90, 200, 0, False, 0)]
bad = 0
for (n, c0, c1, cout, h, w, act, res, ps) in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.randn(n, c0, h, w, device=dev, generator=g)
    x1 = torch.randn(n, c1, h, w, device=dev, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, 3, 3, device=dev, generator=g) * 0.05
    b = torch.randn(cout, device=dev, generator=g)
    r = torch.randn(n, cout, h, w, device=dev, generator=g) if res else None
    shp = (n, cout // 4, 2 * h, 2 * w) if ps else (n, cout, h, w)
    y = torch.empty(shp, device=dev)
    d = L.Conv2dDesc(L.ptr(x0), L.ptr(x1), L.ptr(wt), L.ptr(b), L.ptr(r), L.ptr(y), n, c0, c1, h, w, cout, 3, 1, 1, act, ps, 1, 0, 0)
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "geometry")
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)), 16), dtype=torch.uint8, device=dev)
    ref = None
    diff = 0
    for i in range(reps):
        y.fill_(float("nan"))
        L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "forward_packed")
        if ref is None:
            ref = y.clone()
            assert torch.isfinite(ref).all()
        elif not torch.equal(y, ref):
            diff += 1
    bad += diff
    print("shape", (n, c0, c1, cout, h, w, act, res, ps), "geo", list(geo), "repeats", reps, "differing runs", diff, flush=True)
print("SOAK", "FAILED" if bad else "ok")
