"""Edge semantics of the exact 3-way bf16 split (conv2d_wino4 / conv2d_wino3, conv2d_wgrad_split3) against the fp32 kernels:
per-channel scales over ten decades, inputs in the range where the mid / lo pieces are bf16 subnormals, one inf, one NaN.
Prints what each path returns; tests/test_gpu_ops.py::test_split_kernels_* pin it.
"""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynavsr_amd import _lib as L  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape) * scale)


def conv(x, wt, b, wino):
    os.environ["DVSR_CONV_WINO"] = "2" if wino else "0"
    os.environ["DVSR_CONV_WINO3"] = "1"
    n, c, h, w = x.shape
    cout = wt.shape[0]
    dx, dw, db = x.float().cuda(), wt.float().cuda(), b.float().cuda()
    y = torch.empty(n, cout, h, w, device="cuda")
    d = L.Conv2dDesc(L.ptr(dx), None, L.ptr(dw), L.ptr(db), None, L.ptr(y), n, c, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "geometry")
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "forward_packed")
    torch.cuda.synchronize()
    return y.cpu(), list(geo)


def wgrad(x, gy, split):
    n, c, h, w = x.shape
    cout = gy.shape[1]
    dx, dg = x.float().cuda(), gy.float().cuda()
    gw, gb = torch.empty(cout, c, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
    d = L.Conv2dDesc(L.ptr(dx), None, None, None, None, None, n, c, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_backward_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    if split:
        L.check(L.lib().dvsr_conv2d_wgrad_split3(d, L.ptr(dg), L.ptr(gw), L.ptr(gb), ws.data_ptr(), ws.numel(), L.stream()), "wgrad_split3")
    else:
        gx = torch.empty_like(dx)
        wz = torch.zeros(cout, c, 3, 3, device="cuda")
        d2 = L.Conv2dDesc(L.ptr(dx), None, L.ptr(wz), None, None, None, n, c, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
        os.environ["DVSR_WGRAD_SPLIT3"] = "0"
        L.check(L.lib().dvsr_conv2d_backward(d2, L.ptr(dg), L.ptr(gx), None, L.ptr(gw), L.ptr(gb), ws.data_ptr(), ws.numel(), L.stream()), "backward")
    torch.cuda.synchronize()
    return gw.cpu(), gb.cpu()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def main():
    n, c, cout, h, w = 4, 64, 64, 96, 128   # (large enough for the Winograd kernel: the small-grid K-split kernel takes fewer than 700 workgroups)
    wt = rnd(cout, c, 3, 3, seed=2, scale=1 / np.sqrt(c * 9))
    b = rnd(cout, seed=3, scale=0.1)
    x = rnd(n, c, h, w, seed=1)
    # (a) per-channel scales 1e-6 .. 1e+4 on the input, the inverse on the weights' input channels (outputs stay O(1))
    s = torch.logspace(-6, 4, c, dtype=torch.float64)
    xa, wa = x * s.view(1, c, 1, 1), wt / s.view(1, c, 1, 1)
    ref = F.conv2d(xa.double(), wa.double(), b.double(), 1, 1)
    for wino in (1, 0):
        y, geo = conv(xa, wa, b, wino)
        print("scales 1e-6..1e4, wino=%d geo=%s: rel-L2 %.3e max-abs %.3e" % (wino, geo, rel(y, ref), float((y.double() - ref).abs().max())))
    # (a2) scales on the OUTPUT channels of the weights (outputs span the decades; error measured per channel)
    wb = wt * s.view(c, 1, 1, 1)
    ref = F.conv2d(x.double(), wb.double(), None, 1, 1)
    for wino in (1, 0):
        y, geo = conv(x, wb, torch.zeros(cout), wino)
        pc = ((y.double() - ref).flatten(2).norm(dim=2).norm(dim=0) / ref.flatten(2).norm(dim=2).norm(dim=0))
        print("cout scales, wino=%d: worst per-channel rel-L2 %.3e" % (wino, float(pc.max())))
    # (b) tiny inputs: |x| ~ 2^-115 (mid / lo pieces are bf16 subnormals), weights O(0.04)
    for e in (-100, -108, -112, -116, -120, -124):
        xt = x * 2.0 ** e
        ref = F.conv2d(xt.double(), wt.double(), None, 1, 1)
        out = []
        for wino in (1, 0):
            y, geo = conv(xt, wt, torch.zeros(cout), wino)
            out.append("wino=%d rel-L2 %.3e max-abs/2^e %.3e" % (wino, rel(y, ref), float((y.double() - ref).abs().max()) / 2.0 ** e))
        print("tiny 2^%d: %s" % (e, " | ".join(out)))
    # (b2) tiny weights
    for e in (-100, -112, -120):
        wtt = wt * 2.0 ** e
        ref = F.conv2d(x.double(), wtt.double(), None, 1, 1)
        out = []
        for wino in (1, 0):
            y, geo = conv(x, wtt, torch.zeros(cout), wino)
            out.append("wino=%d rel-L2 %.3e" % (wino, rel(y, ref)))
        print("tiny weights 2^%d: %s" % (e, " | ".join(out)))
    # (b3) huge inputs: hi must not round to inf
    for e in (100, 120, 126):
        xt = x * 2.0 ** e * 0.25
        wtt = wt * 2.0 ** (-e)
        ref = F.conv2d(xt.double(), wtt.double(), None, 1, 1)
        out = []
        for wino in (1, 0):
            y, geo = conv(xt, wtt, torch.zeros(cout), wino)
            out.append("wino=%d rel-L2 %.3e finite %s" % (wino, rel(torch.nan_to_num(y), ref), bool(torch.isfinite(y).all())))
        print("huge 2^%d: %s" % (e, " | ".join(out)))
    # (c) one inf / one NaN input element
    for name, val in (("inf", float("inf")), ("nan", float("nan"))):
        xi = x.clone()
        xi[1, 7, 21, 34] = val
        ref = F.conv2d(xi.double(), wt.double(), b.double(), 1, 1)
        for wino in (1, 0):
            y, geo = conv(xi, wt, b, wino)
            bad_ref, bad = ~torch.isfinite(ref), ~torch.isfinite(y)
            ys, xs = torch.where(bad.any(dim=1)[1])
            print("%s wino=%d: non-finite ref %d got %d; ref subset of got %s; got rows %d..%d cols %d..%d; image 0 finite %s; nan %d inf %d; finite part rel-L2 %.3e" % (
                name, wino, int(bad_ref.sum()), int(bad.sum()), bool((bad | ~bad_ref).all()), int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()),
                bool(torch.isfinite(y[0]).all()), int(torch.isnan(y).sum()), int(torch.isinf(y).sum()),
                rel(torch.where(bad, torch.zeros_like(y), y), torch.where(bad, torch.zeros_like(ref), ref))))
    # ---- weight gradient
    gy = rnd(n, cout, h, w, seed=5)
    wz = torch.zeros(cout, c, 3, 3, dtype=torch.float64, requires_grad=True)
    def gref(xx, gg):
        (g,) = torch.autograd.grad(F.conv2d(xx.double(), wz, padding=1), wz, gg.double())
        return g
    xa = x * s.view(1, c, 1, 1)
    for split in (1, 0):
        gw, gb = wgrad(xa, gy, split)
        r = gref(xa, gy)
        pc = (gw.double() - r).flatten(2).norm(dim=2).norm(dim=0) / r.flatten(2).norm(dim=2).norm(dim=0)
        print("wgrad cin scales split=%d: worst per-cin rel-L2 %.3e" % (split, float(pc.max())))
    for e in (-100, -112, -120):
        for split in (1, 0):
            gw, gb = wgrad(x * 2.0 ** e, gy, split)
            print("wgrad tiny x 2^%d split=%d: rel-L2 %.3e" % (e, split, rel(gw, gref(x * 2.0 ** e, gy))))
    for name, val in (("inf", float("inf")), ("nan", float("nan"))):
        xi = x.clone()
        xi[1, 7, 21, 34] = val
        r = gref(xi, gy)
        for split in (1, 0):
            gw, gb = wgrad(xi, gy, split)
            bad_ref, bad = ~torch.isfinite(r), ~torch.isfinite(gw)
            print("wgrad %s split=%d: non-finite ref %d got %d; subset %s; cin touched %s; nan %d inf %d" % (
                name, split, int(bad_ref.sum()), int(bad.sum()), bool((bad | ~bad_ref).all()),
                sorted(set(torch.where(bad)[1].tolist())), int(torch.isnan(gw).sum()), int(torch.isinf(gw).sum())))


if __name__ == "__main__":
    main()
