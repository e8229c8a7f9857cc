"""Debug aid for conv2d_wino5_kernel: structured inputs, error maps (DVSR_CONV_WINO=2 DVSR_CONV_WINO5=2)."""
import ctypes, os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynavsr_amd import _lib as L  # noqa: E402
DEV = "cuda:0"

def conv(x, wt, b=None, act=0):
    n, c, h, w = x.shape
    cout = wt.shape[0]
    dx, dw = x.float().to(DEV), wt.float().to(DEV)
    db = b.float().to(DEV) if b is not None else None
    y = torch.empty(n, cout, h, w, device=DEV)
    d = L.Conv2dDesc(L.ptr(dx), None, L.ptr(dw), L.ptr(db), None, L.ptr(y), n, c, 0, h, w, cout, 3, 1, 1, act, 0, 1, 0, 0)
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "geometry")
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)) * 2, 1 << 20), dtype=torch.uint8, device=DEV)
    L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "fwd")
    torch.cuda.synchronize()
    return y.cpu().double(), list(geo)

def report(name, x, wt):
    y, geo = conv(x, wt)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    e = (y - ref)
    print(name, "geo", geo, "rel", float(e.norm() / ref.norm()), "finite", bool(torch.isfinite(y).all()))
    return y, ref

def main():
    torch.manual_seed(0)
    n, c, cout, h, w = 1, 16, 64, 8, 64
    # 1. identity centre tap
    wt = torch.zeros(cout, c, 3, 3)
    for o in range(cout):
        wt[o, o % c, 1, 1] = 1.0
    x = torch.randn(n, c, h, w)
    y, ref = report("identity", x, wt)
    e = (y - ref).abs()
    print(" err by cout:", [round(float(v), 3) for v in e.amax(dim=(0, 2, 3))][:64])
    print(" err by row :", [round(float(v), 3) for v in e.amax(dim=(0, 1, 3))])
    print(" err by col :", [round(float(v), 3) for v in e.amax(dim=(0, 1, 2))])
    print(" y[0,0,:4,:8]\n", y[0, 0, :4, :8].numpy().round(3), "\n ref\n", ref[0, 0, :4, :8].numpy().round(3))
    # 2. constant input, single tap weights
    x1 = torch.ones(n, c, h, w)
    for tap in ((1, 1), (0, 0), (2, 2), (0, 2)):
        wt = torch.zeros(cout, c, 3, 3); wt[:, 0, tap[0], tap[1]] = 1.0
        y, ref = report("ones tap%s" % (tap,), x1, wt)
        print(" y[0,0]\n", y[0, 0, :8, :12].numpy().round(3), "\n ref\n", ref[0, 0, :8, :12].numpy().round(3))
    # 3. random everything, bigger
    x = torch.randn(2, 64, 16, 128); wt = torch.randn(64, 64, 3, 3) / 24
    y, ref = report("random", x, wt)


if __name__ == "__main__":
    main()
