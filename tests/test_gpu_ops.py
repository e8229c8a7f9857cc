"""GPU parity of the op-level C ABI (include/dynavsr_hip.h) against the CPU oracle / fp64 torch.

Run on the MI355X box: python -m pytest tests -m gpu.  Tolerances: the kernels compute in exact
fp32 (v_mfma_f32_32x32x2_f32 == fmaf chain), so the only differences to an fp64 reference are
fp32 round-off: rel-L2 <= 2e-6 * sqrt(K) is typical; 2e-5 is asserted.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from dynavsr_amd import hipops
    return hipops


def dev(t):
    return t.float().contiguous().cuda()


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape) * scale)


ACT = {0: lambda v: v, 1: lambda v: F.leaky_relu(v, 0.1), 2: F.relu}


@pytest.mark.parametrize("cin,cout,ks,stride,h,w", [
    (64, 64, 3, 1, 24, 40), (3, 64, 3, 1, 17, 33), (64, 216, 3, 1, 13, 35), (64, 3, 3, 1, 16, 64),
    (64, 64, 3, 2, 24, 40), (64, 64, 3, 2, 22, 34), (64, 64, 1, 1, 9, 70), (320, 64, 1, 1, 16, 32),
    (128, 64, 3, 1, 8, 32), (16, 24, 3, 1, 5, 7),
])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv2d_forward(ops, cin, cout, ks, stride, h, w, act):
    x = rnd(2, cin, h, w, seed=1)
    wt = rnd(cout, cin, ks, ks, seed=2, scale=1 / np.sqrt(cin * ks * ks))
    b = rnd(cout, seed=3, scale=0.1)
    ref = ACT[act](F.conv2d(x, wt, b, stride, ks // 2))
    got = ops.conv2d_forward(dev(x), dev(wt), dev(b), stride=stride, act=act)
    assert got.shape == ref.shape
    assert relerr(got, ref) < TOL


def test_conv2d_two_inputs_broadcast_residual(ops):
    """cat([nbr, ref]) with ref shared by the 5 frames of a clip + residual add after the act."""
    x0, x1 = rnd(10, 64, 12, 36, seed=1), rnd(2, 64, 12, 36, seed=2)
    wt, b = rnd(64, 128, 3, 3, seed=3, scale=0.03), rnd(64, seed=4, scale=0.1)
    res = rnd(10, 64, 12, 36, seed=5)
    ref = F.leaky_relu(F.conv2d(torch.cat([x0, x1.repeat_interleave(5, 0)], 1), wt, b, 1, 1), 0.1) + res
    got = ops.conv2d_forward(dev(x0), dev(wt), dev(b), act=1, x1=dev(x1), res=dev(res), x1_bdiv=5)
    assert relerr(got, ref) < TOL


def test_conv2d_pixel_shuffle(ops):
    x, wt, b = rnd(1, 64, 10, 34, seed=1), rnd(256, 64, 3, 3, seed=2, scale=0.04), rnd(256, seed=3)
    ref = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, wt, b, 1, 1), 2), 0.1)
    got = ops.conv2d_forward(dev(x), dev(wt), dev(b), act=1, pixel_shuffle=2)
    assert got.shape == ref.shape and relerr(got, ref) < TOL


def test_conv2d_no_bias(ops):
    x, wt = rnd(1, 8, 6, 6, seed=1), rnd(4, 8, 3, 3, seed=2)
    assert relerr(ops.conv2d_forward(dev(x), dev(wt)), F.conv2d(x, wt, None, 1, 1)) < TOL


def test_conv2d_rejects_unsupported(ops):
    x, wt = dev(rnd(1, 8, 6, 6)), dev(rnd(4, 8, 5, 5))
    with pytest.raises(RuntimeError, match="ks=5"):
        ops.conv2d_forward(x, wt)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.conv2d_forward(x.cpu(), dev(rnd(4, 8, 3, 3)))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mdcn_forward_golden(ops, tag):
    g = load_golden("dcn_" + tag)
    x, off, m, w, b = (dev(torch.from_numpy(g[k])) for k in ("x", "offset", "mask", "weight", "bias"))
    got = ops.mdcn_forward(x, off, m, w, b, 1, 1, 1, 1, int(g["dg"]))
    assert relerr(got, g["out"]) < TOL


@pytest.mark.parametrize("n,c,dg,cout,h,w,stride,pad,dil,std", [
    (2, 64, 8, 64, 20, 36, 1, 1, 1, 2.0), (1, 64, 8, 64, 45, 80, 1, 1, 1, 0.5),
    (1, 32, 8, 80, 9, 33, 1, 1, 1, 5.0), (1, 128, 8, 128, 12, 16, 1, 1, 1, 1.0),
    (1, 16, 4, 12, 11, 13, 2, 1, 1, 1.0), (1, 16, 2, 8, 10, 10, 1, 2, 2, 1.0),
])
def test_mdcn_forward_vs_oracle(ops, n, c, dg, cout, h, w, stride, pad, dil, std):
    from oracle import dcn as odcn
    ho = (h + 2 * pad - (2 * dil + 1)) // stride + 1
    wo = (w + 2 * pad - (2 * dil + 1)) // stride + 1
    x = rnd(n, c, h, w, seed=1)
    off = rnd(n, dg * 18, ho, wo, seed=2, scale=std)
    m = torch.from_numpy(np.random.RandomState(3).random_sample((n, dg * 9, ho, wo)))
    wt, b = rnd(cout, c, 3, 3, seed=4, scale=1 / np.sqrt(9 * c)), rnd(cout, seed=5, scale=0.1)
    ref = odcn.forward(x, off, m, wt, b, stride, pad, dil, 1, dg)
    got = ops.mdcn_forward(dev(x), dev(off), dev(m), dev(wt), dev(b), stride, pad, dil, 1, dg)
    assert relerr(got, ref) < TOL
    got = ops.mdcn_forward(dev(x), dev(off), dev(m), dev(wt), None, stride, pad, dil, 1, dg, act=1)
    assert relerr(got, F.leaky_relu(ref - b.view(1, -1, 1, 1), 0.1)) < TOL


@pytest.mark.parametrize("n,cout,h,w,std", [(2, 64, 20, 36, 1.0), (1, 64, 45, 80, 3.0), (1, 80, 9, 33, 12.0),
                                            (1, 64, 8, 32, 200.0)])
def test_mdcn_forward_fast_vs_oracle(ops, n, cout, h, w, std):
    """LDS-sampler kernel incl. its global fallback: std=12/200 px pushes most samples outside the
    staged +-4 px window (and outside the image)."""
    from oracle import dcn as odcn
    c, dg = 64, 8
    x = rnd(n, c, h, w, seed=1)
    off = rnd(n, dg * 18, h, w, seed=2, scale=std)
    off[:, :, 0, :] = torch.round(off[:, :, 0, :])          # exactly-integer sampling positions
    m = torch.from_numpy(np.random.RandomState(3).random_sample((n, dg * 9, h, w)))
    wt, b = rnd(cout, c, 3, 3, seed=4, scale=1 / np.sqrt(9 * c)), rnd(cout, seed=5, scale=0.1)
    ref = odcn.forward(x, off, m, wt, b, 1, 1, 1, 1, dg)
    got = ops.mdcn_forward_fast(dev(x), dev(off), dev(m), dev(wt), dev(b), dg)
    assert relerr(got, ref) < TOL
    assert relerr(ops.mdcn_forward_fast(dev(x), dev(off), dev(m), dev(wt), None, dg, act=1),
                  F.leaky_relu(ref - b.view(1, -1, 1, 1), 0.1)) < TOL


def test_mdcn_pack_forward(ops):
    """offset/mask taken from the raw 216-channel conv output, sigmoid inside the sampler."""
    from oracle import dcn as odcn
    x, om = rnd(2, 64, 14, 38, seed=1), rnd(2, 216, 14, 38, seed=2, scale=1.5)
    wt, b = rnd(64, 64, 3, 3, seed=3, scale=0.04), rnd(64, seed=4, scale=0.1)
    ref = odcn.forward(x, om[:, :144].contiguous(), torch.sigmoid(om[:, 144:]).contiguous(), wt, b, 1, 1,
                       1, 1, 8)
    got = ops.mdcn_pack_forward(dev(x), dev(om), dev(wt), dev(b), 8)
    assert relerr(got, ref) < TOL


def test_mdcn_rejects_unsupported(ops):
    x = dev(rnd(1, 24, 6, 6))
    with pytest.raises(RuntimeError, match="C/dg"):
        ops.mdcn_forward(x, dev(rnd(1, 72, 6, 6)), dev(rnd(1, 36, 6, 6)), dev(rnd(8, 24, 3, 3)), None, 1,
                         1, 1, 1, 4)


@pytest.mark.parametrize("scale,mul,h,w", [(2, 1.0, 9, 13), (2, 2.0, 4, 4), (2, 2.0, 45, 80), (2, 1.0, 11, 20), (4, 1.0, 16, 16), (4, 1.0, 7, 5)])
def test_upsample_bilinear(ops, scale, mul, h, w):
    x = rnd(2, 3, h, w, seed=1).requires_grad_()
    ref = F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False) * mul
    got = ops.upsample_bilinear(dev(x.detach()), scale, mul)
    assert relerr(got, ref) < 1e-6
    g = rnd(*ref.shape, seed=2)
    (gref,) = torch.autograd.grad(ref, x, g)
    assert relerr(ops.upsample_bilinear_backward(dev(g), scale, mul), gref) < 1e-6


@pytest.mark.parametrize("h,w", [(16, 16), (10, 14), (9, 7)])
def test_pool3s2(ops, h, w):
    x = rnd(2, 5, h, w, seed=1).float().double().requires_grad_()   # fp32-representable values
    rmax, ravg = F.max_pool2d(x, 3, 2, 1), F.avg_pool2d(x, 3, 2, 1)
    gmax, gavg = ops.pool3s2(dev(x.detach()))
    assert relerr(gmax, rmax) == 0 and relerr(gavg, ravg) < 1e-6
    g1, g2 = rnd(*rmax.shape, seed=2), rnd(*rmax.shape, seed=3)
    (gref,) = torch.autograd.grad([rmax, ravg], x, [g1, g2])
    assert relerr(ops.pool3s2_backward(dev(x.detach()), dev(g1), dev(g2)), gref) < 1e-6


def test_tsa_gate_and_blend(ops):
    b, n, c, h, w = 2, 5, 64, 6, 10
    emb = rnd(b, n, c, h, w, seed=1, scale=0.3).requires_grad_()
    ref_ = rnd(b, c, h, w, seed=2, scale=0.3).requires_grad_()
    al = rnd(b, n, c, h, w, seed=3).requires_grad_()
    cor = torch.sigmoid((emb * ref_.unsqueeze(1)).sum(2))
    gated = (al * cor.unsqueeze(2)).reshape(b, n * c, h, w)
    gcor, ggated = ops.tsa_gate(dev(emb.detach()), dev(ref_.detach()), dev(al.detach()))
    assert relerr(gcor, cor) < 1e-6 and relerr(ggated, gated) < 1e-6
    g = rnd(*gated.shape, seed=4)
    r_emb, r_ref, r_al = torch.autograd.grad(gated, [emb, ref_, al], g)
    g_emb, g_ref, g_al = ops.tsa_gate_backward(dev(emb.detach()), dev(ref_.detach()), dev(al.detach()),
                                               gcor, dev(g))
    assert relerr(g_emb, r_emb) < 1e-5 and relerr(g_ref, r_ref) < 1e-5 and relerr(g_al, r_al) < 1e-6

    fea, att, add = (rnd(b, c, h, w, seed=s).requires_grad_() for s in (5, 6, 7))
    out = fea * torch.sigmoid(att) * 2 + add
    assert relerr(ops.tsa_blend(dev(fea.detach()), dev(att.detach()), dev(add.detach())), out) < 1e-6
    go = rnd(*out.shape, seed=8)
    r_fea, r_att = torch.autograd.grad(out, [fea, att], go)
    pre = rnd(*out.shape, seed=9)
    g_att = dev(pre)
    g_fea = ops.tsa_blend_backward(dev(fea.detach()), dev(att.detach()), dev(go), g_att)
    assert relerr(g_fea, r_fea) < 1e-6 and relerr(g_att, r_att + pre) < 1e-6


@pytest.mark.parametrize("cin,cout,ks,stride,h,w", [
    (64, 64, 3, 1, 12, 40), (3, 64, 3, 1, 9, 33), (64, 216, 3, 1, 7, 35), (64, 3, 3, 1, 8, 32),
    (64, 64, 3, 2, 12, 40), (64, 64, 1, 1, 9, 70), (320, 64, 1, 1, 8, 32), (16, 24, 3, 1, 5, 7),
    # wide staging of the weight gradient (W % 4 == 0: float4 windows; W even: float2) incl. partly filled 64-channel
    # blocks (100 = 64 + 36, 104 = 64 + 40: the second block runs the wide path with channels past the end masked)
    (100, 104, 3, 1, 12, 40), (96, 100, 1, 1, 8, 32), (64, 64, 3, 1, 10, 34), (100, 64, 3, 1, 6, 38), (64, 64, 3, 1, 45, 80),
])
def test_conv2d_backward(ops, cin, cout, ks, stride, h, w):
    x = rnd(2, cin, h, w, seed=1).requires_grad_()
    wt = rnd(cout, cin, ks, ks, seed=2, scale=1 / np.sqrt(cin * ks * ks)).requires_grad_()
    b = rnd(cout, seed=3, scale=0.1).requires_grad_()
    y = F.conv2d(x, wt, b, stride, ks // 2)
    gy = rnd(*y.shape, seed=4)
    rx, rw, rb = torch.autograd.grad(y, [x, wt, b], gy)
    gx, _, gw, gb = ops.conv2d_backward(dev(gy), dev(x.detach()), dev(wt.detach()), stride=stride)
    assert relerr(gx, rx) < TOL and relerr(gw, rw) < TOL and relerr(gb, rb) < TOL


def test_conv2d_backward_two_inputs(ops):
    x0, x1 = rnd(3, 64, 10, 36, seed=1).requires_grad_(), rnd(3, 64, 10, 36, seed=2).requires_grad_()
    wt = rnd(64, 128, 3, 3, seed=3, scale=0.03).requires_grad_()
    y = F.conv2d(torch.cat([x0, x1], 1), wt, None, 1, 1)
    gy = rnd(*y.shape, seed=4)
    r0, r1, rw = torch.autograd.grad(y, [x0, x1, wt], gy)
    g0, g1, gw, gb = ops.conv2d_backward(dev(gy), dev(x0.detach()), dev(wt.detach()), x1=dev(x1.detach()))
    assert relerr(g0, r0) < TOL and relerr(g1, r1) < TOL and relerr(gw, rw) < TOL
    assert relerr(gb, gy.sum((0, 2, 3))) < TOL


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mdcn_backward_golden(ops, tag):
    g = load_golden("dcn_" + tag)
    x, off, m, w, go = (dev(torch.from_numpy(g[k])) for k in ("x", "offset", "mask", "weight", "gout"))
    gx, goff, gm, gw, gb = ops.mdcn_backward(x, off, m, w, go, 1, 1, 1, 1, int(g["dg"]))
    for got, key in ((gx, "gx"), (goff, "goffset"), (gm, "gmask"), (gw, "gweight"), (gb, "gbias")):
        assert relerr(got, g[key]) < TOL, key


@pytest.mark.parametrize("std,h,w", [(1.5, 14, 34), (8.0, 20, 40), (60.0, 9, 33)])
def test_mdcn_backward_vs_oracle(ops, std, h, w):
    """C/dg = 8 takes the LDS-privatised kernel; std=8/60 px exercise its global fallback and the
    window border, the first rows sit exactly on integer sampling positions."""
    from oracle import dcn as odcn
    n, c, dg, cout = 2, 64, 8, 64
    x, off = rnd(n, c, h, w, seed=1), rnd(n, dg * 18, h, w, seed=2, scale=std)
    off[:, :, 0, :] = torch.round(off[:, :, 0, :])
    m = torch.from_numpy(np.random.RandomState(3).random_sample((n, dg * 9, h, w)))
    wt, go = rnd(cout, c, 3, 3, seed=4, scale=0.04), rnd(n, cout, h, w, seed=5)
    ref = odcn.backward(x.float().double(), off.float().double(), m.float().double(), wt.float().double(), True,
                        go.float().double(), 1, 1, 1, 1, dg)
    got = ops.mdcn_backward(dev(x), dev(off), dev(m), dev(wt), dev(go), 1, 1, 1, 1, dg)
    for a, b_, name in zip(got, ref, ("gx", "goffset", "gmask", "gw", "gb")):
        assert relerr(a, b_) < TOL, name


def test_mdcn_backward_fp32_contractions_switch():
    """Since round 6 the fused backward runs its two contractions (dcol = W^T gout, the tile's weight gradient) on the bf16
    pipe under the exact 3-way split where Cout = 64, W % 4 == 0 and the tensors are 16-byte aligned (the 20x40 case of
    test_mdcn_backward_vs_oracle and the goldens; 14x34 / 9x33 and EDVR-L's 128-cout layers keep the fp32 MFMAs).
    DVSR_DCN_BWD=fp32 puts every shape on the fp32 MFMAs again: the same cases, same bars, in a child (the switch is read
    once per process)."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "test_mdcn_backward_vs_oracle or test_mdcn_backward_golden"],
                       env=dict(os.environ, DVSR_DCN_BWD="fp32"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("std", [1.5, 8.0])
def test_mdcn_16_channels_per_group_vs_oracle(ops, std):
    """EDVR-L's DCN: C = 128, dg = 8 -> 16 channels per deformable group.  The LDS-sampler forward and the fused
    backward walk such a group as two 8-channel chunks that share offsets and masks (the chunks' offset / mask
    gradients are added); both against the C oracle, with offsets that leave the staged window too."""
    from oracle import dcn as odcn
    n, c, dg, cout, h, w = 2, 128, 8, 128, 12, 36
    x, off = rnd(n, c, h, w, seed=11), rnd(n, dg * 18, h, w, seed=12, scale=std)
    m = torch.from_numpy(np.random.RandomState(13).random_sample((n, dg * 9, h, w)))
    wt, b, go = rnd(cout, c, 3, 3, seed=14, scale=0.03), rnd(cout, seed=15), rnd(n, cout, h, w, seed=16)
    yo = odcn.forward(x.float().double(), off.float().double(), m.float().double(), wt.float().double(),
                      b.float().double(), 1, 1, 1, 1, dg)
    y = ops.mdcn_forward_fast(dev(x), dev(off), dev(m), dev(wt), dev(b), dg)
    assert relerr(y, yo) < TOL
    ref = odcn.backward(x.float().double(), off.float().double(), m.float().double(), wt.float().double(), True,
                        go.float().double(), 1, 1, 1, 1, dg)
    got = ops.mdcn_backward(dev(x), dev(off), dev(m), dev(wt), dev(go), 1, 1, 1, 1, dg)
    for a_, b_, name in zip(got, ref, ("gx", "goffset", "gmask", "gw", "gb")):
        assert relerr(a_, b_) < TOL, name


def test_charbonnier(ops):
    x, y = rnd(2, 3, 40, 52, seed=1).requires_grad_(), rnd(2, 3, 40, 52, seed=2)
    ref = torch.mean(torch.sqrt((x - y) ** 2 + 1e-6))
    (rg,) = torch.autograd.grad(ref * 3.0, x)
    xg = dev(x.detach()).requires_grad_()
    got = ops.charbonnier(xg, dev(y))
    (gg,) = torch.autograd.grad(got * 3.0, xg)
    assert abs(float(got) - float(ref)) < 1e-6 * float(ref) and relerr(gg, rg) < 1e-6


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 12, 40), (1, 128, 128, 9, 33), (3, 24, 72, 7, 20), (1, 64, 216, 16, 32)])
def test_conv3x3_wgrad_bf16(ops, n, cin, cout, h, w):
    """Weight / bias gradient with bf16 operands (conv2d_wgrad_bf16.hip): against the fp64 gradient of the SAME
    bf16-rounded operands (only fp32 accumulation order differs: 2e-5), and against the unrounded fp64 gradient within
    bf16's reach (2^-9 per operand: 1e-2); ragged tiles, channel counts off the 64-blocks, the bias gradient exact-ish."""
    import torch.nn.functional as F
    from dynavsr_amd import _lib as L
    x, gy = rnd(n, cin, h, w, seed=1), rnd(n, cout, h, w, seed=2)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)

    def ref(xx, gg):
        (gw,) = torch.autograd.grad(F.conv2d(xx.double(), wt, padding=1), wt, gg.double())
        return gw
    gw_exact = ref(x, gy)
    gw_round = ref(x.bfloat16().float(), gy.bfloat16().float())
    xg, gg = dev(x), dev(gy)
    gw, gb = torch.empty(cout, cin, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
    d = L.Conv2dDesc(L.ptr(xg), None, None, None, None, None, n, cin, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_backward_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    L.check(L.lib().dvsr_conv2d_wgrad_bf16(d, L.ptr(gg), L.ptr(gw), L.ptr(gb), ws.data_ptr(), ws.numel(), L.stream()),
            "dvsr_conv2d_wgrad_bf16")
    assert relerr(gw, gw_round) < 2e-5, relerr(gw, gw_round)
    assert 1e-5 < relerr(gw, gw_exact) < 1e-2
    assert relerr(gb, gy.bfloat16().double().sum(dim=(0, 2, 3))) < 1e-5        # fp32 sum of the bf16-rounded operand
    assert relerr(gb, gy.double().sum(dim=(0, 2, 3))) < 5e-3


@pytest.mark.parametrize("n,cin,cout,h,w,pad", [(2, 64, 64, 12, 40, 1), (1, 128, 128, 9, 33, 1), (3, 24, 72, 7, 20, 1),
                                                (1, 64, 216, 16, 32, 1), (5, 64, 64, 46, 82, 0), (8, 64, 64, 44, 80, 1)])
def test_conv3x3_wgrad_split3(ops, n, cin, cout, h, w, pad):
    """Weight / bias gradient on the bf16 pipe with the exact 3-way split of both operands (conv2d_wgrad_split3_kernel, what
    the plans run for their 3x3 stride-1 layers): against the fp64 gradient at the fp32 kernels' bar (2e-6: the six products
    kept cover 2^-24 of every fp32 product), the bias gradient to fp32 summation round-off; ragged tiles, channel counts off
    the 64-blocks, an explicitly padded input (pad = 0: the estimators' layers), a batch of frames."""
    import torch.nn.functional as F
    from dynavsr_amd import _lib as L
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    x, gy = rnd(n, cin, h, w, seed=1), rnd(n, cout, ho, wo, seed=2)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (gw_exact,) = torch.autograd.grad(F.conv2d(x.double(), wt, padding=pad), wt, gy.double())
    xg, gg = dev(x), dev(gy)
    gw, gb = torch.empty(cout, cin, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
    d = L.Conv2dDesc(L.ptr(xg), None, None, None, None, None, n, cin, 0, h, w, cout, 3, 1, pad, 0, 0, 1, 0, 0)
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_backward_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    L.check(L.lib().dvsr_conv2d_wgrad_split3(d, L.ptr(gg), L.ptr(gw), L.ptr(gb), ws.data_ptr(), ws.numel(), L.stream()),
            "dvsr_conv2d_wgrad_split3")
    assert relerr(gw, gw_exact) < 2e-6, relerr(gw, gw_exact)
    assert relerr(gb, gy.double().sum(dim=(0, 2, 3))) < 2e-6


@pytest.mark.parametrize("env", [{"DVSR_WGRAD_S3_KYS_BELOW": "0"}, {"DVSR_WGRAD_S3_KYS_BELOW": "0", "DVSR_WGRAD_S3W": "0"},
                                 {"DVSR_WGRAD_S3V": "0"}, {"DVSR_WGRAD_S3_WGS": "96"}],
                         ids=["eight_waves", "four_waves", "round4_schedule", "row_split_96_workgroups"])
def test_conv3x3_wgrad_split3_other_schedules(env):
    """test_conv3x3_wgrad_split3's shapes are small: by default they all run the row-split form of conv2d_wgrad_split3v_kernel
    (one kernel row per workgroup, two workgroups per CU).  The same cases on the forms with all nine taps per workgroup (what
    the large layers run: eight waves on 16x16x32 MFMAs for the float4-staged launches, four waves on 32x32x16 for the rest
    and under DVSR_WGRAD_S3W=0), on the round-4 kernel the A/B switch keeps, and on the row split with few workgroups (several
    tiles per workgroup, the last one re-staged).  The switches are read once per process: each runs in a child."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "(test_conv3x3_wgrad_split3 and not other_schedules) or test_split_wgrad_scales_range_and_non_finite"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("n,cin,h,w,act", [(1, 320, 44, 80, 1), (2, 64, 9, 12, 0), (3, 16, 1, 4, 2), (1, 320, 180, 320, 1), (2, 112, 33, 20, 1)])
def test_conv1x1_dual(ops, n, cin, h, w, act):
    """conv1x1_dual_kernel (the TSA's fea_fusion + sAtt_1 over one input, EDVR_arch.py:183-202) against fp64: both outputs, the
    three epilogues, pixel counts off the 128-pixel tiles (a partly filled last tile, a single 4-pixel group), one to twenty
    16-channel chunks (the DMA ring with fewer chunks than stages), batches; and what it refuses."""
    from dynavsr_amd import _lib as L
    x = rnd(n, cin, h, w, seed=3)
    w0, w1 = rnd(64, cin, seed=4) / cin ** 0.5, rnd(64, cin, seed=5) / cin ** 0.5
    b0, b1 = rnd(64, seed=6), rnd(64, seed=7)
    f = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.1), 2: F.relu}[act]
    r0 = f(F.conv2d(x.double(), w0.double().view(64, cin, 1, 1), b0.double()))
    r1 = f(F.conv2d(x.double(), w1.double().view(64, cin, 1, 1), b1.double()))
    xg, g = dev(x), [dev(t) for t in (w0, b0, w1, b1)]
    y0, y1 = torch.full((n, 64, h, w), float("nan"), device="cuda"), torch.full((n, 64, h, w), float("nan"), device="cuda")
    L.check(L.lib().dvsr_conv1x1_dual(L.ptr(xg), L.ptr(g[0]), L.ptr(g[1]), L.ptr(g[2]), L.ptr(g[3]), L.ptr(y0), L.ptr(y1),
                                      n, cin, h, w, act, L.stream()), "dvsr_conv1x1_dual")
    assert relerr(y0, r0) < 2e-6 and relerr(y1, r1) < 2e-6, (relerr(y0, r0), relerr(y1, r1))
    # no bias; and the shapes it leaves to the two-launch path
    L.check(L.lib().dvsr_conv1x1_dual(L.ptr(xg), L.ptr(g[0]), None, L.ptr(g[2]), None, L.ptr(y0), L.ptr(y1),
                                      n, cin, h, w, 0, L.stream()), "dvsr_conv1x1_dual")
    assert relerr(y0, F.conv2d(x.double(), w0.double().view(64, cin, 1, 1))) < 2e-6
    assert L.lib().dvsr_conv1x1_dual(L.ptr(xg), L.ptr(g[0]), None, L.ptr(g[2]), None, L.ptr(y0), L.ptr(y1),
                                     n, cin - 8, h, w, 0, L.stream()) == -2          # Cin % 16
    if (h * w) % 4 == 0 and w > 1:
        assert L.lib().dvsr_conv1x1_dual(L.ptr(xg), L.ptr(g[0]), None, L.ptr(g[2]), None, L.ptr(y0), L.ptr(y1),
                                         n, cin, h, w - 1, 0, L.stream()) in (-2, 0)  # (H * W) % 4 unless it happens to hold


def _small_cout_case(n, c, cout, h, w, act, with_res):
    from dynavsr_amd import _lib as L
    x, wt, b = rnd(n, c, h, w, seed=3), rnd(cout, c, 3, 3, seed=4) / (3 * c ** 0.5), rnd(cout, seed=5)
    res = rnd(n, cout, h, w, seed=6) if with_res else None
    f = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.1), 2: F.relu}[act]
    ref = f(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    if with_res:
        ref = ref + res.double()
    y = torch.full((n, cout, h, w), float("nan"), device="cuda")
    xg, wg, bg = dev(x), dev(wt), dev(b)
    rg = dev(res) if with_res else None
    L.check(L.lib().dvsr_conv3x3_small_cout(L.ptr(xg), L.ptr(wg), L.ptr(bg), L.ptr(rg), L.ptr(y), n, c, h, w, cout, act,
                                            L.stream()), "dvsr_conv3x3_small_cout")
    assert relerr(y, ref) < 2e-6, relerr(y, ref)


@pytest.mark.parametrize("n,c,cout,h,w,act,with_res", [(1, 64, 3, 176, 320, 0, True), (2, 64, 3, 13, 60, 1, False),
                                                       (1, 64, 1, 6, 56, 0, True), (1, 64, 2, 7, 116, 2, True),
                                                       (3, 64, 3, 5, 4, 0, False), (1, 64, 3, 720, 1280, 0, True),
                                                       (1, 32, 3, 20, 36, 0, True), (1, 64, 4, 9, 20, 1, False),
                                                       (2, 64, 3, 10, 30, 0, True)])
def test_conv3x3_small_cout(ops, n, c, cout, h, w, act, with_res):
    """EDVR's conv_last (64 -> 3 + the base frame, EDVR_arch.py:307-312) at op level against fp64: whole tiles, tiles cut by
    either image edge, images smaller than a tile, 1 - 4 outputs, the three epilogues, batches, the 720 x 1280 layer, 32
    channels, a width off the 16-byte groups."""
    _small_cout_case(n, c, cout, h, w, act, with_res)


def test_conv3x3_small_cout_matrix_pipe_form():
    """The same cases with DVSR_CONV_LAST_MFMA=1: conv3x3_small_cout_mfma_kernel (a 27-row GEMM over the halo tile on the fp32
    matrix pipe + a shift-add through the LDS; built in round 5, measured no faster than the vector-ALU kernel and therefore
    off by default).  The switch is read once per process: a child."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "test_conv3x3_small_cout and not matrix_pipe_form"],
                       env=dict(os.environ, DVSR_CONV_LAST_MFMA="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


def test_inner_loss_tail(ops):
    """loss_pix + 10 * F.l1_loss(SLR, SLR_fixed) (test_dynavsr.py:264-274) as one native reduction: value, the
    pass-through gradient of the pixel loss, the sign gradient of the L1 term (sign(0) = 0 like torch), ragged
    and > 256 K element sizes."""
    import torch.nn.functional as F
    for shape, seed in (((1, 5, 3, 44, 80), 3), ((7, 13), 4), ((2, 5, 3, 176, 320), 5)):
        x, y = rnd(*shape, seed=seed), rnd(*shape, seed=seed + 10)
        x.view(-1)[::7] = y.view(-1)[::7]                      # exact ties
        x.requires_grad_()
        base = torch.tensor(0.37, requires_grad=True)
        ref = base * 2.0 + 10.0 * F.l1_loss(x, y)
        rgx, rgb = torch.autograd.grad(ref * 1.5, [x, base])
        xg, bg = dev(x.detach()).requires_grad_(), dev(base.detach()).requires_grad_()
        got = ops.inner_loss(bg * 2.0, xg, dev(y), 10.0)
        ggx, ggb = torch.autograd.grad(got * 1.5, [xg, bg])
        assert abs(float(got) - float(ref)) < 2e-6 * abs(float(ref))
        assert relerr(ggx, rgx) < 1e-6 and abs(float(ggb) - float(rgb)) < 1e-6
        assert float(ggx.view(-1)[::7].abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="inner_loss"):
        ops.inner_loss(dev(torch.zeros(())), dev(torch.zeros(3, 4)), dev(torch.zeros(4, 3)))


@pytest.mark.parametrize("kind", ["adam", "sgd"])
def test_native_optimizer_matches_torch(kind):
    """dvsr_adam_step / dvsr_sgd_step against torch.optim over several steps, ragged tensor list (one
    parameter without a gradient, > 48 tensors so that the launch is split)."""
    from dynavsr_amd import optim
    torch.manual_seed(0)
    shapes = [(64, 64, 3, 3), (64,), (3, 64, 3, 3), (216, 64, 3, 3), (1,)] + [(5, 7)] * 50
    a = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    if kind == "adam":
        oa, ob = optim.Adam(a, lr=1e-2, betas=(0.9, 0.99)), torch.optim.Adam(b, lr=1e-2, betas=(0.9, 0.99))
    else:
        oa, ob = optim.SGD(a, lr=1e-2), torch.optim.SGD(b, lr=1e-2)
    for step in range(4):
        for i, (p, q) in enumerate(zip(a, b)):
            g = torch.randn_like(p) * (10.0 ** (step - 2))
            p.grad, q.grad = (None, None) if i == 4 else (g.clone(), g.clone())
        oa.step(); ob.step()
        for p, q in zip(a, b):
            assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max() + 1e-3), (kind, step)
    assert float((a[4] - b[4]).abs().max()) == 0.0
    if kind == "adam":
        assert oa.state[a[0]]['step'] == 4 and relerr(oa.state[a[0]]['exp_avg'], ob.state[b[0]]['exp_avg']) < 1e-6


# ---- halo-by-DMA 3x3 kernel (conv2d_dma_kernel): chosen for grids of >= 700 32x4x32 workgroups ------------------
@pytest.mark.parametrize("n,cin,cout,h,w,act,res", [
    (4, 64, 64, 98, 160, 1, True),     # 5 x 25 x 4 tiles of 64 output channels; the last tile row holds 2 of 4 rows
    (3, 72, 40, 90, 200, 0, False),    # W % 32 != 0 (partial right tiles), Cout % 32 != 0, nine 8-channel chunks
    (1, 64, 64, 180, 320, 2, False),   # the reconstruction trunk of the headline clip
    (2, 8, 64, 128, 128, 1, False),    # ONE chunk: prologue only, no steady state
])
def test_conv3x3_dma_halo(n, cin, cout, h, w, act, res, monkeypatch):
    """Forward and data gradient through dvsr_conv2d_forward_packed / _dgrad_packed on grids large enough for
    the DMA-halo kernel, against fp64 torch on the CPU; dvsr_conv2d_packed_geometry confirms which kernel ran.
    (DVSR_CONV_WINO=0: the Winograd kernel would take most of these shapes, test_conv3x3_winograd.)"""
    import ctypes
    from dynavsr_amd import _lib as L, tofops
    monkeypatch.setenv("DVSR_CONV_WINO", "0")
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1 / np.sqrt(cin * 9))
    b = rnd(cout, seed=3, scale=0.1)
    r = rnd(n, cout, h, w, seed=4) if res else None
    gy = rnd(n, cout, h, w, seed=5)
    xd = dev(x).requires_grad_(True)
    d = L.Conv2dDesc(L.ptr(xd), None, None, None, None, None, n, cin, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
    assert list(geo)[0] == 8 and list(geo)[3] == 1, "expected the DMA-halo kernel, got %s" % list(geo)
    got = tofops.conv(xd, dev(wt), dev(b), dev(r) if res else None, act)
    x64 = x.double().requires_grad_(True)
    ref = ACT[act](F.conv2d(x64, wt.double(), b.double(), 1, 1))
    if res:
        ref = ref + r.double()
    assert relerr(got.detach(), ref.detach()) < TOL
    got.backward(dev(gy))
    ref.backward(gy.double())
    assert relerr(xd.grad, x64.grad) < TOL


# ---- Winograd F(2x2, 3x3) kernel (conv2d_wino_kernel): the large 3x3 / stride-1 layers ----------------------------------
@pytest.mark.parametrize("n,c0,c1,cout,h,w,act,res,ps,force", [
    (5, 64, 0, 64, 180, 320, 2, False, 0, False),   # fe_rb of the headline clip: 4x64-pixel tiles, all full
    (4, 64, 64, 64, 96, 128, 1, True, 0, False),    # two inputs (16 chunks), residual
    (1, 64, 0, 256, 90, 160, 1, False, 2, True),    # PixelShuffle(2) store, partial tile rows and columns
    (3, 72, 0, 40, 90, 200, 0, False, 0, True),     # Cout % 32 != 0, nine chunks, ragged in both directions
    (2, 8, 8, 64, 44, 80, 1, True, 0, True),        # two chunks: first and last block only
    (2, 64, 0, 216, 96, 128, 0, False, 0, True),    # the offset / mask conv: the last 64-cout block holds 24 (its upper-half waves idle)
])
@pytest.mark.parametrize("pipe", ["bf16x3", "fp32"])
def test_conv3x3_winograd(n, c0, c1, cout, h, w, act, res, ps, force, pipe, monkeypatch):
    _winograd_case(n, c0, c1, cout, h, w, act, res, ps, force, pipe, monkeypatch)


@pytest.mark.parametrize("n,c0,c1,cout,h,w,act,res,ps", [
    (16, 64, 0, 64, 44, 80, 2, False, 0),    # the trunk of a batched inner step: 240 tiles of 16x16 pixels, last tile row ragged
    (16, 64, 0, 64, 44, 80, 0, True, 0),     # ... with the residual
    (80, 64, 64, 64, 44, 80, 1, False, 0),   # 16 frames x 5: two inputs, five rounds instead of six
    (16, 64, 0, 256, 44, 80, 1, False, 2),   # PixelShuffle(2) store
])
def test_conv3x3_winograd_16x16_tiles(n, c0, c1, cout, h, w, act, res, ps, monkeypatch):
    """The bf16x3 kernel's third tile shape (TC = 8: 16 x 16 pixels), which the cost model takes where it saves a round of
    workgroups: the 44x80 levels of the batched inner step."""
    geo = _winograd_case(n, c0, c1, cout, h, w, act, res, ps, True, "bf16x3", monkeypatch)
    assert geo[1] == 16, geo


def _winograd_case(n, c0, c1, cout, h, w, act, res, ps, force, pipe, monkeypatch):
    """Forward (and data gradient for single plain inputs) on the Winograd kernels against fp64 torch; the geometry query
    confirms the kernel.  force: DVSR_CONV_WINO=2 takes it wherever it is eligible (the cost model would keep the direct
    kernel on these small grids).  pipe: the sixteen GEMMs on the bf16 matrix pipe with the exact 3-way operand split
    (conv2d_wino3.hip, the default: geometry code 4) or on the fp32 MFMA (conv2d_wino.hip, DVSR_CONV_WINO3=0: code 3) --
    both are held to the same fp32 bar."""
    import ctypes
    from dynavsr_amd import _lib as L
    if force:
        monkeypatch.setenv("DVSR_CONV_WINO", "2")
    monkeypatch.setenv("DVSR_CONV_WINO3", "1" if pipe == "bf16x3" else "0")
    monkeypatch.setenv("DVSR_CONV_WINO5", "0")   # (these cases hold the F(2x2) kernels; F(4x4): test_conv3x3_winograd_f4x4)
    cin = c0 + c1
    x0 = rnd(n, c0, h, w, seed=1)
    x1 = rnd(n, c1, h, w, seed=6) if c1 else None
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1 / np.sqrt(cin * 9))
    b = rnd(cout, seed=3, scale=0.1)
    r = rnd(n, cout, h, w, seed=4) if res else None
    d0, d1, dw, db_, dr = dev(x0), (dev(x1) if c1 else None), dev(wt), dev(b), (dev(r) if res else None)
    y = torch.empty((n, cout // 4, 2 * h, 2 * w) if ps else (n, cout, h, w), device="cuda")
    d = L.Conv2dDesc(L.ptr(d0), L.ptr(d1), L.ptr(dw), L.ptr(db_), L.ptr(dr), L.ptr(y), n, c0, c1, h, w, cout, 3, 1, 1,
                     act, ps, 1, 0, 0)
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
    assert list(geo)[3] == (4 if pipe == "bf16x3" else 3), "expected the Winograd kernel, got %s" % list(geo)
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "dvsr_conv2d_forward_packed")
    x = torch.cat([x0, x1], 1) if c1 else x0
    ref = ACT[act](F.conv2d(x.double(), wt.double(), b.double(), 1, 1))
    if res:
        ref = ref + r.double()
    if ps:
        ref = F.pixel_shuffle(ref, 2)
    assert relerr(y, ref) < 2e-6
    if c1 == 0 and not ps:   # data gradient: the same kernel over the transposed, tap-mirrored weights
        gy = rnd(n, cout, h, w, seed=5)
        gx = torch.empty(n, c0, h, w, device="cuda")
        dgy = dev(gy)
        L.check(L.lib().dvsr_conv2d_dgrad_packed(d, L.ptr(dgy), L.ptr(gx), ws.data_ptr(), ws.numel(), L.stream()),
                "dvsr_conv2d_dgrad_packed")
        ref_g = torch.nn.grad.conv2d_input((n, c0, h, w), wt.double(), gy.double(), 1, 1)
        assert relerr(gx, ref_g) < 2e-6
    return list(geo)


# ---- Winograd F(4x4, 3x3) on the bf16 pipe (conv2d_wino5_kernel, geometry code 5): the no-grad forwards of the large layers ----
def _f4_conv(x0, x1, wt, b, r, act, ps, monkeypatch, mode="2", dgrad_of=None):
    """One launch through dvsr_conv2d_forward_packed (or _dgrad_packed) with DVSR_CONV_WINO5=mode; returns (y, geometry)."""
    import ctypes
    from dynavsr_amd import _lib as L
    monkeypatch.setenv("DVSR_CONV_WINO", "2")
    monkeypatch.setenv("DVSR_CONV_WINO3", "1")
    monkeypatch.setenv("DVSR_CONV_WINO5", mode)
    n, c0, h, w = x0.shape
    c1 = x1.shape[1] if x1 is not None else 0
    cout = wt.shape[0]
    d0, d1, dw = dev(x0), (dev(x1) if c1 else None), dev(wt)
    db_, dr = (dev(b) if b is not None else None), (dev(r) if r is not None else None)
    y = torch.empty((n, cout // 4, 2 * h, 2 * w) if ps else (n, cout, h, w), device="cuda")
    d = L.Conv2dDesc(L.ptr(d0), L.ptr(d1), L.ptr(dw), L.ptr(db_), L.ptr(dr), L.ptr(y), n, c0, c1, h, w, cout, 3, 1, 1, act, ps, 1, 0, 0)
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    if dgrad_of is not None:
        gx = torch.empty(n, c0, h, w, device="cuda")
        dgy = dev(dgrad_of)
        L.check(L.lib().dvsr_conv2d_dgrad_packed(d, L.ptr(dgy), L.ptr(gx), ws.data_ptr(), ws.numel(), L.stream()), "dvsr_conv2d_dgrad_packed")
        return gx.cpu(), list(geo)
    L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "dvsr_conv2d_forward_packed")
    return y.cpu(), list(geo)


@pytest.mark.parametrize("n,c0,c1,cout,h,w,act,res,ps", [
    (5, 64, 0, 64, 180, 320, 2, False, 0),   # fe_rb of the headline clip: 23 x 5 workgroup tiles of 8 x 64 pixels, the last tile row half full
    (4, 64, 64, 64, 96, 128, 1, True, 0),    # two inputs (16 chunks), residual
    (1, 64, 0, 256, 92, 160, 1, False, 2),   # PixelShuffle(2) store, four cout blocks, ragged tile rows and columns
    (3, 72, 0, 40, 88, 200, 0, False, 0),    # Cout % 32 != 0 (the second cout half idle), nine chunks, 200 = 3.125 workgroup tiles wide
    (2, 8, 8, 64, 44, 80, 1, True, 0),       # two chunks: prologue + last chunk only
    (2, 64, 0, 216, 96, 128, 0, False, 0),   # the offset / mask conv: the last 64-cout block holds 24
    (1, 16, 0, 64, 4, 4, 0, False, 0),       # ONE 4 x 4 tile: 31 of the workgroup's 32 tiles outside the image
    (5, 64, 0, 64, 90, 160, 1, True, 0),     # the L2 level of the headline clip: H % 4 == 2, the last tile row is cut (residual read too)
    (2, 64, 0, 128, 90, 160, 1, False, 2),   # ... with the PixelShuffle(2) store
])
@pytest.mark.parametrize("mode", ["2", "3"])
def test_conv3x3_winograd_f4x4(n, c0, c1, cout, h, w, act, res, ps, mode, monkeypatch):
    """conv2d_wino5_kernel -- F(4x4, 3x3), 36 transformed points per 4x4 outputs, the exact 3-way bf16 split of both operands,
    six of nine partial products -- against fp64 torch, both workgroup tile shapes (DVSR_CONV_WINO5=2: 8 x 64 pixels, =3:
    16 x 32).  The bar is 4e-6 rel-L2 (1.5e-6 .. 2.0e-6 observed; F(2x2): 2.4e-7, the direct fp32 sum 4.3e-7): the transform's
    constants (4, -5, 8, 1/24 ...) cost a factor of six, the price of 0.56 x the multiplies; max-abs <= 5e-5 on
    O(1) outputs."""
    cin = c0 + c1
    x0 = rnd(n, c0, h, w, seed=1)
    x1 = rnd(n, c1, h, w, seed=6) if c1 else None
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1 / np.sqrt(cin * 9))
    b = rnd(cout, seed=3, scale=0.1)
    r = rnd(n, cout, h, w, seed=4) if res else None
    y, geo = _f4_conv(x0, x1, wt, b, r, act, ps, monkeypatch, mode)
    assert geo[3] == 5 and geo[1] == (8 if mode == "2" else 16), geo
    x = torch.cat([x0, x1], 1) if c1 else x0
    ref = ACT[act](F.conv2d(x.double(), wt.double(), b.double(), 1, 1))
    if res:
        ref = ref + r.double()
    if ps:
        ref = F.pixel_shuffle(ref, 2)
    assert relerr(y, ref) < 4e-6, relerr(y, ref)
    assert float((y.double() - ref).abs().max()) < 5e-5
    if c1 == 0 and not ps and not res:   # data gradient at op level: the same kernel over the transposed, tap-mirrored weights
        gy = rnd(n, cout, h, w, seed=5)
        gx, geo = _f4_conv(x0, None, wt, None, None, 0, 0, monkeypatch, mode, dgrad_of=gy)
        ref_g = torch.nn.grad.conv2d_input((n, c0, h, w), wt.double(), gy.double(), 1, 1)
        assert relerr(gx, ref_g) < 4e-6


def test_conv3x3_winograd_f4x4_cost_model_and_eligibility(monkeypatch):
    """DVSR_CONV_WINO5 unset: the cost model takes the F(4x4) kernel for the layers it was built for (the 5-frame layers at
    180x320, the 360x640 / 720x1280 tail) and leaves the one-round grids (the N = 1 trunk) and the shapes it cannot store
    whole 4x4 tiles for (H or W not a multiple of 4) to form 4; =0 switches it off."""
    import ctypes
    from dynavsr_amd import _lib as L
    monkeypatch.setenv("DVSR_CONV_WINO3", "1")
    monkeypatch.delenv("DVSR_CONV_WINO", raising=False)

    def geo_of(n, c0, cout, h, w, ps=0):
        x = torch.empty(n, c0, h, w, device="cuda")
        d = L.Conv2dDesc(L.ptr(x), None, None, None, None, None, n, c0, 0, h, w, cout, 3, 1, 1, 0, ps, 1, 0, 0)
        geo = (ctypes.c_int * 4)()
        L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
        return list(geo)

    monkeypatch.delenv("DVSR_CONV_WINO5", raising=False)
    assert geo_of(5, 64, 64, 180, 320)[3] == 5
    assert geo_of(5, 64, 216, 180, 320)[3] == 5
    assert geo_of(1, 64, 256, 360, 640, 2)[3] == 5
    assert geo_of(1, 64, 64, 720, 1280)[3] == 5
    assert geo_of(1, 64, 64, 180, 320)[3] == 4      # one round of workgroups either way: form 4 measures faster
    assert geo_of(5, 64, 64, 90, 160)[3] == 5       # the L2 level (H % 4 == 2: the last tile row is cut): one round instead of two
    assert geo_of(5, 64, 64, 90, 162)[3] != 5       # W % 4 != 0: no 16-byte tile rows (nor any DMA-halo kernel)
    monkeypatch.setenv("DVSR_CONV_WINO5", "0")
    assert geo_of(5, 64, 64, 180, 320)[3] == 4


@pytest.mark.parametrize("name,cout,h,w,ps", [("upconv2", 256, 360, 640, 2), ("HRconv", 64, 720, 1280, 0)])
def test_conv3x3_winograd_f4x4_largest_geometries(name, cout, h, w, ps, monkeypatch):
    """The two largest launches of the headline forward on the F(4x4) kernel: the whole output against form 4 (F(2x2), the
    same split: <= 4e-6), three bands of rows against fp64 torch on the corresponding crops."""
    n, c0 = 1, 64
    x = rnd(n, c0, h, w, seed=11)
    wt = rnd(cout, c0, 3, 3, seed=12, scale=1 / np.sqrt(c0 * 9))
    b = rnd(cout, seed=13, scale=0.1)
    y5, geo5 = _f4_conv(x, None, wt, b, None, 1, ps, monkeypatch, "2")
    y4, geo4 = _f4_conv(x, None, wt, b, None, 1, ps, monkeypatch, "0")
    assert geo5[3] == 5 and geo4[3] == 4
    assert relerr(y5, y4) < 4e-6
    s = 2 if ps else 1
    for r0, r1 in ((0, 12), (h // 2 - 6, h // 2 + 6), (h - 12, h)):
        a0, a1 = max(r0 - 1, 0), min(r1 + 1, h)
        ref = F.leaky_relu(F.conv2d(x[:, :, a0:a1].double(), wt.double(), b.double(), 1, 1), 0.1)[:, :, r0 - a0:r0 - a0 + (r1 - r0)]
        if ps:
            ref = F.pixel_shuffle(ref, 2)
        assert relerr(y5[:, :, s * r0:s * r1], ref) < 4e-6


def test_conv3x3_winograd_f4x4_split_domain(monkeypatch):
    """The split's domain on the F(4x4) kernel: per-channel scales over ten decades on the input channels (the inverse on the
    weights) and on the output channels (error per channel), inputs scaled by 2^+-100 (V = B^T d B is up to 100 |d|: still far
    from the exponent range's ends), and one inf / NaN input element: the outputs are non-finite on the 4x4 output tiles whose
    6x6 patches hold the element (a SUPERSET of its 3x3 support -- the transforms mix a whole tile; arch_util.py:48-52 has no
    non-finite handling to preserve) and match elsewhere."""
    x, wt, b = _split_case()
    c = x.shape[1]
    s = torch.logspace(-6, 4, c, dtype=torch.float64)
    xa, wa = x * s.view(1, c, 1, 1), wt / s.view(1, c, 1, 1)
    y, geo = _f4_conv(xa, None, wa, b, None, 0, 0, monkeypatch)
    assert geo[3] == 5 and relerr(y, F.conv2d(xa, wa, b, 1, 1)) < 4e-6
    wb = wt * s.view(c, 1, 1, 1)
    y, _ = _f4_conv(x, None, wb, None, None, 0, 0, monkeypatch)
    ref = F.conv2d(x, wb, None, 1, 1)
    per_channel = (y.double() - ref).flatten(2).norm(dim=2).norm(dim=0) / ref.flatten(2).norm(dim=2).norm(dim=0)
    assert float(per_channel.max()) < 4e-6, per_channel
    for e in (100, -100):
        xt = x * 2.0 ** e
        y, _ = _f4_conv(xt, None, wt, None, None, 0, 0, monkeypatch)
        assert bool(torch.isfinite(y).all())
        assert relerr(y.double() * 2.0 ** -e, F.conv2d(xt, wt, None, 1, 1) * 2.0 ** -e) < 4e-6
    for val in (float("inf"), float("nan")):
        xi = x.clone()
        xi[1, 7, 21, 34] = val
        ref = F.conv2d(xi, wt, b, 1, 1)
        y, _ = _f4_conv(xi, None, wt, b, None, 0, 0, monkeypatch)
        bad_ref, bad = ~torch.isfinite(ref), ~torch.isfinite(y)
        assert bool((bad | ~bad_ref).all())                       # every output that depends on the element is non-finite
        tiles = torch.zeros_like(bad)
        tiles[1, :, 20:24, 32:36] = True                          # (21, 34) lies in the interior of ONE 4x4 tile's patch only ...
        tiles[1, :, 20:24, 28:32] = True; tiles[1, :, 20:24, 36:40] = True   # ... column 34 = 32 + 2: not on a tile edge; row 21
        tiles[1, :, 16:20, 28:40] = True; tiles[1, :, 24:28, 28:40] = True   # = 20 + 1: the halo rings of the neighbours reach it
        assert bool((~bad | tiles).all())                         # ... and nothing outside the tiles whose patches hold it
        ok = ~tiles
        assert relerr(torch.where(ok, y, torch.zeros_like(y)), torch.where(ok, ref, torch.zeros_like(ref))) < 4e-6


# ---- domain of exactness of the 3-way bf16 split: x = hi + mid + lo is exact while mid and lo, 2^-8 and 2^-16 of
# x, stay NORMAL bf16 numbers (|x| >= 2^-110 or x == 0) and hi does not round to infinity (|x| < 2^127); the kernels below are
# held to the fp32 bar inside it and to what was measured outside it (tools/split_edge_probe.py, profiles/r05_split_edges.txt).
def _wino_split_conv(x, wt, b, monkeypatch, wino=True):
    import ctypes
    from dynavsr_amd import _lib as L
    monkeypatch.setenv("DVSR_CONV_WINO", "2" if wino else "0")
    monkeypatch.setenv("DVSR_CONV_WINO3", "1")
    monkeypatch.setenv("DVSR_CONV_WINO5", "0")
    n, c, h, w = x.shape
    cout = wt.shape[0]
    dx, dw, db_ = dev(x), dev(wt), dev(b)
    y = torch.empty(n, cout, h, w, device="cuda")
    d = L.Conv2dDesc(L.ptr(dx), None, L.ptr(dw), L.ptr(db_), None, L.ptr(y), n, c, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
    assert list(geo)[3] == (4 if wino else 1), list(geo)
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "dvsr_conv2d_forward_packed")
    return y.cpu()


def _split_case():
    n, c, cout, h, w = 4, 64, 64, 96, 128   # (large enough for the cost model's Winograd branch: the K-split kernel takes small grids)
    return (rnd(n, c, h, w, seed=1), rnd(cout, c, 3, 3, seed=2, scale=1 / np.sqrt(c * 9)), rnd(cout, seed=3, scale=0.1))


def test_split_winograd_per_channel_scales(monkeypatch):
    """Per-channel scales over ten decades (1e-6 .. 1e+4) on the INPUT channels (the inverse on the weights, outputs O(1)) and
    on the OUTPUT channels of the weights (error per channel): the split is per value, so no channel borrows precision from
    another -- the fp32 bar holds on every channel."""
    x, wt, b = _split_case()
    c = x.shape[1]
    s = torch.logspace(-6, 4, c, dtype=torch.float64)
    xa, wa = x * s.view(1, c, 1, 1), wt / s.view(1, c, 1, 1)
    assert relerr(_wino_split_conv(xa, wa, b, monkeypatch), F.conv2d(xa, wa, b, 1, 1)) < 2e-6
    wb = wt * s.view(c, 1, 1, 1)
    y, ref = _wino_split_conv(x, wb, torch.zeros(c, dtype=torch.float64), monkeypatch).double(), F.conv2d(x, wb, None, 1, 1)
    per_channel = (y - ref).flatten(2).norm(dim=2).norm(dim=0) / ref.flatten(2).norm(dim=2).norm(dim=0)
    assert float(per_channel.max()) < 2e-6, per_channel


@pytest.mark.parametrize("e,bar", [(100, 2e-6), (-100, 2e-6), (-108, 2e-6), (-116, 1e-5), (-120, 1e-4), (-124, 2e-3)])
def test_split_winograd_magnitude_range(e, bar, monkeypatch):
    """Inputs scaled by 2^e.  Inside the domain of exactness (|V| >= 2^-110: e >= -108 with these O(1) data) the fp32 bar; below
    it the mid / lo pieces turn into bf16 subnormals and the error grows gracefully -- measured 1.9e-6 at 2^-116, 3.0e-5 at
    2^-120, 4.7e-4 at 2^-124, against 4.3e-7 for the fp32 kernels, which are exact down to fp32's own subnormals: the stated
    difference of the split kernels (the absolute error stays below 2^-133 x the reduction length)."""
    x, wt, b = _split_case()
    xt = x * 2.0 ** e
    ref = F.conv2d(xt, wt, None, 1, 1)
    got = _wino_split_conv(xt, wt, torch.zeros(wt.shape[0], dtype=torch.float64), monkeypatch)
    assert bool(torch.isfinite(got).all())
    up = 2.0 ** -e   # (compare at O(1): conftest.relerr guards its denominator with 1e-30)
    assert relerr(got.double() * up, ref * up) < bar, relerr(got.double() * up, ref * up)
    if e <= -116:   # (the fp32 kernel on the same data: the reference behaviour the difference is measured against)
        got32 = _wino_split_conv(xt, wt, torch.zeros(wt.shape[0], dtype=torch.float64), monkeypatch, wino=False)
        assert relerr(got32.double() * up, ref * up) < 2e-6


@pytest.mark.parametrize("val", [float("inf"), float("nan")])
def test_split_winograd_non_finite_input(val, monkeypatch):
    """One inf / one NaN input element.  The outputs are non-finite EXACTLY where the fp64 convolution's are (the 3 x 3 support
    of the element in every output channel: the Winograd transforms of a tile only mix values an output depends on) and
    untouched elsewhere.  The VALUE differs for inf: the split's residual inf - inf makes it NaN where the fp32 kernels return
    +-inf (arch_util.py:48-52 has no non-finite handling to preserve; stated in DESIGN 3.3)."""
    x, wt, b = _split_case()
    x = x.clone()
    x[1, 7, 21, 34] = val
    ref = F.conv2d(x, wt, b, 1, 1)
    got = _wino_split_conv(x, wt, b, monkeypatch)
    bad_ref, bad = ~torch.isfinite(ref), ~torch.isfinite(got)
    assert int(bad_ref.sum()) == 9 * wt.shape[0] and torch.equal(bad, bad_ref)
    assert bool(torch.isnan(got[bad]).all())
    assert relerr(torch.where(bad, torch.zeros_like(got), got), torch.where(bad, torch.zeros_like(ref), ref)) < 2e-6
    ref32 = _wino_split_conv(x, wt, b, monkeypatch, wino=False)
    assert torch.equal(~torch.isfinite(ref32), bad_ref)
    assert bool((torch.isnan(ref32[bad_ref]) if val != val else torch.isinf(ref32[bad_ref])).all())


def _wgrad_split3(x, gy):
    from dynavsr_amd import _lib as L
    n, c, h, w = x.shape
    cout = gy.shape[1]
    xg, gg = dev(x), dev(gy)
    gw, gb = torch.empty(cout, c, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
    d = L.Conv2dDesc(L.ptr(xg), None, None, None, None, None, n, c, 0, h, w, cout, 3, 1, 1, 0, 0, 1, 0, 0)
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_backward_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
    L.check(L.lib().dvsr_conv2d_wgrad_split3(d, L.ptr(gg), L.ptr(gw), L.ptr(gb), ws.data_ptr(), ws.numel(), L.stream()),
            "dvsr_conv2d_wgrad_split3")
    return gw.cpu()


def _wgrad_ref(x, gy):
    wz = torch.zeros(gy.shape[1], x.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    (g,) = torch.autograd.grad(F.conv2d(x, wz, padding=1), wz, gy)
    return g


def test_split_wgrad_scales_range_and_non_finite():
    """conv2d_wgrad_split3_kernel on the same edge cases: per-input-channel scales over ten decades (error per channel), tiny
    inputs (2^-100, 2^-112: fp32 bar; 2^-120: 3.5e-5 measured, bf16-subnormal pieces), one inf / NaN input element (non-finite on
    exactly the 9 x Cout gradient entries of its channel, as NaN)."""
    x, gy = rnd(2, 64, 48, 64, seed=1), rnd(2, 64, 48, 64, seed=5)
    c = x.shape[1]
    s = torch.logspace(-6, 4, c, dtype=torch.float64)
    xa = x * s.view(1, c, 1, 1)
    got, ref = _wgrad_split3(xa, gy).double(), _wgrad_ref(xa, gy)
    per_channel = (got - ref).flatten(2).norm(dim=2).norm(dim=0) / ref.flatten(2).norm(dim=2).norm(dim=0)
    assert float(per_channel.max()) < 2e-6, per_channel
    for e, bar in ((-100, 2e-6), (-112, 2e-6), (-120, 2e-4)):
        up = 2.0 ** -e   # (compare at O(1): conftest.relerr guards its denominator with 1e-30)
        assert relerr(_wgrad_split3(x * 2.0 ** e, gy).double() * up, _wgrad_ref(x * 2.0 ** e, gy) * up) < bar, e
    for val in (float("inf"), float("nan")):
        xi = x.clone()
        xi[1, 7, 20, 33] = val
        got, ref = _wgrad_split3(xi, gy), _wgrad_ref(xi, gy)
        bad_ref, bad = ~torch.isfinite(ref), ~torch.isfinite(got)
        assert int(bad_ref.sum()) == 9 * gy.shape[1] and torch.equal(bad, bad_ref)
        assert bool(torch.isnan(got[bad]).all())
        assert relerr(torch.where(bad, torch.zeros_like(got), got), torch.where(bad, torch.zeros_like(ref), ref)) < 2e-6


@pytest.mark.parametrize("pipe", ["bf16x3", "fp32"])
@pytest.mark.parametrize("name,cout,h,w,ps", [("upconv2", 256, 360, 640, 2), ("HRconv", 64, 720, 1280, 0)])
def test_conv3x3_winograd_largest_geometries(name, cout, h, w, ps, pipe, monkeypatch):
    """The two largest launches of the headline forward at op level -- upconv2 (64 -> 256 + PixelShuffle(2) at 360x640) and
    HRconv (64 -> 64 at 720x1280), EDVR_arch.py:304-306 -- on both Winograd kernels: the whole output against the direct
    kernel (the same sums in another order: <= 2e-6), and three bands of rows (top edge, middle, bottom edge) against fp64
    torch on the corresponding crops (<= 2e-6): a full fp64 evaluation of these shapes is 34 / 68 GFLOP of CPU work."""
    import ctypes
    from dynavsr_amd import _lib as L
    n, c0 = 1, 64
    x = rnd(n, c0, h, w, seed=11)
    wt = rnd(cout, c0, 3, 3, seed=12, scale=1 / np.sqrt(c0 * 9))
    b = rnd(cout, seed=13, scale=0.1)
    dx, dw, db_ = dev(x), dev(wt), dev(b)
    outs = {}
    monkeypatch.setenv("DVSR_CONV_WINO5", "0")
    for mode in ("wino", "direct"):
        monkeypatch.setenv("DVSR_CONV_WINO", "2" if mode == "wino" else "0")
        monkeypatch.setenv("DVSR_CONV_WINO3", "1" if pipe == "bf16x3" else "0")
        y = torch.empty((n, cout // 4, 2 * h, 2 * w) if ps else (n, cout, h, w), device="cuda")
        d = L.Conv2dDesc(L.ptr(dx), None, L.ptr(dw), L.ptr(db_), None, L.ptr(y), n, c0, 0, h, w, cout, 3, 1, 1, 1, ps, 1, 0, 0)
        geo = (ctypes.c_int * 4)()
        L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
        assert list(geo)[3] == ((4 if pipe == "bf16x3" else 3) if mode == "wino" else 1), (mode, list(geo))
        ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)), 16), dtype=torch.uint8, device="cuda")
        L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "dvsr_conv2d_forward_packed")
        outs[mode] = y.cpu()
    assert relerr(outs["wino"], outs["direct"]) < 2e-6
    s = 2 if ps else 1
    for r0, r1 in ((0, 12), (h // 2 - 6, h // 2 + 6), (h - 12, h)):
        a0, a1 = max(r0 - 1, 0), min(r1 + 1, h)                       # input rows incl. the one-row halo inside the image
        ref = F.leaky_relu(F.conv2d(x[:, :, a0:a1].double(), wt.double(), b.double(), 1, 1), 0.1)[:, :, r0 - a0:r0 - a0 + (r1 - r0)]
        if ps:
            ref = F.pixel_shuffle(ref, 2)
        assert relerr(outs["wino"][:, :, s * r0:s * r1], ref) < 2e-6, (name, r0)


def test_conv3x3_dma_halo_not_for_unaligned():
    """W % 4 != 0 or channel counts % 8 != 0 keep the register-staged kernel."""
    import ctypes
    from dynavsr_amd import _lib as L
    x = dev(rnd(4, 64, 98, 162))
    geo = (ctypes.c_int * 4)()
    for cin, wd in ((64, 162), (60, 160)):
        d = L.Conv2dDesc(L.ptr(x), None, None, None, None, None, 4, cin, 0, 98, wd, 64, 3, 1, 1, 0, 0, 1, 0, 0)
        L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
        assert list(geo)[3] == 0
