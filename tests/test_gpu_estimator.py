"""GPU parity of the native estimator tape (dvsr_estimator_forward / _backward: MFDN x4, MFDN x2, SFDN)
against the golden vectors produced by the imported reference modules and against the CPU oracle
(oracle/mfdn.py).  fp32 throughout; asserted: outputs rel-L2 <= 1e-5 (north_star allows 1e-3), every
parameter gradient rel-L2 <= 2e-4 (summation order is the only difference: there are no kinks besides
LeakyReLU, whose two branches are both smooth in the weights)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import load_golden, relerr
from dynavsr_amd import synth

pytestmark = pytest.mark.gpu


def _go(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(tuple(shape)).astype(np.float32))


def _mfdn(sd, **cfg):
    from dynavsr_amd.models.archs.LRimg_estimator import DirectKernelEstimatorVideo
    net = DirectKernelEstimatorVideo(**cfg)
    net.load_state_dict(sd, strict=True)
    return net.cuda()


def test_mfdn_x4_golden_and_oracle():
    from oracle import mfdn as omfdn
    g = load_golden("mfdn_32x32")
    sd = synth.mfdn_state_dict(int(g["wseed"]))
    net = _mfdn(sd, nf=64, in_nc=3, scale=4)
    lq = synth.clip(int(g["xseed"]), 1, 5, 32, 32)
    y = net(lq.transpose(1, 2).contiguous().cuda()).transpose(1, 2)
    assert y.shape == g["out"].shape
    assert relerr(y, g["out"]) < 1e-5
    go = _go(int(g["goseed"]), y.shape)
    y.backward(go.cuda())
    assert np.allclose([float(p.grad.norm()) for p in net.parameters()], g["grad_norms"], rtol=2e-4)
    assert relerr(net.conv6.weight.grad, g["grad__conv6__weight"]) < 2e-4
    assert relerr(net.conv0.bias.grad, g["grad__conv0__bias"]) < 2e-4
    # every parameter against the (golden-pinned) oracle
    MO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    og = torch.autograd.grad(omfdn.mfdn_forward(MO, lq), list(MO.values()), go)
    for (k, p), ref in zip(net.named_parameters(), og):
        assert relerr(p.grad, ref) < 2e-4, k


def test_mfdn_x2_golden():
    g = load_golden("mfdn_x2_24x40")
    nf = int(g["nf"])
    net = _mfdn(synth.mfdn_state_dict(int(g["wseed"]), nf=nf, scale=2), nf=nf, in_nc=3, scale=2)
    lq = synth.clip(int(g["xseed"]), 1, 3, 24, 40)
    y = net(lq.transpose(1, 2).contiguous().cuda()).transpose(1, 2)
    assert relerr(y, g["out"]) < 1e-5
    y.backward(_go(int(g["goseed"]), y.shape).cuda())
    for k, p in net.named_parameters():
        assert relerr(p.grad, g["grad__" + k.replace(".", "__")]) < 2e-4, k


def test_sfdn_golden():
    from dynavsr_amd.models.archs.LRimg_estimator import DirectKernelEstimator_CMS
    g = load_golden("sfdn_20x28")
    nf = int(g["nf"])
    net = DirectKernelEstimator_CMS(nf=nf)
    net.load_state_dict(synth.sfdn_state_dict(int(g["wseed"]), nf=nf), strict=True)
    net = net.cuda()
    y = net(synth.clip(int(g["xseed"]), 2, 1, 20, 28)[:, 0].contiguous().cuda())
    assert relerr(y, g["out"]) < 1e-5
    y.backward(_go(int(g["goseed"]), y.shape).cuda())
    for k, p in net.named_parameters():
        assert relerr(p.grad, g["grad__" + k.replace(".", "__")]) < 2e-4, k


@pytest.mark.parametrize("b,t,h,w", [(2, 5, 16, 48), (1, 7, 36, 20), (1, 5, 12, 12), (1, 5, 176, 320)])   # (12x12: W = 3 after the two stride-2 convs -- column 1 feeds BOTH ring columns of the fused pad store)
def test_mfdn_x4_vs_oracle_shapes(b, t, h, w):
    """Batch > 1, 7 frames, ragged tile edges (36x20 -> 9x5) and the inner-step size of BASELINE configs[1]."""
    from oracle import mfdn as omfdn
    sd = synth.mfdn_state_dict(2)
    net = _mfdn(sd, nf=64, in_nc=3, scale=4)
    lq = synth.clip(21, b, t, h, w, smooth=False)
    y = net(lq.transpose(1, 2).contiguous().cuda()).transpose(1, 2)
    MO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    yo = omfdn.mfdn_forward(MO, lq)
    assert relerr(y, yo) < 1e-5
    assert float((y.cpu() - yo.detach()).abs().max()) < 1e-4
    go = _go(22, y.shape)
    y.backward(go.cuda())
    og = torch.autograd.grad(yo, list(MO.values()), go)
    if h * w <= 4096:
        for (k, p), ref in zip(net.named_parameters(), og):
            assert relerr(p.grad, ref) < 2e-4, k
        return
    # At 5 x 176 x 320 the comparison stops being a rounding question: about one activation in a million
    # lies within 1e-6 of the LeakyReLU kink, and when two fp32 implementations disagree on its sign that ONE
    # element changes an upstream gradient tensor by ~2e-3 relative (traced with tools/estimator_debug.py:
    # every intermediate gradient agrees to 1e-6 except isolated elements whose forward value is ~1e-7).
    # So: a loose bound against the oracle here, the strict bound at the sizes above, and a strict
    # size-independent property: with the activations fixed, backward is linear in grad_out.
    for (k, p), ref in zip(net.named_parameters(), og):
        assert relerr(p.grad, ref) < 5e-3, k
    g1 = [p.grad.clone() for p in net.parameters()]
    go2 = _go(23, y.shape)
    grads = []
    for gg in (go2, go + go2):
        for p in net.parameters():
            p.grad = None
        net(lq.transpose(1, 2).contiguous().cuda()).transpose(1, 2).backward(gg.cuda())
        grads.append([p.grad.clone() for p in net.parameters()])
    for (k, _), a, b_, c in zip(net.named_parameters(), g1, grads[0], grads[1]):
        assert relerr(a + b_, c) < 2e-5, k   # 1.3e-5 observed with the 2x2 data gradient on the bf16 3-way split


@pytest.mark.parametrize("h,w", [(176, 320), (180, 320), (172, 312)])
def test_mfdn_split_2x2_form_matches_fp32_kernels(h, w, monkeypatch):
    """The 2x2 space-to-depth form of the 4x4 stride-2 convolutions on the bf16 3-way operand split (engine.hip
    Builder::conv, DVSR_EST_SPLIT2; taken from a workgroup per CU upwards, hence 2 x 5 x 176 x 320) against the same tape
    with those launches on the fp32 MFMA kernel: output to fp32 round-off, every parameter gradient within the kink-flip
    budget of two fp32 evaluations (see test_mfdn_x4_vs_oracle_shapes).  180 x 320 and 172 x 312 give ragged tiles on both
    levels of the split kernels (90 x 160: rows % 4 = 2; 86 x 156 / 43 x 78: ragged rows and columns, an odd row count)."""
    sd = synth.mfdn_state_dict(5)
    x = synth.clip(31, 2, 5, h, w, smooth=False).transpose(1, 2).contiguous().cuda()
    go = None
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DVSR_EST_SPLIT2", mode)
        net = _mfdn(sd, nf=64, in_nc=3, scale=4)
        y = net(x)
        go = _go(32, y.shape).cuda() if go is None else go
        y.backward(go)
        res[mode] = (y.detach().clone(), OrderedDict((k, p.grad.clone()) for k, p in net.named_parameters()))
    assert not torch.equal(res["0"][0], res["1"][0])          # (the switch did change the kernels)
    assert relerr(res["1"][0], res["0"][0]) < 2e-6
    for k in res["0"][1]:
        assert relerr(res["1"][1][k], res["0"][1][k]) < 5e-3, k


def test_mfdn_split_convs_per_channel_scales_and_non_finite(monkeypatch):
    """The estimators' split convolutions (3x3 over padded inputs and the 2x2 space-to-depth form, both on the exact 3-way bf16
    split: DVSR_EST_SPLIT / DVSR_EST_SPLIT2) with per-channel activation scales over six decades: the chain
    conv_i -> LeakyReLU -> conv_{i+1} is re-parametrised channel by channel (output channel c of conv_i times s_c, input
    channel c of conv_{i+1} divided by it: the same function, LeakyReLU being positively homogeneous), so the tensors the
    split kernels read span 1e-3 .. 1e+3 per channel.  Split tape == fp32 tape on the same weights to the fp32 bar, and both
    == the un-scaled network to re-association round-off.  A NaN (or inf) in frame 3 of a clip poisons its whole plane through
    the image mean (LRimg_estimator.py:92-93) and from there frames 1..4 of that clip through the two temporal convolutions
    (conv0, conv5: three frames each): both tapes return NaN exactly there, frame 0 and the other clip stay finite."""
    sd = synth.mfdn_state_dict(5)
    sd2 = OrderedDict((k, v.clone()) for k, v in sd.items())
    for i, (a, b_) in enumerate((("conv1", "conv2"), ("conv2", "conv3"), ("conv3", "conv4"))):
        c = sd2[a + ".weight"].shape[0]
        sc = torch.logspace(-3, 3, c)[torch.randperm(c, generator=torch.Generator().manual_seed(40 + i))]
        sd2[a + ".weight"] *= sc.view(c, 1, 1, 1)
        sd2[a + ".bias"] *= sc
        sd2[b_ + ".weight"] /= sc.view(1, c, 1, 1)
    x = synth.clip(33, 2, 5, 176, 320, smooth=False).transpose(1, 2).contiguous().cuda()
    out = {}
    for name, state, split in (("plain32", sd, "0"), ("scaled32", sd2, "0"), ("scaled_split", sd2, "1")):
        monkeypatch.setenv("DVSR_EST_SPLIT", split)
        monkeypatch.setenv("DVSR_EST_SPLIT2", split)
        with torch.no_grad():
            out[name] = _mfdn(state, nf=64, in_nc=3, scale=4)(x).clone()
    assert not torch.equal(out["scaled32"], out["scaled_split"])     # (the switch did change the kernels)
    assert relerr(out["scaled_split"], out["scaled32"]) < 2e-6
    assert relerr(out["scaled_split"], out["plain32"]) < 2e-5
    for val in (float("nan"), float("inf")):
        xb = x.clone()
        xb[1, 2, 3, 50, 60] = val
        for split in ("0", "1"):
            monkeypatch.setenv("DVSR_EST_SPLIT", split)
            monkeypatch.setenv("DVSR_EST_SPLIT2", split)
            with torch.no_grad():
                y = _mfdn(sd, nf=64, in_nc=3, scale=4)(xb)
            assert bool(torch.isfinite(y[0]).all()) and bool(torch.isfinite(y[1][:, 0]).all()), (val, split)
            assert bool(torch.isnan(y[1][:, 1:]).all()), (val, split)


def test_estimator_rejects_input_grad_and_bad_shapes():
    net = _mfdn(synth.mfdn_state_dict(0), nf=64, in_nc=3, scale=4)
    x = synth.clip(1, 1, 5, 16, 16).transpose(1, 2).contiguous().cuda()
    with pytest.raises(RuntimeError, match="data"):
        net(x.clone().requires_grad_(True))
    with pytest.raises(RuntimeError, match="multiples of the scale"):
        net(synth.clip(1, 1, 5, 18, 16).transpose(1, 2).contiguous().cuda())


@pytest.mark.parametrize("scale,k,t,h,w", [(4, 3, 5, 32, 48), (2, 2, 3, 24, 40), (4, 2, 5, 176, 320)])   # (+ the bench size)
def test_mfdn_stacked_tape_gives_per_clip_gradients(scale, k, t, h, w):
    """EstimatorStackedFunction (dvsr_estimator_plan_create_grouped): K clips as one batch, per-clip parameter
    gradients in the slices of the stacked parameters == K separate B = 1 passes (incl. the re-laid-out 4x4 stride-2
    weights, whose gradient is mapped back per group)."""
    from dynavsr_amd import engine
    net = _mfdn(synth.mfdn_state_dict(3, scale=scale), nf=64, in_nc=3, scale=scale)
    x = synth.clip(71, k, t, h, w).transpose(1, 2).contiguous().cuda()        # [K,3,T,H,W]
    go = _go(72, (k, 3, t, h // scale, w // scale)).cuda()
    want = []
    for i in range(k):
        yi = net(x[i:i + 1])
        want.append((yi.detach(), torch.autograd.grad(yi, net.ordered_parameters(), go[i:i + 1])))
    stacked = [p.detach().unsqueeze(0).repeat((k,) + (1,) * p.dim()).contiguous().requires_grad_() for p in net.ordered_parameters()]
    cfg = (engine.MFDN, 64, 3, scale, t)
    y = engine.EstimatorStackedFunction.apply(x, cfg, False, *stacked)
    y.backward(go)
    for i in range(k):
        assert relerr(y[i:i + 1], want[i][0]) < 1e-6
        bad = [(j, relerr(s.grad[i], g)) for j, (s, g) in enumerate(zip(stacked, want[i][1])) if relerr(s.grad[i], g) > 1e-4]
        assert not bad, (i, bad)


@pytest.mark.parametrize("scale,k,t,h,w", [(4, 3, 5, 32, 48), (2, 2, 3, 24, 40), (4, 2, 5, 176, 320)])   # (+ the bench size)
def test_mfdn_stacked_tape_per_slice_weights(scale, k, t, h, w):
    """dvsr_estimator_plan_create_ex with weight_sets = K: clip k runs on slice k of the stacked parameters (copies that
    have diverged), output and slice k of every gradient == a B = 1 pass through a network holding slice k."""
    from dynavsr_amd import engine
    net = _mfdn(synth.mfdn_state_dict(3, scale=scale), nf=64, in_nc=3, scale=scale)
    x = synth.clip(73, k, t, h, w).transpose(1, 2).contiguous().cuda()
    go = _go(74, (k, 3, t, h // scale, w // scale)).cuda()
    g = torch.Generator(device="cuda").manual_seed(6)
    stacked = []
    for p in net.ordered_parameters():
        s_ = p.detach().unsqueeze(0).repeat((k,) + (1,) * p.dim())
        stacked.append((s_ * (1.0 + 0.05 * torch.randn(s_.shape, device="cuda", generator=g))).contiguous().requires_grad_())
    y = engine.EstimatorStackedFunction.apply(x, (engine.MFDN, 64, 3, scale, t), True, *stacked)
    y.backward(go)
    for i in range(k):
        with torch.no_grad():
            for p, s_ in zip(net.ordered_parameters(), stacked):
                p.copy_(s_[i])
        yi = net(x[i:i + 1])
        gs = torch.autograd.grad(yi, net.ordered_parameters(), go[i:i + 1])
        assert relerr(y[i:i + 1], yi) < 1e-6
        bad = [(j, relerr(s_.grad[i], g_)) for j, (s_, g_) in enumerate(zip(stacked, gs)) if relerr(s_.grad[i], g_) > 1e-4]
        assert not bad, (i, bad)


@pytest.mark.parametrize("below", ["0", "100000000"], ids=["never_row_split", "always_row_split"])
def test_mfdn_gradients_under_both_weight_gradient_forms(below):
    """The split weight gradient picks its form by launch size (conv2d_wgrad.hip: DVSR_WGRAD_S3_KYS_BELOW): small launches one
    kernel row per workgroup, large ones all taps per workgroup.  The estimator's gradient tests (3x3 and the 2x2
    space-to-depth form, float2 / float4 staging, ragged sizes, per-clip groups, 176x320) with either form forced on every
    launch; the switch is read once per process, so they run in a child."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "vs_oracle_shapes or stacked_tape or golden"],
                       env=dict(os.environ, DVSR_WGRAD_S3_KYS_BELOW=below), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout
