"""CPU: pins the C oracle of modulated deformable conv (oracle/dcn_oracle.c).

The reference cannot execute this op without CUDA, so the pins are: golden vectors from an
independent torch-gather formulation (fp64), torch.autograd.gradcheck, and the
zero-offset == conv2d identities.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, relerr
from oracle import dcn


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_oracle_matches_golden(tag, dtype, tol):
    g = load_golden("dcn_" + tag)
    dg = int(g["dg"])
    x, off, m, w, b, go = (torch.from_numpy(g[k]).to(dtype) for k in
                           ("x", "offset", "mask", "weight", "bias", "gout"))
    out = dcn.forward(x, off, m, w, b, 1, 1, 1, 1, dg)
    assert relerr(out, g["out"]) < tol
    gx, goff, gm, gw, gb = dcn.backward(x, off, m, w, True, go, 1, 1, 1, 1, dg)
    for got, key in ((gx, "gx"), (goff, "goffset"), (gm, "gmask"), (gw, "gweight"), (gb, "gbias")):
        assert relerr(got, g[key]) < tol, key


def test_gradcheck_fp64():
    torch.manual_seed(0)
    n, c, h, w, co, dg = 1, 4, 5, 6, 3, 2
    x = torch.randn(n, c, h, w, dtype=torch.float64, requires_grad=True)
    # keep sampling points away from integer coordinates (bilinear kinks)
    off = (torch.rand(n, dg * 18, h, w, dtype=torch.float64) * 0.6 + 0.2).requires_grad_()
    m = torch.rand(n, dg * 9, h, w, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(co, c, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.randn(co, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(
        lambda *a: dcn.modulated_deform_conv(*a, 1, 1, 1, 1, dg), (x, off, m, wt, b), eps=1e-6,
        atol=1e-6)


@pytest.mark.parametrize("stride,pad,dil,groups", [(1, 1, 1, 1), (2, 1, 1, 1), (1, 2, 2, 2)])
def test_zero_offset_is_conv2d(stride, pad, dil, groups):
    torch.manual_seed(1)
    x = torch.randn(2, 8, 9, 7, dtype=torch.float64)
    wt = torch.randn(4, 8 // groups, 3, 3, dtype=torch.float64)
    b = torch.randn(4, dtype=torch.float64)
    ref = F.conv2d(x, wt, b, stride, pad, dil, groups)
    ho, wo = ref.shape[2:]
    out = dcn.forward(x, torch.zeros(2, 2 * 18, ho, wo, dtype=torch.float64),
                      torch.ones(2, 2 * 9, ho, wo, dtype=torch.float64), wt, b, stride, pad, dil,
                      groups, 2)
    assert relerr(out, ref) < 1e-13


def test_fresh_pack_is_half_conv():
    """A freshly constructed ModulatedDeformConvPack has zero offsets and mask = sigmoid(0) = 0.5
    (deform_conv.py:270-272) -> out = 0.5 * conv2d(x, W) + b."""
    torch.manual_seed(2)
    x = torch.randn(1, 8, 6, 6, dtype=torch.float64)
    wt = torch.randn(8, 8, 3, 3, dtype=torch.float64)
    b = torch.randn(8, dtype=torch.float64)
    out = dcn.forward(x, torch.zeros(1, 36, 6, 6, dtype=torch.float64),
                      torch.full((1, 18, 6, 6), 0.5, dtype=torch.float64), wt, b, 1, 1, 1, 1, 2)
    assert relerr(out, 0.5 * F.conv2d(x, wt, None, 1, 1) + b.view(1, -1, 1, 1)) < 1e-13


def test_gather_and_c_agree_random_geometry():
    r = np.random.RandomState(5)
    for _ in range(3):
        n, dg = 1, int(r.choice([1, 2, 4]))
        c = dg * int(r.randint(1, 4))
        h, w = int(r.randint(3, 9)), int(r.randint(3, 9))
        co = int(r.randint(1, 6))
        x = torch.from_numpy(r.standard_normal((n, c, h, w)))
        off = torch.from_numpy(r.standard_normal((n, dg * 18, h, w)) * 3)
        m = torch.from_numpy(r.random_sample((n, dg * 9, h, w)))
        wt = torch.from_numpy(r.standard_normal((co, c, 3, 3)))
        a = dcn.forward(x, off, m, wt, None, 1, 1, 1, 1, dg)
        b = dcn.gather_reference(x, off, m, wt, None, 1, 1, 1, 1, dg)
        assert relerr(a, b) < 1e-12
