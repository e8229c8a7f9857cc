"""CPU: the functional oracle (oracle/edvr.py, mfdn.py, inner.py) against golden vectors produced
by the imported reference modules/wrappers (oracle/gen_golden.py)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import load_golden, relerr
from dynavsr_amd import synth
from oracle import edvr, inner, mfdn


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_pcd_align_golden():
    g = load_golden("pcd_align")
    P = synth.edvr_state_dict(int(g["seed"]))
    with torch.no_grad():
        y = edvr.pcd_align(P, [_t(g["nbr0"]), _t(g["nbr1"]), _t(g["nbr2"])],
                           [_t(g["ref0"]), _t(g["ref1"]), _t(g["ref2"])], 8)
    assert relerr(y, g["out"]) < 1e-6


def test_tsa_fusion_golden():
    g = load_golden("tsa_fusion")
    P = synth.edvr_state_dict(int(g["seed"]))
    with torch.no_grad():
        y = edvr.tsa_fusion(P, _t(g["aligned"]), 2)
    assert relerr(y, g["out"]) < 1e-6


@pytest.mark.parametrize("tag", ["16x16", "32x48"])
def test_edvr_forward_backward_golden(tag):
    g = load_golden("edvr_" + tag)
    h, w = int(g["h"]), int(g["w"])
    P = OrderedDict((k, v.requires_grad_(True)) for k, v in synth.edvr_state_dict(int(g["wseed"])).items())
    x = synth.clip(int(g["xseed"]), 1, 5, h, w)
    tgt = synth.clip(int(g["tseed"]), 1, 1, 4 * h, 4 * w)[:, 0]
    y = edvr.edvr_forward(P, x)
    loss = edvr.charbonnier(y, tgt)
    assert relerr(y, g["out"]) < 1e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    grads = torch.autograd.grad(loss, list(P.values()))
    norms = np.array([float(x.norm()) for x in grads])
    assert np.allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-9)
    by_name = dict(zip(P.keys(), grads))
    for key in g:
        if key.startswith("grad__"):
            name = key[len("grad__"):].replace("__", ".")
            assert relerr(by_name[name], g[key]) < 2e-4, name


def test_edvr_x2_forward_backward_golden():
    """EDVR-M x2 (the shipped x2 YAMLs): one pixel-shuffle stage less, x2 bilinear base."""
    g = load_golden("edvr_x2_24x32")
    h, w = int(g["h"]), int(g["w"])
    cfg = dict(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=2)
    P = OrderedDict((k, v.requires_grad_(True)) for k, v in synth.edvr_state_dict(int(g["wseed"]), **cfg).items())
    assert "upconv1.weight" not in P and len(P) == 142
    x = synth.clip(int(g["xseed"]), 1, 5, h, w)
    tgt = synth.clip(int(g["tseed"]), 1, 1, 2 * h, 2 * w)[:, 0]
    y = edvr.edvr_forward(P, x, scale=2)
    loss = edvr.charbonnier(y, tgt)
    assert tuple(y.shape) == (1, 3, 2 * h, 2 * w) and relerr(y, g["out"]) < 1e-5
    grads = torch.autograd.grad(loss, list(P.values()))
    assert np.allclose(np.array([float(v.norm()) for v in grads]), g["grad_norms"], rtol=2e-4, atol=1e-9)


def test_mfdn_golden():
    g = load_golden("mfdn_32x32")
    M = OrderedDict((k, v.requires_grad_(True)) for k, v in synth.mfdn_state_dict(int(g["wseed"])).items())
    lq = synth.clip(int(g["xseed"]), 1, 5, 32, 32)
    y = mfdn.mfdn_forward(M, lq)
    assert relerr(y, g["out"]) < 1e-6
    go = _t(np.random.RandomState(int(g["goseed"])).standard_normal(tuple(y.shape)).astype(np.float32))
    grads = torch.autograd.grad(y, list(M.values()), go)
    assert np.allclose([float(x.norm()) for x in grads], g["grad_norms"], rtol=2e-4)


def _check_all_grads(g, names, grads, tol=2e-4):
    for name, gr in zip(names, grads):
        assert relerr(gr, g["grad__" + name.replace(".", "__")]) < tol, name


def test_mfdn_x2_golden():
    """scale 2 (3x3 conv3) and T = 3: every parameter gradient of the reference module."""
    g = load_golden("mfdn_x2_24x40")
    M = OrderedDict((k, v.requires_grad_(True))
                    for k, v in synth.mfdn_state_dict(int(g["wseed"]), nf=int(g["nf"]), scale=2).items())
    y = mfdn.mfdn_forward(M, synth.clip(int(g["xseed"]), 1, 3, 24, 40), scale=2)
    assert relerr(y, g["out"]) < 1e-6
    go = _t(np.random.RandomState(int(g["goseed"])).standard_normal(tuple(y.shape)).astype(np.float32))
    _check_all_grads(g, list(M), torch.autograd.grad(y, list(M.values()), go))


def test_sfdn_golden():
    g = load_golden("sfdn_20x28")
    S = OrderedDict((k, v.requires_grad_(True)) for k, v in synth.sfdn_state_dict(int(g["wseed"]), nf=int(g["nf"])).items())
    y = mfdn.sfdn_forward(S, synth.clip(int(g["xseed"]), 2, 1, 20, 28)[:, 0])
    assert relerr(y, g["out"]) < 1e-6
    go = _t(np.random.RandomState(int(g["goseed"])).standard_normal(tuple(y.shape)).astype(np.float32))
    _check_all_grads(g, list(S), torch.autograd.grad(y, list(S.values()), go))


@pytest.mark.parametrize("optimizer", ["SGD", "Adam"])
def test_inner_step_golden(optimizer):
    """BASELINE.json configs[0]: EDVR-M x4, LR 64x64 -> SLR 16x16, one inner step."""
    g = load_golden("inner_step_" + optimizer.lower())
    PG, PE, PEF = synth.edvr_state_dict(0), synth.mfdn_state_dict(0), synth.mfdn_state_dict(1)
    lqs = synth.clip(1, 1, 5, 64, 64)
    losses, PGa, PEa, sr = inner.inner_adapt(PG, PE, PEF, lqs, 1, optimizer, 1e-5, (0.9, 0.99))
    assert abs(losses[0] - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    ref_sr = g["sr"]
    if optimizer == "Adam":
        sr = sr[..., 96:160, 96:160]
    assert relerr(sr, ref_sr) < 1e-5
    tol = 1e-3 if optimizer == "SGD" else 5e-2
    for key in g:
        if key.startswith("dG__") or key.startswith("dE__"):
            name = key[4:].replace("__", ".")
            src, new = (PG, PGa) if key.startswith("dG__") else (PE, PEa)
            delta = new[name].detach().double() - src[name].double()
            assert relerr(delta, g[key]) < tol, name


def test_psnr_golden():
    g = load_golden("psnr")
    gt = synth.clip(int(g["gtseed"]), 1, 1, 256, 256)[0, 0]
    assert abs(inner.psnr_uint8(g["img"], inner.tensor2img_rgb(gt)) - float(g["psnr"])) < 1e-12


def test_winograd_f2x2_3x3_restatement_equals_correlation():
    """oracle/winograd.py (the transforms conv2d_wino.hip uses) against a plain 3x3 / stride-1 / pad-1 correlation, in fp64
    (identity up to round-off) and with the transformed-domain products rounded to fp32 (the kernel's arithmetic)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import winograd
    rs = np.random.RandomState(3)
    x = rs.standard_normal((2, 5, 8, 12))
    w = rs.standard_normal((7, 5, 3, 3)) / np.sqrt(45.0)
    b = rs.standard_normal(7)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, 1).numpy()
    got = winograd.conv3x3_winograd(x, w, b)
    assert np.abs(got - ref).max() < 1e-12
    got32 = winograd.conv3x3_winograd(x, w, b, dtype=np.float32)
    assert np.linalg.norm(got32 - ref) / np.linalg.norm(ref) < 2e-6
    # the weight transform alone: U[0][0] = g[0][0], U[3][3] = g[2][2], U[1][1] = sum(g) / 4
    U = winograd.weight_transform(w)
    assert np.allclose(U[0, 0], w[:, :, 0, 0]) and np.allclose(U[3, 3], w[:, :, 2, 2])
    assert np.allclose(U[1, 1], w.sum((2, 3)) / 4)


def test_winograd_f4x4_3x3_restatement_equals_correlation():
    """oracle/winograd.py's F(4x4, 3x3) (the transforms conv2d_wino5.hip uses) against a plain 3x3 / stride-1 / pad-1
    correlation: in fp64 the identity up to round-off; in the kernel's arithmetic (V in fp32, three bf16 pieces per operand,
    six of nine partial products, fp32 sums) the bar tests/test_gpu_ops.py holds the kernel to (4e-6; 1e-6 .. 2e-6 observed on
    64-channel layers: six times F(2x2), the price of the transform's larger constants)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import winograd
    rs = np.random.RandomState(5)
    x = rs.standard_normal((2, 16, 8, 12))
    w = rs.standard_normal((7, 16, 3, 3)) / 12.0
    b = rs.standard_normal(7)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, 1).numpy()
    got = winograd.conv3x3_winograd_f4(x, w, b)
    assert np.abs(got - ref).max() < 1e-11
    x32, w32 = x.astype(np.float32), w.astype(np.float32)
    ref32 = F.conv2d(torch.from_numpy(x32).double(), torch.from_numpy(w32).double(), torch.from_numpy(b), 1, 1).numpy()
    got3 = winograd.conv3x3_winograd_f4(x32, w32, b, split=True)
    assert np.linalg.norm(got3 - ref32) / np.linalg.norm(ref32) < 4e-6
    # the split is exact: hi + mid + lo == x for every fp32 value in the domain (|x| >= 2^-110)
    v = (rs.standard_normal(4096) * 10.0 ** rs.uniform(-20, 20, 4096)).astype(np.float32)
    h, m, lo = winograd.split3_bf16(v)
    assert np.array_equal((h.astype(np.float64) + m + lo).astype(np.float32), v)
    assert np.array_equal(h.view(np.uint32) & 0xFFFF, np.zeros(4096, np.uint32))
