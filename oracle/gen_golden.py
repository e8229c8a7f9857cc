#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on CPU in this container.

TEST INFRASTRUCTURE.  Run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
(/root/reference is read-only; it exists only in the build container, never on the GPU box, so
the vectors are committed).  How the reference is made importable without a GPU:
  1. an empty module is registered as ``models.archs.dcn.deform_conv_cuda`` (the CUDA extension),
  2. ``models.archs.dcn.deform_conv.modulated_deform_conv`` is rebound to the C oracle
     (oracle/dcn.py) -- ModulatedDeformConvPack.forward looks the name up at call time
     (deform_conv.py:289), so every line of EDVR_arch.py still runs as reference code,
  3. ``cv2`` and ``torchvision.utils.make_grid`` are stubbed so utils/util.py imports.
Weights/inputs come from dynavsr_amd.synth (frozen numpy RandomState streams), so fixtures hold
seeds + expected outputs only.  Every reference result is also asserted against the functional
oracle here, i.e. generating the goldens IS the pinning of oracle/{edvr,mfdn,inner}.py.
DCN itself cannot be executed from the reference (CUDA only): dcn_*.npz is produced by the
independent torch-gather formulation (oracle/dcn.py:gather_reference) in fp64 -> "parity
unpinned by reference execution", see DESIGN.md.
"""
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DYNAVSR_REFERENCE", "/root/reference/codes")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from dynavsr_amd import synth  # noqa: E402
from oracle import dcn as odcn, edvr as oedvr, mfdn as omfdn, inner as oinner  # noqa: E402


def import_reference():
    sys.modules["models.archs.dcn.deform_conv_cuda"] = types.ModuleType("deform_conv_cuda")
    cv2 = types.ModuleType("cv2")
    sys.modules.setdefault("cv2", cv2)
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    tv.utils = tvu
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.utils", tvu)
    import models.archs.dcn  # noqa: F401
    sys.modules["models.archs.dcn.deform_conv"].modulated_deform_conv = odcn.modulated_deform_conv
    import models.archs.EDVR_arch as E
    import models.archs.LRimg_estimator as L
    import models
    import utils.util as U
    return E, L, models, U


def relerr(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).norm() / (b.norm() + 1e-30))


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def dcn_cases():
    """G1: DCN forward + 5 gradients, fp64, gather formulation.  Case B adds hand-placed offsets:
    exactly-integer positions, -1+eps, H-eps, far outside, and the +-1 boundaries themselves."""
    for tag, (n, c, dg, h, w, co, seed) in {"a": (2, 16, 4, 7, 9, 12, 11), "b": (1, 32, 8, 10, 12, 32, 12)}.items():
        r = np.random.RandomState(seed)
        x = torch.from_numpy(r.standard_normal((n, c, h, w)))
        off = torch.from_numpy(r.standard_normal((n, dg * 18, h, w)) * 2.0)
        if tag == "b":
            o = off.view(n, dg, 9, 2, h, w)
            o[0, 0, :, :, 0, :] = 0.0                     # integer sampling positions
            o[0, 1, :, 0, 1, :] = -1.0 + 1e-3              # just inside the top gate at row 1/tap 0
            o[0, 1, 0, 0, 0, :] = -1e-3                    # h_im = -1 - 1e-3 -> outside
            o[0, 2, :, 0, h - 1, :] = 1.0 - 1e-3           # h_im = H - 1e-3 at the bottom row, tap 1
            o[0, 3, :, :, 3, :] = 50.0                     # far outside
            o[0, 4, :, :, 4, :] = -50.0
            o[0, 5, 4, 0, 0, :] = -1.0                     # centre tap on row 0 -> h_im == -1 exactly
            o[0, 5, 4, 1, :, w - 1] = 1.0                  # w_im == W exactly
        msk = torch.from_numpy(r.random_sample((n, dg * 9, h, w)))
        wt = torch.from_numpy(r.standard_normal((co, c, 3, 3)) / np.sqrt(c * 9.0))
        b = torch.from_numpy(r.standard_normal(co) * 0.1)
        go = torch.from_numpy(r.standard_normal((n, co, h, w)))
        leaves = [t.clone().requires_grad_(True) for t in (x, off, msk, wt, b)]
        out = odcn.gather_reference(*leaves, 1, 1, 1, 1, dg)
        grads = torch.autograd.grad(out, leaves, go)
        # the C oracle must agree with the independent formulation before anything is written
        leaves2 = [t.clone().requires_grad_(True) for t in (x, off, msk, wt, b)]
        out2 = odcn.modulated_deform_conv(*leaves2, 1, 1, 1, 1, dg)
        grads2 = torch.autograd.grad(out2, leaves2, go)
        assert relerr(out2, out) < 1e-12, relerr(out2, out)
        for g2, g1 in zip(grads2, grads):
            assert relerr(g2, g1) < 1e-11, relerr(g2, g1)
        save("dcn_" + tag, x=x, offset=off, mask=msk, weight=wt, bias=b, gout=go, out=out,
             gx=grads[0], goffset=grads[1], gmask=grads[2], gweight=grads[3], gbias=grads[4],
             dg=dg)


def load_sd(module, sd):
    module.load_state_dict(OrderedDict((k, v.clone()) for k, v in sd.items()), strict=True)
    return module


def pcd_tsa(E):
    P = synth.edvr_state_dict(3)
    pcd = E.PCD_Align(nf=64, groups=8)
    load_sd(pcd, OrderedDict((k[len("pcd_align."):], v) for k, v in P.items() if k.startswith("pcd_align.")))
    r = np.random.RandomState(21)
    mk = lambda s: torch.from_numpy(r.standard_normal((1, 64, s, s)).astype(np.float32) * 0.5)
    nbr, ref = [mk(16), mk(8), mk(4)], [mk(16), mk(8), mk(4)]
    with torch.no_grad():
        y = pcd(nbr, ref)
        yo = oedvr.pcd_align(P, nbr, ref, 8)
    assert relerr(yo, y) < 1e-6, relerr(yo, y)
    save("pcd_align", seed=3, nbr0=nbr[0], nbr1=nbr[1], nbr2=nbr[2], ref0=ref[0], ref1=ref[1],
         ref2=ref[2], out=y)
    tsa = E.TSA_Fusion(nf=64, nframes=5, center=2)
    load_sd(tsa, OrderedDict((k[len("tsa_fusion."):], v) for k, v in P.items() if k.startswith("tsa_fusion.")))
    a = torch.from_numpy(r.standard_normal((1, 5, 64, 16, 16)).astype(np.float32) * 0.3)
    with torch.no_grad():
        y = tsa(a.clone())
        yo = oedvr.tsa_fusion(P, a, 2)
    assert relerr(yo, y) < 1e-6, relerr(yo, y)
    save("tsa_fusion", seed=3, aligned=a, out=y)


FULL_GRADS = ("conv_first.weight", "pcd_align.L1_dcnpack.conv_offset_mask.bias",
              "pcd_align.L3_dcnpack.weight", "tsa_fusion.tAtt_1.bias", "conv_last.weight")


def edvr_full(E):
    """G4: EDVR-M x4 forward + d(charbonnier)/d(params) on two sizes (second is non-square)."""
    P = synth.edvr_state_dict(0)
    for tag, (h, w, seed) in {"16x16": (16, 16, 1), "32x48": (32, 48, 2)}.items():
        net = load_sd(E.EDVR(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=4), P)
        x = synth.clip(seed, 1, 5, h, w)
        tgt = synth.clip(seed + 100, 1, 1, 4 * h, 4 * w)[:, 0]
        y = net(x.clone())
        loss = oedvr.charbonnier(y, tgt)
        loss.backward()
        ref_g = OrderedDict((k, p.grad.detach()) for k, p in net.named_parameters())
        PO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in P.items())
        yo = oedvr.edvr_forward(PO, x)
        lo = oedvr.charbonnier(yo, tgt)
        og = torch.autograd.grad(lo, list(PO.values()))
        assert relerr(yo, y) < 1e-5, relerr(yo, y)
        worst = max(relerr(a, ref_g[k]) for k, a in zip(PO, og))
        assert worst < 1e-4, worst
        save("edvr_" + tag, wseed=0, xseed=seed, tseed=seed + 100, h=h, w=w, out=y, loss=float(loss),
             grad_norms=np.array([float(g.norm()) for g in ref_g.values()]),
             **{"grad__" + k.replace(".", "__"): ref_g[k] for k in FULL_GRADS})


def edvr_x2(E):
    """EDVR-M x2 (the shipped x2 YAMLs, e.g. options/test/EDVR/EDVR_V.yml: nf 64, back_RBs 10, scale 2): one
    pixel-shuffle stage less (EDVR_arch.py:244-245,303-304) and a x2 bilinear base."""
    cfg = dict(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=2)
    P = synth.edvr_state_dict(6, **cfg)
    net = load_sd(E.EDVR(**cfg), P)
    h, w, seed = 24, 32, 7
    x = synth.clip(seed, 1, 5, h, w)
    tgt = synth.clip(seed + 100, 1, 1, 2 * h, 2 * w)[:, 0]
    y = net(x.clone())
    loss = oedvr.charbonnier(y, tgt)
    loss.backward()
    ref_g = OrderedDict((k, p.grad.detach()) for k, p in net.named_parameters())
    PO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in P.items())
    yo = oedvr.edvr_forward(PO, x, scale=2)
    og = torch.autograd.grad(oedvr.charbonnier(yo, tgt), list(PO.values()))
    assert relerr(yo, y) < 1e-5, relerr(yo, y)
    assert max(relerr(a, ref_g[k]) for k, a in zip(PO, og)) < 1e-4
    save("edvr_x2_24x32", wseed=6, xseed=seed, tseed=seed + 100, h=h, w=w, out=y, loss=float(loss),
         grad_norms=np.array([float(g.norm()) for g in ref_g.values()]),
         **{"grad__" + k.replace(".", "__"): ref_g[k] for k in FULL_GRADS if k in ref_g})


def edvr_l(E):
    """BASELINE.json configs[4] / SURVEY 8d config 5: EDVR-L x4 (nf 128, 7 frames, 40 reconstruction blocks,
    options/test/EDVR/EDVR_V_S4.yml:52-58) on a 1x7x3x64x64 clip (256x256 HR tile), forward + d(charbonnier)/d(all
    parameters) through the reference module.  The 3x256x256 output is stored sub-sampled (every 4th pixel) with
    its norm and sum -- the -m gpu test compares the full tensor against the oracle, which is pinned here."""
    cfg = dict(nf=128, nframes=7, groups=8, front_RBs=5, back_RBs=40, scale=4)
    P = synth.edvr_state_dict(8, **cfg)
    net = load_sd(E.EDVR(**cfg), P)
    h, w, seed = 64, 64, 9
    x = synth.clip(seed, 1, 7, h, w)
    tgt = synth.clip(seed + 100, 1, 1, 4 * h, 4 * w)[:, 0]
    y = net(x.clone())
    loss = oedvr.charbonnier(y, tgt)
    loss.backward()
    ref_g = OrderedDict((k, p.grad.detach()) for k, p in net.named_parameters())
    PO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in P.items())
    yo = oedvr.edvr_forward(PO, x, **cfg)
    og = torch.autograd.grad(oedvr.charbonnier(yo, tgt), list(PO.values()))
    assert relerr(yo, y) < 1e-5, relerr(yo, y)
    worst = max(relerr(a, ref_g[k]) for k, a in zip(PO, og))
    assert worst < 1e-4, worst
    save("edvr_l_64x64", wseed=8, xseed=seed, tseed=seed + 100, h=h, w=w, out_sub=y[:, :, ::4, ::4],
         out_norm=float(y.double().norm()), out_sum=float(y.double().sum()), loss=float(loss),
         grad_norms=np.array([float(g.norm()) for g in ref_g.values()]),
         **{"grad__" + k.replace(".", "__"): ref_g[k] for k in
            ("conv_first.weight", "pcd_align.L1_dcnpack.conv_offset_mask.bias", "tsa_fusion.tAtt_1.bias",
             "recon_trunk.39.conv2.bias", "conv_last.weight")})


def tof_cases():
    """the reference's TOFlow module (TOF_arch.py; plain torch, runs on CPU as shipped) on 1x7x3x32x48
    clips, adapt_official=True as networks.py:39 builds it: eval mode (running statistics) forward, and training
    mode (batch statistics, the mode the inner MAML step runs in) forward + d(charbonnier)/d(all parameters) +
    the updated running estimates.  oracle/tof.py is asserted against the module before anything is written."""
    import models.archs.TOF_arch as TOF
    from oracle import tof as otof
    P = synth.tof_state_dict(2)
    h, w, seed = 32, 48, 11
    x = synth.clip(seed, 1, 7, h, w)
    tgt = synth.clip(seed + 100, 1, 1, h, w)[:, 0]
    net = load_sd(TOF.TOFlow(adapt_official=True), P)
    net.eval()
    with torch.no_grad():
        y_eval = net(x.clone())
        yo = otof.toflow_forward(OrderedDict((k, v.clone()) for k, v in P.items()), x, training=False)
    assert relerr(yo, y_eval) < 1e-6, relerr(yo, y_eval)
    net = load_sd(TOF.TOFlow(adapt_official=True), P)
    net.train()
    y = net(x.clone())
    loss = oedvr.charbonnier(y, tgt)
    loss.backward()
    ref_g = OrderedDict((k, p.grad.detach()) for k, p in net.named_parameters())
    PO = OrderedDict((k, (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()))
                     for k, v in P.items())
    taps = {}
    yo = otof.toflow_forward(PO, x, training=True, taps=taps)
    og = torch.autograd.grad(oedvr.charbonnier(yo, tgt), [PO[k] for k in ref_g])
    assert relerr(yo, y) < 1e-6, relerr(yo, y)
    worst = max(relerr(a, ref_g[k]) for k, a in zip(ref_g, og))
    assert worst < 1e-4, worst
    sd_after = net.state_dict()
    for k in sd_after:
        if "running" in k:
            assert relerr(PO[k], sd_after[k]) < 1e-6, k
    keep = ("SpyNet.blocks.0.block.0.weight", "SpyNet.blocks.3.block.12.bias", "SpyNet.blocks.2.block.4.weight",
            "conv_64_64_1x1.weight", "conv_64_3_1x1.bias")
    save("tof_32x48", wseed=2, xseed=seed, tseed=seed + 100, h=h, w=w, out_eval=y_eval, out_train=y, loss=float(loss.detach()),
         flow_last=torch.stack(taps["flow_l3"], 1), warped=taps["warped"],
         grad_names=np.array(list(ref_g.keys())), grad_norms=np.array([float(g.norm()) for g in ref_g.values()]),
         running_mean_b3_1=sd_after["SpyNet.blocks.3.block.1.running_mean"],
         running_var_b3_1=sd_after["SpyNet.blocks.3.block.1.running_var"],
         **{"grad__" + k.replace(".", "__"): ref_g[k] for k in keep})
    # sizes that are NOT multiples of 16 (the drivers feed 180x320 = a 45x80 SLR clip x4, Vid4's 144x180, 22x22 patches):
    # the pyramid floors, the first flow is H//16 x W//16 zeros resized to each level's own size (TOF_arch.py:69-90)
    for (h, w, seed) in ((44, 40, 21), (20, 30, 22)):
        x = synth.clip(seed, 1, 7, h, w)
        tgt = synth.clip(seed + 100, 1, 1, h, w)[:, 0]
        net = load_sd(TOF.TOFlow(adapt_official=True), P)
        net.eval()
        with torch.no_grad():
            y_eval = net(x.clone())
            yo = otof.toflow_forward(OrderedDict((k, v.clone()) for k, v in P.items()), x, training=False)
        assert relerr(yo, y_eval) < 1e-6, relerr(yo, y_eval)
        net.train()
        y = net(x.clone())
        loss = oedvr.charbonnier(y, tgt)
        loss.backward()
        ref_g = OrderedDict((k, p.grad.detach()) for k, p in net.named_parameters())
        save("tof_%dx%d" % (h, w), wseed=2, xseed=seed, tseed=seed + 100, h=h, w=w, out_eval=y_eval, out_train=y,
             loss=float(loss.detach()), grad_names=np.array(list(ref_g.keys())),
             grad_norms=np.array([float(g.norm()) for g in ref_g.values()]),
             **{"grad__" + k.replace(".", "__"): ref_g[k] for k in keep[:2]})


def duf_cases():
    """the reference's DUF modules (DUF_arch.py; plain torch) on CPU, adapt_official=True as networks.py:29-36
    builds them.  DUF_16L x4 on 1x7x3x16x24: eval forward; training-mode forward + d(charbonnier)/d(all parameters)
    + a running estimate.  DUF_16L x2, DUF_28L x4 and DUF_52L x3 on 1x7x3x8x12: eval forward (pins the block structure
    and the other scales)."""
    import models.archs.DUF_arch as DUF
    from oracle import duf as oduf
    P = synth.duf_state_dict(3, 16, 4)
    h, w, seed = 16, 24, 13
    x = synth.clip(seed, 1, 7, h, w)
    tgt = synth.clip(seed + 100, 1, 1, 4 * h, 4 * w)[:, 0]
    net = load_sd(DUF.DUF_16L(scale=4, adapt_official=True), P).eval()
    with torch.no_grad():
        y_eval = net(x.clone())
        yo = oduf.duf_forward(OrderedDict((k, v.clone()) for k, v in P.items()), x, 16, 4, True, False)
    assert relerr(yo, y_eval) < 1e-6, relerr(yo, y_eval)
    net = load_sd(DUF.DUF_16L(scale=4, adapt_official=True), P).train()
    y = net(x.clone())
    loss = oedvr.charbonnier(y, tgt)
    loss.backward()
    ref_g = OrderedDict((k, p.grad.detach()) for k, p in net.named_parameters())
    PO = OrderedDict((k, (v.clone().requires_grad_(True) if k in ref_g else v.clone())) for k, v in P.items())
    yo = oduf.duf_forward(PO, x, 16, 4, True, True)
    og = torch.autograd.grad(oedvr.charbonnier(yo, tgt), [PO[k] for k in ref_g])
    assert relerr(yo, y) < 1e-6, relerr(yo, y)
    worst = max(relerr(a, ref_g[k]) for k, a in zip(ref_g, og) if float(ref_g[k].norm()) > 1e-7)
    assert worst < 1e-4, worst
    sd_after = net.state_dict()
    assert relerr(PO["bn3d_2.running_var"], sd_after["bn3d_2.running_var"]) < 1e-6
    keep = ("conv3d_1.weight", "dense_block_1.conv3d_2.weight", "dense_block_2.bn3d_5.weight", "conv3d_r2.bias", "conv3d_f2.bias")
    out = dict(wseed=3, xseed=seed, tseed=seed + 100, h=h, w=w, out_eval=y_eval, out_train=y, loss=float(loss.detach()),
               grad_names=np.array(list(ref_g.keys())), grad_norms=np.array([float(g.norm()) for g in ref_g.values()]),
               running_mean_bn3d_2=sd_after["bn3d_2.running_mean"], running_var_bn3d_2=sd_after["bn3d_2.running_var"],
               **{"grad__" + k.replace(".", "__"): ref_g[k] for k in keep})
    for layers, scale, cls in ((16, 2, DUF.DUF_16L), (28, 4, DUF.DUF_28L), (52, 3, DUF.DUF_52L)):
        Pv = synth.duf_state_dict(4, layers, scale)
        xv = synth.clip(seed + layers, 1, 7, 8, 12)
        netv = load_sd(cls(scale=scale, adapt_official=True), Pv).eval()
        with torch.no_grad():
            yv = netv(xv.clone())
            yov = oduf.duf_forward(OrderedDict((k, v.clone()) for k, v in Pv.items()), xv, layers, scale, True, False)
        assert relerr(yov, yv) < 1e-6, (layers, relerr(yov, yv))
        out["out_eval_%dL_x%d" % (layers, scale)] = yv
    save("duf_16x24", **out)


def crop_cases():
    """train.maml.use_patch: the `crop` closure of test_dynavsr.py:118-145 cannot be imported (it lives inside main()),
    so its body is driven here statement by statement over the reference's own preprocessing.common_crop, with a
    seeded `random`: positions are recovered from a coordinate-ramp input, crops of a random clip are stored."""
    import random
    from data.meta_learner import preprocessing
    from oracle import inner as oin
    t, c, h, w, s, n, psz = 5, 3, 22, 30, 4, 6, 16
    r = np.random.RandomState(91)
    seq = torch.from_numpy(r.rand(1, t, c, h, w).astype(np.float32))
    hr = torch.from_numpy(r.rand(1, c, s * h, s * w).astype(np.float32))
    ramp = (torch.arange(h)[:, None] * 1000 + torch.arange(w)[None, :]).float().expand(1, 1, 1, h, w)
    random.seed(1234)
    lrs, hrs = [], []
    for _ in range(n):
        a, b = preprocessing.common_crop(seq[0], hr[0], patch_size=psz // 2)
        lrs.append(a); hrs.append(b)
    lr_p, hr_p = torch.stack(lrs, 0), torch.stack(hrs, 0)
    random.seed(1234)
    pos = []
    for _ in range(n):
        a, _b = preprocessing.common_crop(ramp[0], hr[0], patch_size=psz // 2)
        v = int(a[0, 0, 0, 0])
        pos.append((v // 1000, v % 1000))
    random.seed(1234)
    o_lr, o_hr = oin.crop(seq, hr, n, psz)
    assert torch.equal(o_lr, lr_p) and torch.equal(o_hr, hr_p)
    save("crop_patches", seed=1234, dseed=91, t=t, c=c, h=h, w=w, scale=s, n=n, patch_size=psz,
         py=np.array([p[0] for p in pos]), px=np.array([p[1] for p in pos]), lr=lr_p, hr_sum=hr_p.double().sum(dim=(1, 2, 3)))


def mfdn_full(L):
    """G5: MFDN x4 forward/backward on 1x5x3x32x32 through the reference module."""
    M = synth.mfdn_state_dict(0)
    net = load_sd(L.DirectKernelEstimatorVideo(nf=64, in_nc=3, scale=4), M)
    lq = synth.clip(5, 1, 5, 32, 32)
    y = net(lq.transpose(1, 2)).transpose(1, 2)
    go = torch.from_numpy(np.random.RandomState(6).standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(go)
    MO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in M.items())
    yo = omfdn.mfdn_forward(MO, lq)
    og = torch.autograd.grad(yo, list(MO.values()), go)
    assert relerr(yo, y) < 1e-6
    for (k, p), g in zip(net.named_parameters(), og):
        assert relerr(g, p.grad) < 1e-4, (k, relerr(g, p.grad))
    save("mfdn_32x32", wseed=0, xseed=5, goseed=6, out=y,
         grad_norms=np.array([float(p.grad.norm()) for p in net.parameters()]),
         grad__conv6__weight=net.conv6.weight.grad, grad__conv0__bias=net.conv0.bias.grad)


def estimator_variants(L):
    """G5b: MFDN x2 on 1x3x3x24x40 (3x3 conv3 variant, T = 3) and SFDN on 2x3x20x28 through the reference
    modules; every parameter gradient is stored (the nets are small at nf = 16)."""
    nf = 16
    M = synth.mfdn_state_dict(3, nf=nf, scale=2)
    net = load_sd(L.DirectKernelEstimatorVideo(nf=nf, in_nc=3, scale=2), M)
    lq = synth.clip(15, 1, 3, 24, 40)
    y = net(lq.transpose(1, 2)).transpose(1, 2)
    go = torch.from_numpy(np.random.RandomState(16).standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(go)
    MO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in M.items())
    yo = omfdn.mfdn_forward(MO, lq, scale=2)
    og = torch.autograd.grad(yo, list(MO.values()), go)
    assert relerr(yo, y) < 1e-6
    for (k, p), g in zip(net.named_parameters(), og):
        assert relerr(g, p.grad) < 1e-4, (k, relerr(g, p.grad))
    save("mfdn_x2_24x40", wseed=3, xseed=15, goseed=16, nf=nf, out=y,
         **{"grad__" + k.replace(".", "__"): p.grad for k, p in net.named_parameters()})

    S = synth.sfdn_state_dict(4, nf=nf)
    net = load_sd(L.DirectKernelEstimator_CMS(nf=nf), S)
    x = synth.clip(17, 2, 1, 20, 28)[:, 0]
    y = net(x)
    go = torch.from_numpy(np.random.RandomState(18).standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(go)
    SO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in S.items())
    yo = omfdn.sfdn_forward(SO, x)
    og = torch.autograd.grad(yo, list(SO.values()), go)
    assert relerr(yo, y) < 1e-6
    for (k, p), g in zip(net.named_parameters(), og):
        assert relerr(g, p.grad) < 1e-4, (k, relerr(g, p.grad))
    save("sfdn_20x28", wseed=4, xseed=17, goseed=18, nf=nf, out=y,
         **{"grad__" + k.replace(".", "__"): p.grad for k, p in net.named_parameters()})


def make_opt(optimizer):
    from options.options import dict_to_nonedict
    return dict_to_nonedict({
        "name": "golden", "model": "video_base+lrimgestimator", "scale": 4, "gpu_ids": None,
        "dist": False, "is_train": False, "distortion": "sr",
        "datasets": {"train": {"kernel_size": 21, "patch_size": 128, "batch_size": 1},
                     "val": {"N_frames": 5}},
        "network_G": {"which_model_G": "EDVR", "nf": 64, "nframes": 5, "groups": 8, "front_RBs": 5,
                      "back_RBs": 10, "predeblur": False, "HR_in": False, "w_TSA": True},
        "network_E": {"which_model_E": "MFDN", "mode": "video", "nf": 64, "in_nc": 3},
        "path": {"strict_load": True},
        "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "loss_ftn": "l1", "use_real": False,
                  "maml": {"optimizer": optimizer, "lr_alpha": 1e-5, "beta1": 0.9, "beta2": 0.99,
                           "adapt_iter": 1, "use_patch": False}}})


TRACK = ("conv_first.weight", "pcd_align.cas_dcnpack.conv_offset_mask.bias",
         "tsa_fusion.fea_fusion.bias", "recon_trunk.9.conv2.weight", "conv_last.bias")
TRACK_E = ("conv0.weight", "conv6.bias")


def inner_step(models, U):
    """G6: one inner step exactly as codes/test_dynavsr.py:208-283 drives the reference wrappers
    (CPU, gpu_ids=None).  LR 64x64 -> SLR 16x16 (BASELINE.json configs[0])."""
    from copy import deepcopy
    import torch.nn.functional as F
    PG, PE, PEF = synth.edvr_state_dict(0), synth.mfdn_state_dict(0), synth.mfdn_state_dict(1)
    lqs = synth.clip(1, 1, 5, 64, 64)
    for optimizer in ("SGD", "Adam"):
        opt = make_opt(optimizer)
        model, est_model = models.create_model(opt)
        modelcp, est_modelcp = models.create_model(opt)
        _, est_fixed = models.create_model(opt)
        load_sd(model.netG.module, PG); load_sd(est_model.netE.module, PE); load_sd(est_fixed.netE.module, PEF)
        val_data = {"LQs": lqs}
        modelcp.netG, est_modelcp.netE = deepcopy(model.netG), deepcopy(est_model.netE)
        params = [v for _, v in modelcp.netG.named_parameters()] + [v for _, v in est_modelcp.netE.named_parameters()]
        m = opt["train"]["maml"]
        io = (torch.optim.Adam(params, lr=m["lr_alpha"], betas=(m["beta1"], m["beta2"]))
              if optimizer == "Adam" else torch.optim.SGD(params, lr=m["lr_alpha"]))
        est_modelcp.feed_data(val_data); est_modelcp.forward_without_optim()
        slr = est_modelcp.fake_L
        io.zero_grad()
        modelcp.feed_data({"LQs": slr, "GT": lqs[:, 2]})
        loss = modelcp.calculate_loss()
        est_fixed.feed_data(val_data); est_fixed.test()
        loss = loss + 10 * F.l1_loss(slr, est_fixed.fake_L)
        loss.backward()
        gG = OrderedDict((k, p.grad.detach().clone()) for k, p in modelcp.netG.module.named_parameters())
        gE = OrderedDict((k, p.grad.detach().clone()) for k, p in est_modelcp.netE.module.named_parameters())
        io.step()
        modelcp.feed_data({"LQs": lqs}, need_GT=False); modelcp.test()
        sr = modelcp.fake_H
        # functional oracle must reproduce it
        losses, PGa, PEa, sro = oinner.inner_adapt(PG, PE, PEF, lqs, 1, optimizer, 1e-5, (0.9, 0.99))
        assert abs(losses[0] - float(loss)) < 1e-5 * abs(float(loss)), (losses, float(loss))
        assert relerr(sro, sr) < 1e-5, relerr(sro, sr)
        new = dict(modelcp.netG.module.named_parameters()); newE = dict(est_modelcp.netE.module.named_parameters())
        arrs = {"loss": float(loss), "sr": sr if optimizer == "SGD" else sr[..., 96:160, 96:160], "slr": slr.detach(),
                "gradG_norms": np.array([float(g.norm()) for g in gG.values()]),
                "gradE_norms": np.array([float(g.norm()) for g in gE.values()])}
        for k in TRACK:
            arrs["dG__" + k.replace(".", "__")] = (new[k].detach().double() - PG[k].double()).float()
            assert relerr(PGa[k].detach().double() - PG[k].double(), new[k].detach().double() - PG[k].double()) < (1e-3 if optimizer == "SGD" else 5e-2)
        for k in TRACK_E:
            arrs["dE__" + k.replace(".", "__")] = (newE[k].detach().double() - PE[k].double()).float()
        save("inner_step_" + optimizer.lower(), **arrs)
        if optimizer == "SGD":      # G7: image conversion + PSNR of the adapted output vs LR-upsampled GT proxy
            gt = synth.clip(9, 1, 1, 256, 256)[0, 0]
            a, b = U.tensor2img(sr[0], mode="rgb"), U.tensor2img(gt, mode="rgb")
            assert (a == oinner.tensor2img_rgb(sr[0])).all()
            p = U.calculate_psnr(a, b)
            assert abs(p - oinner.psnr_uint8(a, b)) < 1e-12
            save("psnr", gtseed=9, img=a, psnr=p)


def meta_step(models):
    """One outer (meta) iteration with the semantics of codes/train_dynavsr.py:265-438 (EDVR branch, use_real and
    use_patch off), evaluated on CPU with the REFERENCE's wrapper objects (its create_model / feed_data /
    calculate_loss / forward_without_optim / MyLoss are what runs) -- the driver keeps this loop inside main(), so it
    cannot be imported; the steps below are its data flow, restated:
      per task: fresh deep copies (:326) whose parameters the inner optimiser holds (:328-351); adapt_iter times:
      SLR from est_model, loss_train = model's Charbonnier on (SLR -> LR centre) + L1(SLR, SuperLQs), backward,
      inner step (:355-399; quirk Q1: the losses go through model / est_model, so the step moves nothing and the
      gradients land on the meta-parameters); then loss_q on (LR -> GT centre) / B and loss_e / (10 B) are added to
      the meta-gradients with autograd.grad (:403-426); finally one step of the meta optimiser (:438).
    B = 2 tasks, adapt_iter = 2, inner Adam, meta SGD (the update is -lr_G * meta-gradient).
    LR 32x32 -> SLR 8x8, GT 128x128."""
    from copy import deepcopy
    import torch.nn.functional as F
    PG, PE = synth.edvr_state_dict(0), synth.mfdn_state_dict(0)
    opt = make_opt("Adam")
    maml = opt["train"]["maml"]
    maml["adapt_iter"] = 2
    lr_G, n_tasks, centre = 1e-3, 2, 2
    model, est_model = models.create_model(opt)
    modelcp, est_modelcp = models.create_model(opt)
    load_sd(model.netG.module, PG); load_sd(est_model.netE.module, PE)
    g_params, e_params = list(model.netG.parameters()), list(est_model.netE.parameters())
    meta_opt = torch.optim.SGD(g_params + e_params, lr=lr_G)
    batch = {"LQs": synth.clip(31, n_tasks, 5, 32, 32), "SuperLQs": synth.clip(32, n_tasks, 5, 8, 8),
             "GT": synth.clip(33, n_tasks, 5, 128, 128)}
    meta_opt.zero_grad()
    inner_losses, loss_q_sum = [], 0.0
    for t in range(n_tasks):
        task = {k: v[t:t + 1] for k, v in batch.items()}
        modelcp.netG, est_modelcp.netE = deepcopy(model.netG), deepcopy(est_model.netE)
        inner_opt = torch.optim.Adam([{"params": list(modelcp.netG.parameters()), "lr": maml["lr_alpha"]},
                                      {"params": list(est_modelcp.netE.parameters()), "lr": maml["lr_alpha"]}],
                                     lr=maml["lr_alpha"], betas=(maml["beta1"], maml["beta2"]))
        for _ in range(maml["adapt_iter"]):
            inner_opt.zero_grad()
            est_model.feed_data(task)
            est_model.forward_without_optim()
            slr = est_model.fake_L
            model.feed_data({"LQs": slr, "GT": task["LQs"][:, centre]})
            loss_train = model.calculate_loss() + F.l1_loss(slr, task["SuperLQs"])
            loss_train.backward()
            inner_opt.step()
            inner_losses.append(float(loss_train.detach()))
        model.feed_data({"LQs": task["LQs"], "GT": task["GT"][:, centre]})
        loss_q = model.calculate_loss()
        for p, g in zip(g_params, torch.autograd.grad(loss_q / n_tasks, g_params)):
            p.grad += g
        est_model.feed_data(task)
        est_model.forward_without_optim()
        loss_e = est_model.MyLoss(est_model.fake_L, est_model.real_L)
        for p, g in zip(e_params, torch.autograd.grad(loss_e / (n_tasks * 10), e_params)):
            p.grad += g
        loss_q_sum += float(loss_q.detach()) / n_tasks
    gG = OrderedDict((k, p.grad.detach().clone()) for k, p in model.netG.module.named_parameters())
    gE = OrderedDict((k, p.grad.detach().clone()) for k, p in est_model.netE.module.named_parameters())
    meta_opt.step()
    newG = dict(model.netG.module.named_parameters())
    arrs = {"loss_q": loss_q_sum, "loss_train": np.array(inner_losses), "lr_G": lr_G,
            "gradG_norms": np.array([float(g.norm()) for g in gG.values()]),
            "gradE_norms": np.array([float(g.norm()) for g in gE.values()])}
    for k in TRACK:
        arrs["gG__" + k.replace(".", "__")] = gG[k]
        assert relerr(newG[k].detach().double() - PG[k].double(), -lr_G * gG[k].double()) < 1e-3   # rounding of p - lr*g in fp32
    for k in TRACK_E:
        arrs["gE__" + k.replace(".", "__")] = gE[k]
    save("meta_step", **arrs)


def degradation_cases():
    """Degradation (codes/data/random_kernel_generator.py) run as shipped, on CPU.  Its kernel_shift calls
    np.int (:72), removed in numpy 1.24: for this run only the name is restored as the alias of int it was.
    Cases: scale 4 and 2, an isotropic / two anisotropic kernels / the delta kernel, a single image, the
    per-frame kernel array with T and T + 2 frames, and vsrbase.py:184-186's HR -> LR -> (8-bit) -> SLR chain."""
    from oracle import degradation as od
    if not hasattr(np, "int"):
        np.int = int
    import data.random_kernel_generator as rkg
    out = {}
    seed = [770]

    def frames(*shape):          # inputs are regenerated from the seed by the tests: fixtures hold outputs only
        seed[0] += 1
        return np.random.RandomState(seed[0]).rand(*shape).astype(np.float32), seed[0]
    cases = [("s4_aniso", 21, 4, 0.7, [2.0, 0.8], (5, 3, 64, 80)), ("s2_aniso", 21, 2, -1.9, [0.4, 3.1], (5, 3, 40, 48)),
             ("s4_iso", 21, 4, 0.0, [1.0, 1.0], (2, 3, 36, 36)), ("s4_delta", 21, 4, 0.0, [0, 0], (2, 3, 32, 40)),
             ("s2_k11", 11, 2, 0.3, [1.5, 0.6], (3, 3, 24, 28)),
             ("s2_k10_even", 10, 2, 0.5, [1.2, 0.7], (2, 3, 22, 26))]    # even kernel_size: K stays even after the shift pad
    for tag, ks, scale, theta, sigma, shape in cases:
        img, sd = frames(*shape)
        d = rkg.Degradation(ks, scale, theta=theta, sigma=sigma)
        lr = d.apply(torch.from_numpy(img)).numpy()
        assert np.allclose(d.kernel, od.build_kernel(ks, theta, sigma), rtol=0, atol=1e-15)
        assert np.allclose(d.kernel_shift(d.kernel), od.kernel_shift(d.kernel, scale), rtol=0, atol=1e-15)
        e = np.abs(lr - od.apply(img, d.kernel, scale)).max()
        assert e < 2e-6, (tag, e)
        out.update({tag + "__seed_shape": np.array((sd,) + shape), tag + "__params": np.array([ks, scale, theta, sigma[0], sigma[1]]),
                    tag + "__kernel": d.kernel, tag + "__shifted": d.kernel_shift(d.kernel), tag + "__lr": lr})
    # single image C H W
    img, sd = frames(3, 32, 32)
    d = rkg.Degradation(21, 4, theta=1.1, sigma=[3.0, 1.2])
    out.update({"single__seed_shape": np.array((sd, 3, 32, 32)), "single__params": np.array([21, 4, 1.1, 3.0, 1.2]),
                "single__lr": d.apply(torch.from_numpy(img)).numpy()})
    # one kernel per frame (T = 5), clips of T and T + 2 frames
    kset = np.stack([rkg.Degradation(21, 4, theta=-3 + i, sigma=[0.5 + i, 4.0 - 0.7 * i]).kernel for i in range(5)], 0)
    d = rkg.Degradation(21, 4)
    d.set_kernel_directly(kset)
    for n in (5, 7):
        img, sd = frames(n, 3, 32, 36)
        lr = d.apply(torch.from_numpy(img)).numpy()
        assert np.abs(lr - od.apply(img, kset, 4)).max() < 2e-6
        out.update({"perframe%d__seed_shape" % n: np.array((sd, n, 3, 32, 36)), "perframe%d__lr" % n: lr})
    out["perframe__kernels"] = kset
    # the dataset's chain (vsrbase.py:184-186)
    img, sd = frames(5, 3, 128, 128)
    d = rkg.Degradation(21, 4, theta=0.4, sigma=[1.7, 2.9])
    lr = d.apply(torch.from_numpy(img)).mul(255).clamp(0, 255).round().div(255)
    slr = d.apply(lr)
    out.update({"chain__seed_shape": np.array((sd, 5, 3, 128, 128)), "chain__params": np.array([21, 4, 0.4, 1.7, 2.9]), "chain__lr": lr.numpy(),
                "chain__slr": slr.numpy()})
    save("degradation", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    E, L, models, U = import_reference()
    which = sys.argv[1:] or ["dcn", "pcd", "edvr", "mfdn", "estimators", "inner", "degradation", "meta", "tof", "duf", "crop"]
    if "dcn" in which: dcn_cases()
    if "pcd" in which: pcd_tsa(E)
    if "edvr" in which: edvr_full(E)
    if "edvr" in which or "edvr_x2" in which: edvr_x2(E)
    if "crop" in which: crop_cases()
    if "tof" in which: tof_cases()
    if "duf" in which: duf_cases()
    if "edvr_l" in which: edvr_l(E)      # several minutes on 8 cores; not part of the default set
    if "mfdn" in which: mfdn_full(L)
    if "estimators" in which: estimator_variants(L)
    if "inner" in which: inner_step(models, U)
    if "degradation" in which: degradation_cases()
    if "meta" in which: meta_step(models)
