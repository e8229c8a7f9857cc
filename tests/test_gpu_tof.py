"""GPU parity of the TOFlow backbone: every op against a plain PyTorch fp64 CPU reference of the same
op, and the whole module (training and eval mode, forward and backward) against the golden produced by the
reference's own TOFlow module and against oracle/tof.py.  Tolerances: forward rel-L2 <= 2e-5 per op / 2e-4 for the
network, gradients rel-L2 <= 2e-4 per op."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, relerr
from dynavsr_amd import synth

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape) * scale).float()


@pytest.mark.parametrize("ks,cin,cout,h,w", [(7, 8, 32, 20, 37), (7, 64, 32, 16, 32), (7, 16, 2, 9, 33), (9, 21, 64, 24, 40),
                                             (9, 64, 64, 16, 16), (1, 64, 3, 12, 20),
                                             # W % 4 == 0: the row-split DMA kernel (forward and / or data gradient)
                                             (7, 8, 32, 21, 36), (7, 16, 2, 12, 36), (7, 32, 64, 41, 100), (7, 32, 16, 8, 64),
                                             (9, 64, 64, 30, 72), (9, 21, 64, 9, 132)])
def test_tof_conv_forward_backward(ks, cin, cout, h, w):
    """SpyNet's 7x7, the head's 9x9 / 1x1: bias + ReLU + residual fused forward, and dX / dW / db."""
    import ctypes
    from dynavsr_amd import _lib as L, tofops as T
    if ks > 3:   # which kernel: row-split DMA (geo[3] == 2) exactly when W % 4 == 0
        xa = torch.empty(2, cin, h, w, device="cuda")
        geo = (ctypes.c_int * 4)()
        d = L.Conv2dDesc(L.ptr(xa), None, None, None, None, None, 2, cin, 0, h, w, cout, ks, 1, ks // 2, 0, 0, 1, 0, 0)
        L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
        assert (geo[3] == 2) == (w % 4 == 0)
    x, wt, b = rnd(2, cin, h, w, seed=1), rnd(cout, cin, ks, ks, seed=2, scale=(cin * ks * ks) ** -0.5), rnd(cout, seed=3, scale=0.1)
    res, go = rnd(2, cout, h, w, seed=4), rnd(2, cout, h, w, seed=5)
    for act, use_res in ((L.ACT_NONE, True), (L.ACT_RELU, False)):
        xd, wd, bd, rd = [t.double().requires_grad_() for t in (x, wt, b, res)]
        y = F.conv2d(xd, wd, bd, padding=ks // 2)
        y = (F.relu(y) if act == L.ACT_RELU else y) + (rd if use_res else 0)
        gr = torch.autograd.grad(y, [xd, wd, bd] + ([rd] if use_res else []), go.double())
        xg, wg, bg, rg = [t.cuda().requires_grad_() for t in (x, wt, b, res)]
        yg = T.conv(xg, wg, bg, res=rg if use_res else None, act=act)
        gg = torch.autograd.grad(yg, [xg, wg, bg] + ([rg] if use_res else []), go.cuda())
        assert relerr(yg, y) < 2e-5
        for a, r_ in zip(gg, gr):
            assert relerr(a, r_) < 2e-4


@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (1, 3, 7, 9), (3, 3, 32, 48)])
def test_flow_warp_forward_backward(shape):
    """arch_util.flow_warp: against the reference's own formulation (grid / (size-1), grid_sample align_corners=False)
    in fp64, with flows that leave the image, land on integer positions and on the -0.5 / W-0.5 borders."""
    from dynavsr_amd import tofops as T
    from oracle import tof as otof
    n, c, h, w = shape
    x = rnd(n, c, h, w, seed=1)
    flow = rnd(n, 2, h, w, seed=2, scale=2.5)
    flow[0, :, 0, :] = 0.0                      # exact grid positions
    flow[0, 0, 1, :] = 40.0                     # far outside
    flow[0, 1, 2, :] = -40.0
    flow[0, 0, 3, 0] = -0.5 * (w - 1) / w       # ix == -0.5 exactly for x = 0 ... (border of the zero padding)
    go = rnd(n, c, h, w, seed=3)
    xd, fd = x.double().requires_grad_(), flow.double().requires_grad_()
    y = otof.flow_warp(xd, fd.permute(0, 2, 3, 1))
    gx, gf = torch.autograd.grad(y, [xd, fd], go.double())
    xg, fg = x.cuda().requires_grad_(), flow.cuda().requires_grad_()
    yg = T.flow_warp(xg, fg)
    ggx, ggf = torch.autograd.grad(yg, [xg, fg], go.cuda())
    assert relerr(yg, y) < 2e-5 and relerr(ggx, gx) < 2e-5
    # d/dflow flips sign at integer sampling positions (floor kink): compare away from them, like the DCN tests
    frac = (torch.stack([(torch.arange(w)[None, None, :] + fd[:, 0].detach()) * w / (w - 1) - 0.5,
                         (torch.arange(h)[None, :, None] + fd[:, 1].detach()) * h / (h - 1) - 0.5], 1) % 1.0)
    smooth = ((frac > 1e-3) & (frac < 1 - 1e-3)).all(1, keepdim=True).expand_as(gf)
    assert float((ggf.cpu().double() - gf)[smooth].norm() / gf[smooth].norm()) < 2e-4
    with pytest.raises(RuntimeError, match="flow must be"):
        T.flow_warp(xg, fg[:, :, :-1])


def test_avgpool_resize_affine_ops():
    from dynavsr_amd import tofops as T
    x = rnd(2, 3, 18, 28, seed=1)
    go = rnd(2, 3, 9, 14, seed=2)
    xd = x.double().requires_grad_()
    y = F.avg_pool2d(xd, 2, 2, count_include_pad=False)
    (gr,) = torch.autograd.grad(y, xd, go.double())
    xg = x.cuda().requires_grad_()
    yg = T.avg_pool2(xg)
    (gg,) = torch.autograd.grad(yg, xg, go.cuda())
    assert relerr(yg, y) < 1e-6 and relerr(gg, gr) < 1e-6
    assert T.avg_pool2(rnd(1, 2, 7, 9).cuda()).shape == (1, 2, 3, 4)          # odd sizes: floor, last row / column dropped
    for (h, w), (ho, wo) in (((4, 6), (8, 12)), ((2, 3), (5, 7)), ((1, 1), (2, 2)), ((3, 5), (3, 5))):
        f = rnd(2, 2, h, w, seed=3)
        go = rnd(2, 2, ho, wo, seed=4)
        fd = f.double().requires_grad_()
        y = F.interpolate(fd, size=(ho, wo), mode="bilinear", align_corners=True) * 2.0
        (gr,) = torch.autograd.grad(y, fd, go.double())
        fg = f.cuda().requires_grad_()
        yg = T.resize_bilinear_ac(fg, (ho, wo), 2.0)
        (gg,) = torch.autograd.grad(yg, fg, go.cuda())
        assert relerr(yg, y) < 1e-6 and relerr(gg, gr) < 1e-5, ((h, w), (ho, wo))
    s, t = torch.tensor([2.0, 0.5, -1.0]), torch.tensor([0.1, 0.2, 0.3])
    xg = x.cuda().requires_grad_()
    yg = T.channel_affine(xg, s.cuda(), t.cuda())
    (gg,) = torch.autograd.grad(yg, xg, torch.ones_like(yg))
    assert relerr(yg, x * s.view(1, 3, 1, 1) + t.view(1, 3, 1, 1)) < 1e-7 and relerr(gg, s.view(1, 3, 1, 1).expand_as(x)) < 1e-7


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("relu", [True, False])
def test_batchnorm_forward_backward(training, relu):
    from dynavsr_amd import tofops as T
    n, c, h, w = 3, 32, 13, 21
    x = rnd(n, c, h, w, seed=1) * 2.0 + 0.5
    go = rnd(n, c, h, w, seed=2)
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * rnd(c, seed=3)); bn.bias.copy_(0.1 * rnd(c, seed=4))
        bn.running_mean.copy_(0.3 * rnd(c, seed=5)); bn.running_var.copy_(0.5 + rnd(c, seed=6).abs())
    ref = torch.nn.BatchNorm2d(c).double()
    ref.load_state_dict(bn.state_dict())
    bn, ref = bn.cuda().train(training), ref.train(training)
    xd = x.double().requires_grad_()
    y = ref(xd)
    y = F.relu(y) if relu else y
    gr = torch.autograd.grad(y, [xd, ref.weight, ref.bias], go.double())
    xg = x.cuda().requires_grad_()
    yg = T.batchnorm(xg, bn, relu=relu)
    gg = torch.autograd.grad(yg, [xg, bn.weight, bn.bias], go.cuda())
    assert relerr(yg, y) < 2e-6
    for a, r_ in zip(gg, gr):
        assert relerr(a, r_) < 2e-5
    assert relerr(bn.running_mean, ref.running_mean) < 1e-6 and relerr(bn.running_var, ref.running_var) < 1e-6
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == (1 if training else 0)


def _tof(seed):
    from dynavsr_amd.models.archs.TOF_arch import TOFlow
    net = TOFlow(adapt_official=True)
    net.load_state_dict(synth.tof_state_dict(seed), strict=True)
    return net.cuda()


def test_toflow_eval_and_training_golden():
    """The module against the reference's golden: eval forward; training forward, Charbonnier loss, all 80 parameter
    gradient norms, five full gradients, and the running statistics after the six SpyNet calls."""
    from dynavsr_amd import hipops
    g = load_golden("tof_32x48")
    h, w = int(g["h"]), int(g["w"])
    x = synth.clip(int(g["xseed"]), 1, 7, h, w).cuda()
    tgt = synth.clip(int(g["tseed"]), 1, 1, h, w)[:, 0].cuda()
    net = _tof(int(g["wseed"]))
    net.eval()
    with torch.no_grad():
        y = net(x)
    assert relerr(y, g["out_eval"]) < 2e-4 and float((y.cpu() - torch.from_numpy(g["out_eval"])).abs().max()) < 1e-3
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.batch_neighbors = True                   # eval: one SpyNet pass over the 6 neighbours == the loop
    with torch.no_grad():
        assert relerr(net(x), y) < 1e-6
    net.batch_neighbors = False
    assert all(torch.equal(v, sd0[k]) for k, v in net.state_dict().items())      # eval mode leaves the buffers alone
    net.train()
    y = net(x)
    assert relerr(y, g["out_train"]) < 2e-4
    loss = hipops.charbonnier(y, tgt)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    by_name = dict(net.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    assert names == list(by_name.keys())
    bad = [(k, float(by_name[k].grad.norm()), b) for k, b in zip(names, g["grad_norms"])
           if abs(float(by_name[k].grad.norm()) - b) > 2e-3 * b + 1e-6]
    assert not bad, bad[:6]
    for key in g:
        if key.startswith("grad__"):
            name = key[len("grad__"):].replace("__", ".")
            assert relerr(by_name[name].grad, g[key]) < 1e-2, name
    sd = net.state_dict()
    assert relerr(sd["SpyNet.blocks.3.block.1.running_mean"], g["running_mean_b3_1"]) < 1e-5
    assert relerr(sd["SpyNet.blocks.3.block.1.running_var"], g["running_var_b3_1"]) < 1e-5
    assert int(sd["SpyNet.blocks.0.block.1.num_batches_tracked"]) == 6


@pytest.mark.parametrize("name", ["tof_44x40", "tof_20x30"])
def test_toflow_sizes_not_multiples_of_16_golden(name):
    """Sizes the drivers actually feed TOFlow (180x320 = a 45x80 SLR clip x4, Vid4's 144x180, 22x22 patches) are not
    multiples of 16: the pyramid floors and the H//16 x W//16 zero flow is resized to each level's own size
    (TOF_arch.py:69-90).  Goldens from the reference's module: eval forward, training forward, loss, all 80 gradient norms."""
    from dynavsr_amd import hipops
    g = load_golden(name)
    h, w = int(g["h"]), int(g["w"])
    x = synth.clip(int(g["xseed"]), 1, 7, h, w).cuda()
    tgt = synth.clip(int(g["tseed"]), 1, 1, h, w)[:, 0].cuda()
    net = _tof(int(g["wseed"]))
    net.eval()
    with torch.no_grad():
        y = net(x)
    assert relerr(y, g["out_eval"]) < 2e-4 and float((y.cpu() - torch.from_numpy(g["out_eval"])).abs().max()) < 1e-3
    net.train()
    y = net(x)
    assert relerr(y, g["out_train"]) < 2e-4
    loss = hipops.charbonnier(y, tgt)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    by_name = dict(net.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    bad = [(k, float(by_name[k].grad.norm()), b) for k, b in zip(names, g["grad_norms"])
           if abs(float(by_name[k].grad.norm()) - b) > 2e-3 * b + 1e-6]
    assert not bad, bad[:6]
    for key in g:
        if key.startswith("grad__"):
            name_ = key[len("grad__"):].replace("__", ".")
            assert relerr(by_name[name_].grad, g[key]) < 1e-2, name_
    with pytest.raises(RuntimeError, match="at least 16"):
        net(x[..., :12, :])


def test_toflow_batch2_vs_oracle_and_wrapper():
    """B = 2 at another size against oracle/tof.py (flows, warped stack, output), and the wrapper API with
    network_G.which_model_G = TOF (networks.py:37-39): feed_data / test() / calculate_loss / backward."""
    from oracle import tof as otof
    P = synth.tof_state_dict(5)
    x = synth.clip(21, 2, 7, 48, 32)
    taps = {}
    with torch.no_grad():
        ref = otof.toflow_forward(OrderedDict((k, v.clone()) for k, v in P.items()), x, training=False, taps=taps)
    net = _tof(5).eval()
    with torch.no_grad():
        y = net(x.cuda())
    assert relerr(y, ref) < 2e-4
    from dynavsr_amd.models import create_model
    from dynavsr_amd.options.options import dict_to_nonedict
    opt = dict_to_nonedict({"name": "tof", "model": "video_base", "scale": 4, "gpu_ids": [0], "dist": False, "is_train": False,
                            "network_G": {"which_model_G": "TOF"}, "path": {"strict_load": True},
                            "train": {"pixel_criterion": "cb", "pixel_weight": 1.0}})
    model = create_model(opt)
    model.netG.load_state_dict(P)
    model.feed_data({"LQs": x[:1], "GT": x[:1, 3]})
    model.test()
    assert relerr(model.fake_H, ref[:1]) < 2e-4
    loss = model.calculate_loss()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.netG.parameters())


def test_bicubic_align_corners_upsample():
    """F.interpolate(scale_factor=s, mode='bicubic', align_corners=True) -- what the drivers apply in front of TOFlow
    (test_dynavsr.py:188-193) -- forward and adjoint against torch in fp64, 5-D clips included."""
    from dynavsr_amd import tofops as T
    for shape, s in (((2, 3, 9, 13), 2), ((1, 7, 3, 8, 12), 4), ((1, 1, 1, 1), 3), ((2, 2, 5, 4), 4)):
        x = rnd(*shape, seed=1)
        h, w = shape[-2:]
        go = rnd(*(shape[:-2] + (h * s, w * s)), seed=2)
        xd = x.double().requires_grad_()
        y = F.interpolate(xd.reshape(-1, 1, h, w), scale_factor=s, mode="bicubic", align_corners=True).reshape(go.shape)
        (gr,) = torch.autograd.grad(y, xd, go.double())
        xg = x.cuda().requires_grad_()
        yg = T.upsample_bicubic_ac(xg, s)
        (gg,) = torch.autograd.grad(yg, xg, go.cuda())
        assert relerr(yg, y) < 2e-6 and relerr(gg, gr) < 2e-6, (shape, s)


def test_toflow_inner_step_through_adapt_frame():
    """The TOF branch of the per-frame adaptation (test_dynavsr.py:188-193, 245-250, 271-272): the SLR clip is brought to
    the LR size with the bicubic kernel before TOFlow, the L1 term uses the SLR clip itself; loss and adapted output
    against the same statements in plain torch on the CPU oracle."""
    from oracle import mfdn as omfdn, tof as otof, edvr as oedvr
    from dynavsr_amd.adapt import adapt_frame
    from dynavsr_amd.models import create_model
    from dynavsr_amd.options.options import dict_to_nonedict
    scale = 2
    opt = dict_to_nonedict({"name": "tof", "model": "video_base+lrimgestimator", "scale": scale, "gpu_ids": [0], "dist": False,
                            "is_train": False, "network_G": {"which_model_G": "TOF"},
                            "network_E": {"which_model_E": "MFDN", "mode": "video", "nf": 16, "in_nc": 3},
                            "datasets": {"train": {"kernel_size": 21, "patch_size": 128, "batch_size": 1}, "val": {"N_frames": 7}},
                            "path": {"strict_load": True},
                            "train": {"pixel_criterion": "cb", "pixel_weight": 1.0, "use_real": False, "loss_ftn": "l1",
                                      "maml": {"optimizer": "SGD", "lr_alpha": 1e-4, "adapt_iter": 1, "use_patch": False}}})
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    PG, PE, PF = synth.tof_state_dict(7), synth.mfdn_state_dict(2, nf=16, scale=scale), synth.mfdn_state_dict(3, nf=16, scale=scale)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE); est_fixed.netE.load_state_dict(PF)
    lqs = synth.clip(55, 1, 7, 64, 96)
    out = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, {"LQs": lqs.cuda()})
    # the same statements on the CPU
    G = OrderedDict((k, (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()))
                    for k, v in PG.items())
    E = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in PE.items())
    up = lambda c: F.interpolate(c.reshape(-1, 3, c.shape[-2], c.shape[-1]), scale_factor=scale, mode="bicubic",
                                 align_corners=True).reshape(c.shape[:3] + (c.shape[-2] * scale, c.shape[-1] * scale))
    with torch.no_grad():
        slr_fixed = omfdn.mfdn_forward(PF, lqs, scale)
    slr = omfdn.mfdn_forward(E, lqs, scale)
    loss = oedvr.charbonnier(otof.toflow_forward(G, up(slr), training=True), lqs[:, 3]) + 10 * F.l1_loss(slr, slr_fixed)
    params = [v for v in list(G.values()) + list(E.values()) if v.requires_grad]
    grads = torch.autograd.grad(loss, params)
    with torch.no_grad():
        for p, g_ in zip(params, grads):
            p -= 1e-4 * g_
        sr = otof.toflow_forward(G, up(lqs), training=False)
    assert abs(float(out["losses"][0]) - float(loss)) < 2e-5 * abs(float(loss))
    assert relerr(out["sr"], sr) < 2e-4
