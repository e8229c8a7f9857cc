"""GPU parity of the whole-network engine (dvsr_edvr_forward) against the CPU oracle and the
golden vectors produced by the imported reference.  north_star tolerance: outputs within 1e-3
relative; asserted here: rel-L2 <= 2e-4 and max-abs <= 1e-3 on O(1) activations."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, relerr
from dynavsr_amd import synth

pytestmark = pytest.mark.gpu


def make_net(seed=0, bf16_mfma=0, **cfg):
    from dynavsr_amd.models.archs.EDVR_arch import EDVR
    net = EDVR(bf16_mfma=bf16_mfma, **cfg)
    net.load_state_dict(synth.edvr_state_dict(seed, **{k: v for k, v in cfg.items()}), strict=True)
    return net.cuda()


@pytest.mark.parametrize("tag", ["16x16", "32x48"])
def test_edvr_forward_golden(tag):
    g = load_golden("edvr_" + tag)
    h, w = int(g["h"]), int(g["w"])
    net = make_net(int(g["wseed"]))
    x = synth.clip(int(g["xseed"]), 1, 5, h, w).cuda()
    with torch.no_grad():
        y = net(x)
    assert y.shape == g["out"].shape
    assert relerr(y, g["out"]) < 2e-4
    assert float((y.cpu() - torch.from_numpy(g["out"])).abs().max()) < 1e-3


def test_edvr_x2_forward_backward_golden():
    """EDVR-M x2 -- the configuration of the shipped x2 YAMLs (options/test/EDVR/EDVR_V.yml: nf 64, back_RBs 10,
    scale 2; EDVR_arch.py:244-245,303-304 drop upconv1) -- against the reference's output and gradients."""
    from dynavsr_amd import hipops
    g = load_golden("edvr_x2_24x32")
    h, w = int(g["h"]), int(g["w"])
    cfg = dict(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=2)
    net = make_net(int(g["wseed"]), **cfg)
    assert len(list(net.parameters())) == 142
    x = synth.clip(int(g["xseed"]), 1, 5, h, w).cuda()
    tgt = synth.clip(int(g["tseed"]), 1, 1, 2 * h, 2 * w)[:, 0].cuda()
    y = net(x)
    assert tuple(y.shape) == (1, 3, 2 * h, 2 * w)
    assert relerr(y, g["out"]) < 2e-4 and float((y.detach().cpu() - torch.from_numpy(g["out"])).abs().max()) < 1e-3
    loss = hipops.charbonnier(y, tgt)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    norms = np.array([float(p.grad.norm()) for p in net.ordered_parameters()])
    bad = [(n, a, b) for n, a, b in zip(net._names, norms, g["grad_norms"]) if abs(a - b) > 1e-3 * abs(b) + 1e-9]
    assert not bad, bad[:8]
    by_name = dict(zip(net._names, net.ordered_parameters()))
    for key in g:
        if key.startswith("grad__"):
            name = key[len("grad__"):].replace("__", ".")
            assert relerr(by_name[name].grad, g[key]) < 1e-2, name


def test_edvr_forward_layerwise_vs_oracle():
    """Every named intermediate of the engine against the oracle's taps (B=2 exercises batching)."""
    from oracle import edvr as oedvr
    P = synth.edvr_state_dict(4)
    x = synth.clip(11, 2, 5, 24, 40)
    taps = {}
    with torch.no_grad():
        ref = oedvr.edvr_forward(P, x, taps=taps)
    net = make_net(4)
    net._debug_ws = []
    with torch.no_grad():
        y = net(x.cuda())
    plan, ws = net._debug_ws[-1]
    b, n, c, h, w = 2, 5, 64, 24, 40
    report = []
    for name, shape, ref_t in [
        ("L1_fea", (b, n, c, h, w), taps["L1_fea"]),
        ("L2_fea", (b, n, c, h // 2, w // 2), taps["L2_fea"]),
        ("L3_fea", (b, n, c, h // 4, w // 4), taps["L3_fea"]),
        ("L3_offset", (b, n, c, h // 4, w // 4), torch.stack([taps["pcd%d_L3_offset" % i] for i in range(n)], 1)),
        ("L3_aligned", (b, n, c, h // 4, w // 4), torch.stack([taps["pcd%d_L3_fea" % i] for i in range(n)], 1)),
        ("L2_offset", (b, n, c, h // 2, w // 2), torch.stack([taps["pcd%d_L2_offset" % i] for i in range(n)], 1)),
        ("L2_aligned", (b, n, c, h // 2, w // 2), torch.stack([taps["pcd%d_L2_fea" % i] for i in range(n)], 1)),
        ("L1_offset", (b, n, c, h, w), torch.stack([taps["pcd%d_L1_offset" % i] for i in range(n)], 1)),
        ("L1_aligned", (b, n, c, h, w), torch.stack([taps["pcd%d_L1_fea" % i] for i in range(n)], 1)),
        ("aligned", (b, n, c, h, w), taps["aligned"]),
        ("tsa_cor", (b, n, h, w), taps["tsa_cor"]),
        ("tsa_gated", (b, n * c, h, w), taps["tsa_gated"]),
        ("tsa_att", (b, c, h, w), taps["tsa_att"]),
        ("tsa_out", (b, c, h, w), taps["tsa_out"]),
        ("recon", (b, c, h, w), taps["recon"]),
    ]:
        got = plan.tensor(ws, name, shape)
        e = relerr(got, ref_t)
        # kink flips: elements whose activation branch (the sign an (L)ReLU backward keys on) differs between the GPU's
        # fp32 summation order and the oracle's -- what makes per-tensor GRADIENT bars looser than the forward's
        flips = int(((got.cpu() > 0) != (ref_t > 0)).sum())
        report.append((name, e, flips, ref_t.numel()))
    print("\n".join("%-12s %.3e  sign flips %d / %d" % r for r in report))
    bad = [r for r in report if not r[1] < 2e-4]
    assert not bad, bad
    # the flips are counted, not assumed: a handful of near-zero elements per million (they sit within ~1e-6 of the kink)
    act_taps = [r for r in report if r[0] in ("L1_fea", "L2_fea", "L3_fea", "L3_offset", "L2_offset", "L1_offset", "aligned")]
    assert sum(r[2] for r in act_taps) <= 2e-5 * sum(r[3] for r in act_taps), [(r[0], r[2], r[3]) for r in act_taps]
    assert relerr(y, ref) < 2e-4


def test_edvr_state_dict_contract():
    """Reference checkpoints must load with strict=True: same 144 names/shapes, 'module.' optional."""
    from dynavsr_amd.models.archs.EDVR_arch import EDVR
    from dynavsr_amd.spec import edvr_param_spec
    net = EDVR(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=4)
    sd = net.state_dict()
    spec = edvr_param_spec()
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)


def test_edvr_rejects_bad_input():
    net = make_net(0)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        net(torch.zeros(1, 5, 3, 18, 16).cuda())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 5, 3, 16, 16))


@pytest.mark.parametrize("tag", ["16x16", "32x48"])
def test_edvr_backward_golden(tag):
    """d(charbonnier)/d(all 144 params) through dvsr_edvr_backward vs the reference's autograd."""
    from dynavsr_amd import hipops
    g = load_golden("edvr_" + tag)
    h, w = int(g["h"]), int(g["w"])
    net = make_net(int(g["wseed"]))
    x = synth.clip(int(g["xseed"]), 1, 5, h, w).cuda()
    tgt = synth.clip(int(g["tseed"]), 1, 1, 4 * h, 4 * w)[:, 0].cuda()
    loss = hipops.charbonnier(net(x), tgt)
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    params = net.ordered_parameters()
    norms = np.array([float(p.grad.norm()) for p in params])
    bad = [(n, a, b) for n, a, b in zip(net._names, norms, g["grad_norms"]) if abs(a - b) > 1e-3 * abs(b) + 1e-9]
    assert not bad, bad[:8]
    by_name = dict(zip(net._names, params))
    for key in g:
        if key.startswith("grad__"):
            name = key[len("grad__"):].replace("__", ".")
            # full tensors: the reference's own fp32-vs-fp64 spread on these gradients is ~2e-3
            # (ReLU / floor kinks, see test_edvr_backward_all_grads_vs_oracle), hence 1e-2 here
            assert relerr(by_name[name].grad, g[key]) < 1e-2, name


@pytest.mark.parametrize("mfma_mode", [0, 2])
def test_edvr_backward_all_grads_vs_oracle(mfma_mode):
    """Every gradient tensor (144 params AND the input clip), B=2, against the fp64 oracle.  mfma_mode 2 = the
    experimental 3-way bf16 split of the 3x3 convs (forward and data gradient), held to the same bars.

    EDVR's gradient is only piecewise smooth (ReLU/LeakyReLU signs, max-pool arg-max, floor() in
    the DCN sampler), so two correct fp32 evaluations differ by ~1e-3 in rel-L2 on small clips.
    The tolerance is therefore the envelope of the reference arithmetic itself: the HIP result
    must be as close to the fp64 oracle as the fp32 CPU oracle is (x3, floor 3e-4)."""
    from oracle import edvr as oedvr

    def cpu(dt):
        P = OrderedDict((k, v.to(dt).requires_grad_(True)) for k, v in synth.edvr_state_dict(5).items())
        x = synth.clip(21, 2, 5, 16, 24).to(dt).requires_grad_(True)
        go = torch.from_numpy(np.random.RandomState(3).standard_normal((2, 3, 64, 96))).to(dt)
        y = oedvr.edvr_forward(P, x)
        return y.detach(), [t.detach() for t in torch.autograd.grad(y, [x] + list(P.values()), go)], go

    y64, g64, go = cpu(torch.float64)
    y32, g32, _ = cpu(torch.float32)
    net = make_net(5, bf16_mfma=mfma_mode)
    xg = synth.clip(21, 2, 5, 16, 24).cuda().requires_grad_(True)
    yg = net(xg)
    yg.backward(go.float().cuda())
    assert relerr(yg, y64) < 2e-5
    names = ["x"] + net._names
    got = [xg.grad] + [p.grad for p in net.ordered_parameters()]
    e_gpu = np.array([relerr(a, b) for a, b in zip(got, g64)])
    e_cpu = np.array([relerr(a, b) for a, b in zip(g32, g64)])
    print("median rel-L2 vs fp64 oracle: HIP %.2e, CPU-fp32 oracle %.2e; worst HIP %.2e (%s)"
          % (np.median(e_gpu), np.median(e_cpu), e_gpu.max(), names[int(e_gpu.argmax())]))
    # A ReLU sign / max-pool arg-max / floor() flip that happens on one side only perturbs every
    # gradient upstream of it by ~1/sqrt(#elements) ~ 1e-3..5e-3, and it hits different tensors on
    # the two sides, so the envelope is asserted on the distribution, not per tensor.  Wiring bugs
    # give O(1) errors; exactness of each linear op is covered by tests/test_gpu_ops.py (2e-5).
    assert e_gpu.max() < 2e-2, (names[int(e_gpu.argmax())], e_gpu.max())
    assert np.median(e_gpu) < 3 * np.median(e_cpu) + 1e-4


def _gpu_opt(optimizer):
    import os
    from conftest import ROOT
    from dynavsr_amd.options import options as option
    opt = option.dict_to_nonedict(option.parse(os.path.join(
        ROOT, "dynavsr_amd", "options", "test", "EDVR", "EDVR_M_S4.yml"), is_train=False))
    opt["dist"] = False
    for k in ("pretrain_model_G", "pretrain_model_E"):
        opt["path"][k] = None
    opt["train"]["maml"]["optimizer"] = optimizer
    return opt


@pytest.mark.parametrize("optimizer", ["SGD", "Adam"])
def test_inner_step_golden(optimizer):
    """BASELINE.json configs[0] on the GPU: one inner MAML step through the wrapper API
    (create_model / feed_data / forward_without_optim / calculate_loss / test) vs the golden
    produced by the reference's wrappers on CPU: loss, adapted SR frame, parameter updates."""
    from dynavsr_amd.adapt import adapt_frame
    from dynavsr_amd.models import create_model
    g = load_golden("inner_step_" + optimizer.lower())
    opt = _gpu_opt(optimizer)
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    PG, PE, PEF = synth.edvr_state_dict(0), synth.mfdn_state_dict(0), synth.mfdn_state_dict(1)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE); est_fixed.netE.load_state_dict(PEF)
    lqs = synth.clip(1, 1, 5, 64, 64).cuda()
    r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, {"LQs": lqs})
    assert abs(float(r["losses"][0]) - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    assert relerr(r["slr"], g["slr"]) < 1e-5
    sr = r["sr"] if optimizer == "SGD" else r["sr"][..., 96:160, 96:160]
    assert relerr(sr, g["sr"]) < 2e-4
    newG, newE = dict(modelcp.netG.named_parameters()), dict(estcp.netE.named_parameters())
    tol = 1e-2 if optimizer == "SGD" else 1e-1          # Adam's first step is ~lr*sign(g)
    for key in g:
        if key.startswith("dG__") or key.startswith("dE__"):
            name = key[4:].replace("__", ".")
            src, new = (PG, newG) if key.startswith("dG__") else (PE, newE)
            delta = new[name].detach().cpu().double() - src[name].double()
            assert relerr(delta, g[key]) < tol, name
    # the meta-parameters themselves must be untouched (adaptation works on deep copies)
    assert torch.equal(model.netG.state_dict()["conv_first.weight"].cpu(), PG["conv_first.weight"])


@pytest.mark.parametrize("optimizer", ["SGD", "Adam"])
def test_adapt_frame_reuses_copies_like_a_fresh_deepcopy(optimizer):
    """test_dynavsr.py:208 deep-copies netG / netE and builds a new optimiser for EVERY frame.  adapt_frame
    refreshes the previous frame's copies in place and resets the native optimiser; the second frame must come
    out as if everything had been rebuilt (same clip -> same result), and the meta-parameters stay untouched."""
    from dynavsr_amd.adapt import adapt_frame
    from dynavsr_amd.models import create_model
    opt = _gpu_opt(optimizer)
    opt["train"]["maml"]["adapt_iter"] = 2
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    PG, PE = synth.edvr_state_dict(0), synth.mfdn_state_dict(0)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE); est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    a, b = {"LQs": synth.clip(1, 1, 5, 32, 48).cuda()}, {"LQs": synth.clip(2, 1, 5, 32, 48).cuda()}
    r1 = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, a)
    sr1, l1 = r1["sr"].clone(), [float(v) for v in r1["losses"]]
    g_first, e_first = modelcp.netG, estcp.netE
    adapt_frame(opt, model, est, modelcp, estcp, est_fixed, b)          # another clip in between
    r3 = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, a)
    assert modelcp.netG is g_first and estcp.netE is e_first            # refreshed in place, not re-created
    from copy import deepcopy
    modelcp.netG, estcp.netE = deepcopy(model.netG), deepcopy(est.netE)  # a caller swaps the copies: the cached
    r4 = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, a)      # optimiser must not be reused for them
    assert relerr(r4["sr"], sr1) < 1e-5
    assert all(abs(x - float(y)) <= 1e-5 * abs(x) for x, y in zip(l1, r3["losses"]))
    assert relerr(r3["sr"], sr1) < 1e-5
    for k, v in model.netG.state_dict().items():
        assert torch.equal(v.cpu(), PG[k]), k
    for k, v in est.netE.state_dict().items():
        assert torch.equal(v.cpu(), PE[k]), k


def test_inner_step_x2_wrapper_path_vs_oracle():
    """The x2 family (options/test/EDVR/EDVR_M.yml = the geometry of the reference's EDVR_V.yml: EDVR-M x2 + MFDN
    x2): one inner step through create_model / adapt_frame against the functional oracle."""
    import os
    from conftest import ROOT
    from dynavsr_amd.adapt import adapt_frame
    from dynavsr_amd.models import create_model
    from dynavsr_amd.options import options as option
    from oracle import inner as oinner
    opt = option.dict_to_nonedict(option.parse(os.path.join(ROOT, "dynavsr_amd", "options", "test", "EDVR", "EDVR_M.yml"),
                                               is_train=False))
    opt["dist"] = False
    for k in ("pretrain_model_G", "pretrain_model_E"):
        opt["path"][k] = None
    opt["train"]["maml"]["optimizer"] = "SGD"
    assert opt["scale"] == 2
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    cfg = dict(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=2)
    PG, PE, PEF = synth.edvr_state_dict(0, **cfg), synth.mfdn_state_dict(0, scale=2), synth.mfdn_state_dict(1, scale=2)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE); est_fixed.netE.load_state_dict(PEF)
    lqs = synth.clip(1, 1, 5, 32, 48)
    r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, {"LQs": lqs.cuda()})
    assert tuple(r["slr"].shape) == (1, 5, 3, 16, 24) and tuple(r["sr"].shape) == (1, 3, 64, 96)
    losses, _, _, sro = oinner.inner_adapt(PG, PE, PEF, lqs, 1, "SGD", 1e-5, (0.9, 0.99), scale=2)
    assert abs(float(r["losses"][0]) - losses[0]) < 2e-5 * abs(losses[0])
    assert relerr(r["sr"], sro) < 2e-4


def test_inner_step_x2_sfdn_image_mode_vs_oracle():
    """EDVR-M x2 + SFDN (network_E: which_model_E SFDN, mode image -- the single-frame estimator of
    options/train/MAML/EDVR/EDVR_REDS_SFDN.yml, frames folded into the batch by LRestimator_model.feed_data :99-101):
    one inner SGD step through the wrappers, loss and adapted output against the functional oracle."""
    import os
    import torch.nn.functional as F
    from conftest import ROOT
    from dynavsr_amd.adapt import adapt_frame
    from dynavsr_amd.models import create_model
    from dynavsr_amd.options import options as option
    from oracle import edvr as oedvr, mfdn as omfdn
    opt = option.dict_to_nonedict(option.parse(os.path.join(ROOT, "dynavsr_amd", "options", "test", "EDVR", "EDVR_M.yml"),
                                               is_train=False))
    opt["dist"] = False
    for k in ("pretrain_model_G", "pretrain_model_E"):
        opt["path"][k] = None
    opt["train"]["maml"]["optimizer"] = "SGD"
    opt["network_E"] = option.dict_to_nonedict({"which_model_E": "SFDN", "mode": "image", "nf": 64})
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    cfg = dict(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=2)
    PG, PE, PEF = synth.edvr_state_dict(0, **cfg), synth.sfdn_state_dict(0), synth.sfdn_state_dict(1)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE); est_fixed.netE.load_state_dict(PEF)
    lqs = synth.clip(3, 1, 5, 32, 48)
    lr = opt["train"]["maml"]["lr_alpha"]
    r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, {"LQs": lqs.cuda()})
    # oracle: the same step with plain autograd
    OG = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in PG.items())
    OE = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in PE.items())
    slr = omfdn.sfdn_forward(OE, lqs[0]).unsqueeze(0)                       # [1,5,3,16,24]
    with torch.no_grad():
        slr_fixed = omfdn.sfdn_forward(PEF, lqs[0]).unsqueeze(0)
    loss = oedvr.charbonnier(oedvr.edvr_forward(OG, slr, scale=2), lqs[:, 2]) + 10 * F.l1_loss(slr, slr_fixed)
    grads = torch.autograd.grad(loss, list(OG.values()) + list(OE.values()))
    with torch.no_grad():
        for p, g in zip(list(OG.values()) + list(OE.values()), grads):
            p -= lr * g
        sro = oedvr.edvr_forward(OG, lqs, scale=2)
    assert relerr(r["slr"], slr) < 1e-5
    assert abs(float(r["losses"][0]) - float(loss.detach())) < 2e-5 * abs(float(loss.detach()))
    assert relerr(r["sr"], sro) < 2e-4


def test_inner_step_use_real_vs_oracle():
    """train.use_real = True (test_dynavsr.py:242-243, 225-226): the SLR clip comes from the data ('SuperLQs'), the
    estimator is neither run nor optimised; one SGD step of netG against plain autograd on the oracle."""
    import torch.nn.functional as F
    from dynavsr_amd.adapt import adapt_frame
    from dynavsr_amd.models import create_model
    from oracle import edvr as oedvr, mfdn as omfdn
    opt = _gpu_opt("SGD")
    opt["train"]["use_real"] = True
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    PG, PE, PEF = synth.edvr_state_dict(0), synth.mfdn_state_dict(0), synth.mfdn_state_dict(1)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE); est_fixed.netE.load_state_dict(PEF)
    lqs, slr = synth.clip(1, 1, 5, 64, 64), synth.clip(2, 1, 5, 16, 16)
    r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, {"LQs": lqs.cuda(), "SuperLQs": slr.cuda()})
    OG = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in PG.items())
    with torch.no_grad():
        slr_fixed = omfdn.mfdn_forward(PEF, lqs)
    loss = oedvr.charbonnier(oedvr.edvr_forward(OG, slr), lqs[:, 2]) + 10 * F.l1_loss(slr, slr_fixed)
    grads = torch.autograd.grad(loss, list(OG.values()))
    with torch.no_grad():
        for p, g in zip(OG.values(), grads):
            p -= opt["train"]["maml"]["lr_alpha"] * g
        sro = oedvr.edvr_forward(OG, lqs)
    assert abs(float(r["losses"][0]) - float(loss.detach())) < 2e-5 * abs(float(loss.detach()))
    assert relerr(r["sr"], sro) < 2e-4
    for k, v in estcp.netE.state_dict().items():
        assert torch.equal(v.cpu(), PE[k]), k            # the estimator copy was not stepped


@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_super_resolve_video_clips_in_flight_equal_one_at_a_time(in_flight):
    """adapt.super_resolve_video keeps `in_flight` clips on the GPU, one HIP stream each: every result must be bit-identical
    to the plain `net(clip)` (test_dynavsr.py:200-204), in order, for more clips than streams, clips of two sizes (a plan per
    stream and size) and CPU-resident clips; an empty stream of clips yields nothing."""
    from dynavsr_amd.adapt import super_resolve_video
    opt = _gpu_opt("Adam")
    net = make_net(0)
    clips = [synth.clip(60 + i, 1, 5, *((32, 48) if i != 3 else (24, 40))) for i in range(7)]
    clips = [c.cuda() if i % 3 else c for i, c in enumerate(clips)]
    with torch.no_grad():
        want = [net(c.cuda()).clone() for c in clips]
    got = []
    for sr in super_resolve_video(opt, net, ({"LQs": c} if i % 2 else c for i, c in enumerate(clips)), in_flight=in_flight):
        got.append(sr.clone())                        # (a yielded frame is only valid until the generator moves on)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert list(super_resolve_video(opt, net, [], in_flight=in_flight)) == []
    assert net.training


def test_adapt_video_overlap_equals_sequential_loop():
    """adapt_video runs the next clip's baseline forward on a second stream underneath the current clip's
    adaptation; the per-clip results must be those of the plain loop (baseline test() + adapt_frame)."""
    from dynavsr_amd.adapt import adapt_frame, adapt_video
    from dynavsr_amd.models import create_model
    opt = _gpu_opt("Adam")
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    clips = [{"LQs": synth.clip(40 + i, 1, 5, 32, 48).cuda()} for i in range(4)]
    want = []
    for c in clips:
        model.feed_data(c, need_GT=False); model.test()
        base = model.fake_H.clone()
        want.append((base, adapt_frame(opt, model, est, modelcp, estcp, est_fixed, c)["sr"].clone()))
    got = [(a.clone(), r["sr"].clone()) for a, r in adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips)]
    assert len(got) == len(want)
    for (a, b), (c, d) in zip(got, want):
        assert torch.equal(a, c)                       # forward only: bit-identical
        assert relerr(b, d) < 1e-5                     # through a backward with fp32 atomics in the weight gradients
    assert list(adapt_video(opt, model, est, modelcp, estcp, est_fixed, [])) == []


def test_adapt_video_transient_clips_are_not_reused_under_the_side_streams():
    """The DataLoader case: clips arrive on the CPU (adapt_video copies them to the GPU itself) or as GPU
    temporaries nobody else holds.  The side-stream forwards read the clip until the end of their tape, so the
    caching allocator must not hand its block out on the main stream meanwhile: the consumer below allocates and
    overwrites same-sized blocks between the yields, and the results must still be those of the plain loop."""
    from dynavsr_amd.adapt import adapt_frame, adapt_video
    from dynavsr_amd.models import create_model
    opt = _gpu_opt("Adam")
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    host = [synth.clip(60 + i, 1, 5, 32, 48) for i in range(5)]
    want = []
    for c in host:
        d = {"LQs": c.cuda()}
        model.feed_data(d, need_GT=False); model.test()
        want.append((model.fake_H.clone(), adapt_frame(opt, model, est, modelcp, estcp, est_fixed, d)["sr"].clone()))
    for make in (lambda c: {"LQs": c},                          # CPU clips
                 lambda c: {"LQs": c.cuda()}):                  # GPU temporaries owned by the generator only
        got = []
        for a, r in adapt_video(opt, model, est, modelcp, estcp, est_fixed, (make(c) for c in host)):
            got.append((a.clone(), r["sr"].clone()))
            junk = [torch.full_like(host[0], float(k + 7), device="cuda") for k in range(6)]   # same-size blocks
            del junk
        assert len(got) == len(want)
        for (a, b), (c, d) in zip(got, want):
            assert torch.equal(a, c)
            assert relerr(b, d) < 1e-5


def test_use_patch_crop_and_inner_step():
    """train.maml.use_patch (test_dynavsr.py:118-145,255-260): adapt.crop against the crops the reference's common_crop
    produced (golden; bit-exact, it is a gather), its adjoint against autograd of the slicing form (overlapping
    patches add up), and one inner step with patches against the oracle driven by the same `random` seed."""
    import random
    from dynavsr_amd.adapt import adapt_frame, crop
    from dynavsr_amd.models import create_model
    from oracle import inner as oin
    g = load_golden("crop_patches")
    t, c, h, w, s, n, psz = (int(g[k]) for k in ("t", "c", "h", "w", "scale", "n", "patch_size"))
    r = np.random.RandomState(int(g["dseed"]))
    seq = torch.from_numpy(r.rand(1, t, c, h, w).astype(np.float32))
    hr = torch.from_numpy(r.rand(1, c, s * h, s * w).astype(np.float32))
    random.seed(int(g["seed"]))
    sg = seq.cuda().requires_grad_()
    lr_p, hr_p = crop(sg, hr.cuda(), n, psz)
    assert torch.equal(lr_p.cpu(), torch.from_numpy(g["lr"]))
    assert np.allclose(hr_p.double().sum(dim=(1, 2, 3)).cpu().numpy(), g["hr_sum"], rtol=0, atol=1e-9)
    go = torch.randn(lr_p.shape, generator=torch.Generator().manual_seed(2))
    (gs,) = torch.autograd.grad(lr_p, sg, go.cuda())
    sd = seq.clone().requires_grad_()
    random.seed(int(g["seed"]))
    o_lr, _ = oin.crop(sd, hr, n, psz)
    (gr,) = torch.autograd.grad(o_lr, sd, go)
    assert relerr(gs, gr) < 1e-6
    with pytest.raises(RuntimeError, match="leaves the"):
        from dynavsr_amd import hipops
        hipops.patch_gather(sg[0], [h], [0], psz // 2, 1)
    # one inner step on patches (LR 64x64 -> SLR 16x16, 3 patches of 8x8) vs the oracle
    opt = _gpu_opt("SGD")
    opt["train"]["maml"].update({"use_patch": True, "num_patch": 3, "patch_size": 16})
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    PG, PE, PF = synth.edvr_state_dict(0), synth.mfdn_state_dict(0), synth.mfdn_state_dict(1)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE); est_fixed.netE.load_state_dict(PF)
    lqs = synth.clip(77, 1, 5, 64, 64)
    random.seed(5)
    out = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, {"LQs": lqs.cuda()})
    random.seed(5)
    losses, _PGa, _PEa, sr = oin.inner_adapt(PG, PE, PF, lqs, steps=1, optimizer="SGD", lr=opt["train"]["maml"]["lr_alpha"],
                                            use_patch=True, num_patch=3, patch_size=16)
    assert abs(float(out["losses"][0]) - losses[0]) < 1e-5 * abs(losses[0])
    assert relerr(out["sr"], sr) < 2e-4


def test_meta_gradient_allreduce_over_rccl():
    """dist.py on its real backend: a one-rank `nccl` (= RCCL) process group with the collective forced, so the
    branch bench.py's `meta_step` leg and an N-rank job take is executed on the GPU here as well -- the flat-buffer
    pack, ONE all-reduce, the average, the unpack; and meta_train_step drives it (train_dynavsr.py:438)."""
    import socket
    import torch.distributed as tdist
    from dynavsr_amd import dist as D
    from dynavsr_amd.adapt import meta_train_step
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                             device_id=torch.device("cuda", 0))
    try:
        a, b = torch.nn.Linear(5, 3).cuda(), torch.nn.Conv2d(2, 4, 3).cuda()
        for i, p in enumerate(list(a.parameters()) + [b.weight]):
            p.grad = torch.full_like(p, float(i + 1))
        want = [p.grad.clone() for p in a.parameters()] + [b.weight.grad.clone()]
        nbytes = D.allreduce_meta_gradients([a, b], average=True, force=True)
        assert nbytes == 4 * sum(p.numel() for m in (a, b) for p in m.parameters())
        got = [p.grad for p in a.parameters()] + [b.weight.grad]
        assert all(torch.equal(g, w) for g, w in zip(got, want))          # mean over one rank
        assert float(b.bias.grad.abs().max()) == 0.0                       # a missing .grad enters as zeros
        opt, model, est, modelcp, estcp, params, data, PG, PE = _meta_setup(adapt_iter=1)
        optimizer = torch.optim.SGD(params, lr=1e-3)
        r0 = meta_train_step(opt, model, est, modelcp, estcp, data, optimizer, force_collective=True)
        g0 = [p.grad.clone() for p in params]
        model.netG.load_state_dict(PG); est.netE.load_state_dict(PE)
        tdist.destroy_process_group()
        r1 = meta_train_step(opt, model, est, modelcp, estcp, data, optimizer)   # no process group: no collective
        assert abs(r0["loss_q"] - r1["loss_q"]) < 1e-6 * abs(r1["loss_q"])
        assert max(relerr(a_, b_) for a_, b_ in zip(g0, [p.grad for p in params])) < 1e-4
    finally:
        if tdist.is_initialized():
            tdist.destroy_process_group()


def test_engine_refuses_stale_weights_and_second_backward():
    """The tape keeps detached aliases of the parameters and releases its arena after one backward: an in-place
    update between forward and backward, or a second backward, must raise instead of returning wrong numbers."""
    from dynavsr_amd import hipops
    net = make_net(0)
    x = synth.clip(5, 1, 5, 16, 16).cuda()
    y = net(x)
    with torch.no_grad():
        next(net.parameters()).add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        y.sum().backward()
    y = net(x)
    y.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        y.sum().backward()
    with pytest.raises(RuntimeError, match="differ in shape"):
        hipops.charbonnier(y.detach(), y.detach()[:, :2])


def _meta_setup(adapt_iter=2):
    from dynavsr_amd.models import create_model
    opt = _gpu_opt("Adam")
    opt["train"]["maml"]["adapt_iter"] = adapt_iter
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    PG, PE = synth.edvr_state_dict(0), synth.mfdn_state_dict(0)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE)
    params = [p for p in model.netG.parameters() if p.requires_grad] + [p for p in est.netE.parameters() if p.requires_grad]
    data = {"LQs": synth.clip(31, 2, 5, 32, 32).cuda(), "SuperLQs": synth.clip(32, 2, 5, 8, 8).cuda(),
            "GT": synth.clip(33, 2, 5, 128, 128).cuda()}
    return opt, model, est, modelcp, estcp, params, data, PG, PE


@pytest.mark.parametrize("batched", [True, False])
def test_meta_train_step_golden(batched):
    """One outer iteration of train_dynavsr.py:265-438 (B = 2 tasks, adapt_iter = 2, inner Adam, meta SGD) against
    the golden produced by driving the reference's wrappers through the same statements on CPU: loss_q, the inner
    losses, every meta-gradient norm, five full gradient tensors, and the SGD update of the meta-parameters -- for the
    task loop and for all tasks as one batch (every loss of the shipped driver is taken at the same weights, quirk Q1)."""
    from dynavsr_amd.adapt import meta_train_step
    g = load_golden("meta_step")
    opt, model, est, modelcp, estcp, params, data, PG, PE = _meta_setup()
    lr_G = float(g["lr_G"])
    optimizer = torch.optim.SGD(params, lr=lr_G)
    r = meta_train_step(opt, model, est, modelcp, estcp, data, optimizer, inner="reference", batched=batched)
    assert bool(r.get("batched")) == batched
    assert abs(r["loss_q"] - float(g["loss_q"])) < 2e-5 * abs(float(g["loss_q"]))
    lt = np.array([float(v) for v in r["loss_train"]])
    assert np.abs(lt - g["loss_train"]).max() < 2e-5 * np.abs(g["loss_train"]).max()
    gn = np.array([float(p.grad.norm()) for p in model.netG.parameters()])
    ge = np.array([float(p.grad.norm()) for p in est.netE.parameters()])
    assert np.abs(gn - g["gradG_norms"]).max() < 2e-3 * g["gradG_norms"].max() and np.allclose(gn, g["gradG_norms"], rtol=2e-2, atol=1e-6)
    assert np.allclose(ge, g["gradE_norms"], rtol=2e-2, atol=1e-6)
    byG, byE = dict(model.netG.named_parameters()), dict(est.netE.named_parameters())
    for key in g:
        if key.startswith("gG__") or key.startswith("gE__"):
            name = key[4:].replace("__", ".")
            p = (byG if key.startswith("gG__") else byE)[name]
            assert relerr(p.grad, g[key]) < 1e-2, name          # kink-flip envelope, see test_edvr_backward_golden
            src = (PG if key.startswith("gG__") else PE)[name]
            assert relerr(p.detach().cpu().double() - src.double(), -lr_G * torch.from_numpy(g[key]).double()) < 1e-2


def test_meta_train_step_first_order_maml_mode():
    """inner='copies': the copies are really adapted (the meta-parameters are not touched by the inner steps), loss_q
    is taken at the adapted weights and only its first-order gradient reaches the meta-parameters."""
    from dynavsr_amd.adapt import meta_train_step
    opt, model, est, modelcp, estcp, params, data, PG, PE = _meta_setup(adapt_iter=1)
    opt["train"]["maml"]["lr_alpha"] = 1e-4
    optimizer = torch.optim.SGD(params, lr=0.0)                 # lr 0: the meta step must leave the parameters alone
    r = meta_train_step(opt, model, est, modelcp, estcp, data, optimizer, inner="copies")
    gb = [p.grad.detach().clone() for p in list(model.netG.parameters()) + list(est.netE.parameters())]
    for k, v in model.netG.state_dict().items():
        assert torch.equal(v.cpu(), PG[k]), k                   # inner steps ran on the copies only
    moved = sum(float((a.detach() - b.detach()).abs().max()) > 0 for a, b in zip(modelcp.netG.parameters(), model.netG.parameters()))
    assert moved > 100                                          # ... and the copies did move (Adam, lr_alpha)
    # meta-gradient of netG = sum over tasks of d loss_q(adapted) / B only (no inner-loss terms as in 'reference')
    ref = meta_train_step(opt, model, est, modelcp, estcp, data, torch.optim.SGD(params, lr=0.0), inner="reference")
    gq = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.netG.parameters())))
    assert np.isfinite(gq) and gq > 0 and np.isfinite(r["loss_q"])
    assert r["loss_q"] != ref["loss_q"]                         # loss_q was taken at the ADAPTED weights
    # the task loop and the batched form (FrameBatch + per-task weight sets) are the same computation
    assert r.get("batched")
    r2 = meta_train_step(opt, model, est, modelcp, estcp, data, torch.optim.SGD(params, lr=0.0), inner="copies", batched=False)
    assert not r2.get("batched") and abs(r2["loss_q"] - r["loss_q"]) < 1e-5 * abs(r["loss_q"])
    assert all(abs(float(a) - float(b)) < 1e-5 * abs(float(b)) for a, b in zip(r["loss_train"], r2["loss_train"]))
    gl = [p.grad for p in list(model.netG.parameters()) + list(est.netE.parameters())]
    na, nb = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gb))), float(torch.sqrt(sum((g.double() ** 2).sum() for g in gl)))
    assert abs(na - nb) < 2e-3 * nb
    worst = max(relerr(a, b) for a, b in zip(gb, gl) if float(b.norm()) > 1e-8 * nb)
    assert worst < 2e-2, worst


def test_dcn_dropin_module_matches_engine_and_oracle():
    """Op-level drop-in (models/archs/dcn) forward+backward vs the C oracle."""
    from dynavsr_amd.models.archs.dcn import ModulatedDeformConvPack
    from oracle import edvr as oedvr
    torch.manual_seed(0)
    m = ModulatedDeformConvPack(64, 64, 3, stride=1, padding=1, dilation=1, deformable_groups=8,
                                extra_offset_mask=True).cuda()
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.05)
        m.conv_offset_mask.bias.normal_(0, 0.05)
    x = torch.randn(2, 64, 12, 20).cuda().requires_grad_()
    f = torch.randn(2, 64, 12, 20).cuda().requires_grad_()
    y = m([x, f])
    go = torch.randn_like(y)
    y.backward(go)
    P = {"d." + k: v.detach().cpu().requires_grad_() for k, v in m.named_parameters()}
    xc, fc = x.detach().cpu().requires_grad_(), f.detach().cpu().requires_grad_()
    yr = oedvr.dcn_pack(P, "d", xc, fc, 8)
    yr.backward(go.cpu())
    assert relerr(y, yr) < 2e-5
    assert relerr(x.grad, xc.grad) < 1e-4 and relerr(f.grad, fc.grad) < 1e-4
    for k, v in m.named_parameters():
        assert relerr(v.grad, P["d." + k].grad) < 1e-4, k


def test_edvr_L_forward_backward_vs_oracle():
    """BASELINE.json configs[4] architecture (EDVR-L: nf=128, 7 frames, 40 reconstruction blocks, x4)
    in fp32: exercises C/dg = 16 (generic DCN kernel), 128/256-channel convs, 2 cout blocks."""
    from oracle import edvr as oedvr
    cfg = dict(nf=128, nframes=7, groups=8, front_RBs=5, back_RBs=40, scale=4)
    P = OrderedDict((k, v.requires_grad_(True)) for k, v in synth.edvr_state_dict(7, **cfg).items())
    x = synth.clip(31, 1, 7, 16, 16)
    go = torch.from_numpy(np.random.RandomState(4).standard_normal((1, 3, 64, 64)).astype(np.float32))
    y = oedvr.edvr_forward(P, x, **cfg)
    ref = torch.autograd.grad(y, list(P.values()), go)
    net = make_net(7, **cfg)
    assert len(net._names) == 264
    yg = net(x.cuda())
    yg.backward(go.cuda())
    assert relerr(yg, y) < 2e-4
    errs = np.array([relerr(p.grad, r) for p, r in zip(net.ordered_parameters(), ref)])
    assert errs.max() < 3e-2 and np.median(errs) < 5e-3, (errs.max(), np.median(errs))


def test_config3_psnr_parity_synthetic_video():
    """BASELINE.json configs[2] substitute (no Vid4 / checkpoints offline): a seeded
    synthetic clip sequence, 3 inner MAML steps per frame (SGD, larger lr so that adaptation moves
    the output), PSNR-vs-GT of the adapted frame through this build and through the CPU oracle:
    |delta PSNR| <= 0.02 dB per frame (north_star), with the reference's uint8 PSNR definition."""
    from dynavsr_amd.adapt import adapt_frame
    from dynavsr_amd.models import create_model
    from dynavsr_amd.utils import util
    from oracle import inner as oinner
    opt = _gpu_opt("SGD")
    opt["train"]["maml"]["adapt_iter"] = 3
    opt["train"]["maml"]["lr_alpha"] = 1e-3
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    PG, PE, PEF = synth.edvr_state_dict(0), synth.mfdn_state_dict(0), synth.mfdn_state_dict(1)
    model.netG.load_state_dict(PG); est.netE.load_state_dict(PE); est_fixed.netE.load_state_dict(PEF)
    video = synth.clip(77, 1, 7, 48, 64)                       # 7 LR frames -> 3 sliding windows of 5
    gt = synth.clip(78, 1, 7, 192, 256)                        # synthetic HR "ground truth"
    for t in range(3):
        lqs = video[:, t:t + 5].contiguous()
        r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, {"LQs": lqs.cuda()})
        _, _, _, sr_ref = oinner.inner_adapt(PG, PE, PEF, lqs, 3, "SGD", 1e-3)
        hr = util.tensor2img(gt[0, t + 2], mode="rgb")
        p_gpu = util.calculate_psnr(util.tensor2img(r["sr"][0], mode="rgb"), hr)
        p_ref = util.calculate_psnr(oinner.tensor2img_rgb(sr_ref[0]), hr)
        assert abs(p_gpu - p_ref) <= 0.02, (t, p_gpu, p_ref)
        assert relerr(r["sr"], sr_ref) < 1e-3


def test_edvr_bf16_mfma_path():
    """BASELINE configs[4]: the 3x3 convolutions on the bf16 MFMA (operands rounded to bf16, fp32 accumulate,
    fp32 storage).  Not a parity configuration: the bound is bf16's (2^-8 per operand), gated the way the
    north_star gates reduced precision -- PSNR of the output against the fp32 oracle's own output far above
    the 0.02 dB sensitivity (>= 50 dB means < 1e-4 dB change of a 30 dB PSNR) -- and gradients stay close."""
    from oracle import edvr as oedvr
    P = synth.edvr_state_dict(4)
    x = synth.clip(11, 1, 5, 32, 48)
    from dynavsr_amd.models.archs.EDVR_arch import EDVR
    net = EDVR(bf16_mfma=True)
    net.load_state_dict(P, strict=True)
    net = net.cuda()
    ref = EDVR()
    ref.load_state_dict(P, strict=True)
    ref = ref.cuda()
    xg = x.cuda()
    y = net(xg)
    with torch.no_grad():
        y32 = ref(xg)
        yo = oedvr.edvr_forward(P, x)
    assert relerr(y32, yo) < 2e-4
    e = relerr(y, yo)
    assert 1e-5 < e < 2e-2, e          # really a different arithmetic, and within bf16's reach
    mse = float(((y.detach().cpu() - yo) ** 2).mean())
    assert 10 * np.log10(1.0 / mse) > 50.0
    # backward runs (bf16 data gradient, fp32 weight gradient) and stays close to the fp32 engine
    go = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).cuda()   # (unseeded, the 5e-2 bound below was hit once in ~10 runs)
    y.backward(go)
    ref(xg).backward(go)
    gn = lambda m: float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
    assert abs(gn(net) - gn(ref)) / gn(ref) < 8e-2
    errs = sorted(relerr(a.grad, b.grad) for a, b in zip(net.parameters(), ref.parameters()))
    # (a random-weight 40-layer network amplifies the 4e-3 per-layer operand rounding; measured median 0.12)
    assert errs[len(errs) // 2] < 0.25 and errs[-1] < 0.6, (errs[len(errs) // 2], errs[-1])


def test_edvr_split_bf16_mfma_path():
    """bf16_mfma = 2 (experimental, off by default): every fp32 operand of the 3x3 convolutions is split into
    three bf16 pieces (8+8+8 mantissa bits) and the six products that matter go through the bf16 MFMA with
    fp32 accumulation.  Unlike mode 1 this IS held to the parity bars of the fp32 path: the output AND the
    residual branch alone (output minus the bilinear base, which otherwise hides half of the error) against
    the oracle; gradients: test_edvr_backward_all_grads_vs_oracle[2]."""
    import torch.nn.functional as F
    from oracle import edvr as oedvr
    P = synth.edvr_state_dict(4)
    x = synth.clip(11, 1, 5, 32, 48)
    net = make_net(4, bf16_mfma=2)
    ref = make_net(4)
    xg = x.cuda()
    with torch.no_grad():
        y, y32 = net(xg).cpu(), ref(xg).cpu()
        yo = oedvr.edvr_forward(P, x)
    base = F.interpolate(x[:, 2], scale_factor=4, mode="bilinear", align_corners=False)
    assert relerr(y, yo) < 2e-5
    assert float((y - yo).abs().max()) <= 1e-4
    e2, e32 = relerr(y - base, yo - base), relerr(y32 - base, yo - base)
    assert e2 < 2e-5 and e2 < 3 * e32 + 1e-6, (e2, e32)   # measured 1.9e-6 for both


def test_config2_full_size_forward_parity():
    """BASELINE configs[1] at its full size (1x5x3x180x320 -> 3x720x1280): SURVEY 8d's parity line, output vs the
    CPU oracle max-abs <= 1e-3 on [0,1] data and PSNR(GPU, CPU) >= 60 dB.  (The oracle forward takes a few
    seconds on the GPU box's host cores.)"""
    from oracle import edvr as oedvr
    P = synth.edvr_state_dict(0)
    x = synth.clip(1, 1, 5, 180, 320, smooth=False)
    net = make_net(0)
    with torch.no_grad():
        y = net(x.cuda()).cpu()
        yo = oedvr.edvr_forward(P, x)
    assert y.shape == (1, 3, 720, 1280)
    d = (y - yo).abs()
    assert float(d.max()) <= 1e-3, float(d.max())
    assert relerr(y, yo) < 2e-4
    assert 10 * np.log10(1.0 / float(((y - yo) ** 2).mean())) >= 60.0


def test_inner_step_size_forward_backward_vs_oracle():
    """EDVR on the inner-step clip of the north_star (SLR 44x80 of an LR 176x320 clip): loss and the gradient
    w.r.t. the input clip (which feeds the estimator's backward) against the oracle, and the parameter
    gradients by norm (isolated ReLU / floor kink flips allow ~1e-3 per tensor)."""
    from oracle import edvr as oedvr
    from dynavsr_amd import hipops
    P = synth.edvr_state_dict(0)
    x = synth.clip(3, 1, 5, 44, 80)
    tgt = synth.clip(4, 1, 1, 176, 320)[:, 0]
    net = make_net(0)
    xg = x.cuda().requires_grad_(True)
    loss = hipops.charbonnier(net(xg), tgt.cuda())
    loss.backward()
    PO = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in P.items())
    xo = x.clone().requires_grad_(True)
    lo = oedvr.charbonnier(oedvr.edvr_forward(PO, xo), tgt)
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) / float(lo.detach()) < 1e-5
    assert relerr(xg.grad, xo.grad) < 5e-3
    gn = lambda ps: float(torch.sqrt(sum((p.grad.detach().double().cpu() ** 2).sum() for p in ps)))
    assert abs(gn(net.parameters()) - gn(PO.values())) / gn(PO.values()) < 5e-3


def test_winograd_path_equals_direct_path_forward_and_backward(monkeypatch):
    """The same clip through the tape with the large 3x3 convolutions on the Winograd kernel (default cost model: every
    96x128 level-1 layer of this 5-frame clip takes it, forward and data gradient) and with DVSR_CONV_WINO=0 (direct
    kernels): output, loss and every parameter gradient agree to fp32 round-off -- the two are different algorithms for
    the same sums, kink flips bound the gradients."""
    from dynavsr_amd import engine, hipops
    x = synth.clip(5, 1, 5, 96, 128)
    tgt = synth.clip(6, 1, 1, 384, 512)[:, 0]
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DVSR_CONV_WINO", mode)
        engine._plans.clear()          # geometry is chosen when a plan is built
        net = make_net(0)
        plan = engine.get_plan(net._cfg(), 1, 96, 128)
        tags = [nm for (_k, nm, _f, _b) in plan.op_info()]
        nw = sum(1 for t in tags if t.endswith("w]") or t.endswith("w3]") or t.endswith("w5]"))
        assert (nw > 20) if mode == "1" else (nw == 0), (mode, nw)
        xg = x.cuda().requires_grad_(True)
        y = net(xg)
        loss = hipops.charbonnier(y, tgt.cuda())
        loss.backward()
        res[mode] = (y.detach().cpu(), float(loss.detach()), xg.grad.cpu(), [p.grad.detach().cpu() for p in net.parameters()])
    engine._plans.clear()
    (y1, l1, gx1, g1), (y0, l0, gx0, g0) = res["1"], res["0"]
    assert relerr(y1, y0) < 5e-6 and abs(l1 - l0) / abs(l0) < 1e-6
    assert relerr(gx1, gx0) < 5e-3
    gn = lambda gs: float(torch.sqrt(sum((g.double() ** 2).sum() for g in gs)))
    assert abs(gn(g1) - gn(g0)) / gn(g0) < 5e-3
    worst = max(relerr(a, b) for a, b in zip(g1, g0))
    assert worst < 5e-2, worst


# ---- BASELINE.json configs[4]: EDVR-L x4 (nf 128, 7 frames, 40 blocks) on 1x7x3x64x64 (256x256 HR tiles) -----------
EDVR_L = dict(nf=128, nframes=7, groups=8, front_RBs=5, back_RBs=40, scale=4)
_edvr_l_cache = {}


def _edvr_l_oracle():
    """Full output + every parameter gradient of the fp32 CPU oracle (pinned against the reference module by
    oracle/gen_golden.py edvr_l), computed once per session: ~1.1 TFLOP on the host cores."""
    if "o" not in _edvr_l_cache:
        from oracle import edvr as oedvr
        g = load_golden("edvr_l_64x64")
        P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in synth.edvr_state_dict(int(g["wseed"]), **EDVR_L).items())
        x = synth.clip(int(g["xseed"]), 1, 7, 64, 64)
        tgt = synth.clip(int(g["tseed"]), 1, 1, 256, 256)[:, 0]
        y = oedvr.edvr_forward(P, x, **EDVR_L)
        loss = oedvr.charbonnier(y, tgt)
        grads = torch.autograd.grad(loss, list(P.values()))
        _edvr_l_cache["o"] = (g, x, tgt, y.detach(), float(loss.detach()), OrderedDict(zip(P.keys(), grads)))
    return _edvr_l_cache["o"]


@pytest.mark.parametrize("mode", [0, 2, 1])
def test_edvr_l_config4_forward_backward(mode):
    """The workload BASELINE configs[4] names, forward + backward of the Charbonnier loss, on the three MFMA paths:
    mode 0 (exact fp32 MFMA) and mode 2 (3-way bf16 split, fp32-accurate) are held to the fp32 parity bars against
    the reference's golden (sub-sampled output, loss, all 264 gradient norms, five full gradient tensors) and
    against the full oracle tensors; mode 1 (operands rounded to bf16) to a stated bf16 bound."""
    from dynavsr_amd import hipops
    g, x, tgt, yo, loss_o, grads_o = _edvr_l_oracle()
    # the oracle on this box reproduces the golden made from the imported reference in the build container
    assert relerr(yo[:, :, ::4, ::4], g["out_sub"]) < 1e-5 and abs(loss_o - float(g["loss"])) < 1e-5 * float(g["loss"])
    net = make_net(int(g["wseed"]), bf16_mfma=mode, **EDVR_L)
    assert len(list(net.parameters())) == len(g["grad_norms"]) == 264
    y = net(x.cuda())
    assert tuple(y.shape) == (1, 3, 256, 256)
    loss = hipops.charbonnier(y, tgt.cuda())
    loss.backward()
    yc = y.detach().cpu()
    norms = np.array([float(p.grad.norm()) for p in net.ordered_parameters()])
    ref_norms = g["grad_norms"]
    by_name = dict(zip(net._names, net.ordered_parameters()))
    full = {k[len("grad__"):].replace("__", "."): g[k] for k in g if k.startswith("grad__")}
    if mode in (0, 2):
        assert relerr(yc[:, :, ::4, ::4], g["out_sub"]) < 2e-4
        assert relerr(yc, yo) < 2e-4 and float((yc - yo).abs().max()) < 1e-3
        assert abs(float(yc.double().norm()) - float(g["out_norm"])) < 1e-4 * float(g["out_norm"])
        assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * float(g["loss"])
        # gradients: per-tensor norms within 2e-3 of the reference's (a 95-conv-deep net, ReLU / floor kink flips:
        # DESIGN 3.3), the full tensors within 1e-2
        bad = [(n, a, b) for n, a, b in zip(net._names, norms, ref_norms) if abs(a - b) > 2e-3 * abs(b) + 1e-9]
        assert not bad, (len(bad), bad[:6])
        for name, ref in full.items():
            assert relerr(by_name[name].grad, ref) < 1e-2, name
        errs = sorted(relerr(by_name[k].grad, v) for k, v in grads_o.items())
        assert errs[len(errs) // 2] < 2e-3 and errs[-1] < 2e-2, (errs[len(errs) // 2], errs[-1])
    else:
        # bf16 operands (2^-9 relative rounding per operand) through ~95 convolutions with random weights: stated
        # bound rel-L2 <= 3e-2 on the output, PSNR vs the fp32 result >= 35 dB, loss within 2 % (measured 0.84 %),
        # gradient norms: median within 5 %, none off by more than 50 %; the arithmetic must actually differ from
        # fp32 (>= 1e-5)
        e = relerr(yc, yo)
        psnr = 10 * np.log10(1.0 / float(((yc - yo) ** 2).mean()))
        rel = np.abs(norms - ref_norms) / (np.abs(ref_norms) + 1e-12)
        print("EDVR-L bf16 mode 1: out rel-L2 %.2e, PSNR %.1f dB, loss %.5f vs %.5f, grad-norm rel median %.2e max %.2e"
              % (e, psnr, float(loss.detach()), loss_o, np.median(rel), rel.max()))
        assert 1e-5 < e < 3e-2, e
        assert psnr > 35.0
        assert abs(float(loss.detach()) - loss_o) < 2e-2 * loss_o
        assert np.median(rel) < 0.05 and rel.max() < 0.5, (np.median(rel), rel.max())


def test_edvr_l_bf16_psnr_gate_in_north_star_terms():
    """BASELINE configs[4] / north_star: reduced precision is gated by PSNR-vs-ground-truth within 0.02 dB of the
    reference arithmetic.  No checkpoints or REDS tiles exist on the box, so -- as SURVEY 8d prescribes for configs[2] --
    the substitute is synthetic, and since round 5 it sits at a MEANINGFUL operating point (the r04 review: at 8.28 dB a
    0.03 dB difference says little): synth.sr_pair -- a smooth HR clip, its area-downscaled LR clip, the HR centre frame as
    ground truth; bilinear up-sampling alone scores 30 dB -- and EDVR-L x4 (nf 128, 7 frames, 40 blocks) with its residual
    branch damped to the size of a trained network's correction (synth.damp_residual_branch).  PSNR of the exact-fp32 path
    (mode 0, itself held to the reference's golden), the plain bf16-operand path (mode 1) and the three-way bf16 split
    (mode 2) against the same GT with the reference's uint8 definition (utils/util.py:262-269 on tensor2img).
    Both reduced modes must hold the gate here; the round-3 substitute (unrelated noise images, 8 dB) stays as the stress
    case below, where mode 1's deviation is only bounded (0.030 dB measured)."""
    from dynavsr_amd.utils import util
    lr, gt = synth.sr_pair(91, 7, 64, 64)
    hr = util.tensor2img(gt, mode="rgb")
    psnr = {}
    for mode in (0, 1, 2):
        net = EDVR_l_net(mode, damped=True)
        with torch.no_grad():
            sr = net(lr.cuda())
        psnr[mode] = util.calculate_psnr(util.tensor2img(sr[0], mode="rgb"), hr)
        del net
    print("EDVR-L PSNR vs synthetic GT (sr_pair, damped residual): fp32 %.4f dB, bf16 operands %.4f dB, bf16 split-3 %.4f dB"
          % (psnr[0], psnr[1], psnr[2]))
    assert psnr[0] >= 25.0, psnr                          # the operating point means something
    assert abs(psnr[2] - psnr[0]) <= 0.02, psnr          # the gate
    assert abs(psnr[1] - psnr[0]) <= 0.02, psnr          # ... also for the plain bf16-operand path at this operating point
    # stress case of rounds 3-4: random-init residual branch against an unrelated target (8 dB)
    x = synth.clip(91, 1, 7, 64, 64).cuda()
    hr = util.tensor2img(synth.clip(92, 1, 1, 256, 256)[0, 0], mode="rgb")
    for mode in (0, 1, 2):
        net = EDVR_l_net(mode, damped=False)
        with torch.no_grad():
            psnr[mode] = util.calculate_psnr(util.tensor2img(net(x)[0], mode="rgb"), hr)
        del net
    assert abs(psnr[2] - psnr[0]) <= 0.02, psnr
    assert abs(psnr[1] - psnr[0]) <= 0.1, psnr           # stated bound of the opt-in path there (measured 0.030 dB)


def EDVR_l_net(mode, damped, seed=8, gain=0.02):
    from dynavsr_amd.models.archs.EDVR_arch import EDVR
    net = EDVR(bf16_mfma=mode, **EDVR_L)
    sd = synth.edvr_state_dict(seed, **EDVR_L)
    net.load_state_dict(synth.damp_residual_branch(sd, gain) if damped else sd, strict=True)
    return net.cuda()


def test_edvr_l_bf16_gate_sweep_over_gain_and_seed():
    """The gate of the test above at ONE gain of the damped residual branch (0.02), one seed and one clip says how the reduced
    modes behave at that operating point -- not that they hold it everywhere (r05 review / ADVICE: every trunk perturbation
    reaches the output through conv_last, so the PSNR delta scales with the gain by construction).  The sweep: gain in
    {0.01, 0.02, 0.05, 0.1, 1.0} x weight seeds {8, 9, 10}.  What is asserted:
      * mode 2 (exact 3-way split) holds 0.02 dB at EVERY point -- it is the fp32 arithmetic;
      * mode 1 (plain bf16 operands, opt-in) is characterised, not blessed: its deviation normalised by the gain -- the
        rel-L2 of the residual branch's output against mode 0's, which does not depend on the gain -- stays below 3e-2, and
        the PSNR delta is reported per point together with the largest gain at which all three seeds pass 0.02 dB
        (profiles/r06_edvr_l_gate_sweep.txt has the table; bench.py reports `gate_holds_up_to_gain`)."""
    from dynavsr_amd.utils import util
    lr, gt = synth.sr_pair(91, 7, 64, 64)
    hr = util.tensor2img(gt, mode="rgb")
    base = F.interpolate(lr[0, 3:4], scale_factor=4, mode="bilinear", align_corners=False)[0].cuda()
    rows = []
    for seed in (8, 9, 10):
        for gain in (0.01, 0.02, 0.05, 0.1, 1.0):
            out = {}
            for mode in (0, 1, 2):
                net = EDVR_l_net(mode, damped=True, seed=seed, gain=gain)
                with torch.no_grad():
                    out[mode] = net(lr.cuda())[0]
                del net
            ps = {m: util.calculate_psnr(util.tensor2img(out[m], mode="rgb"), hr) for m in out}
            branch = {m: (out[m] - base) for m in out}              # the residual branch's contribution (gain x trunk output)
            nrm = {m: float((branch[m] - branch[0]).norm() / branch[0].norm()) for m in (1, 2)}
            rows.append((seed, gain, ps[0], ps[1] - ps[0], ps[2] - ps[0], nrm[1], nrm[2]))
    print("seed gain  psnr_fp32  d_bf16  d_split3  rel_branch_bf16  rel_branch_split3")
    for r in rows:
        print("%4d %5.2f %9.4f %+8.4f %+8.5f %12.3e %12.3e" % r)
    assert all(abs(r[4]) <= 0.02 for r in rows), rows                 # mode 2: the gate, everywhere
    assert all(r[6] < 2e-5 for r in rows), rows
    assert all(r[5] < 3e-2 for r in rows), rows                       # mode 1: bounded in gain-independent terms
    assert all(abs(r[3]) <= 0.02 for r in rows if r[1] <= 0.02), rows  # ... and inside the gate at the gains <= the r05 point


# ---- per-clip parameter gradients: K clips through one tape (dvsr_edvr_plan_create_grouped) -----------------------------
def _stack(params, k):
    return [p.detach().unsqueeze(0).repeat((k,) + (1,) * p.dim()).contiguous().requires_grad_() for p in params]


@pytest.mark.parametrize("k,h,w", [(3, 16, 24), (2, 44, 80), (2, 96, 128)])   # (96x128: DMA-halo / Winograd / FAST wgrad kernels)
def test_edvr_stacked_tape_gives_per_clip_gradients(k, h, w):
    """EdvrStackedFunction: K clips as one batch, parameters stacked [K, ...] with equal slices; slice k of every gradient
    must be what a B = 1 pass over clip k alone produces (same kernels; launch geometry and atomic order may differ:
    1e-3 per tensor, outputs 1e-6), the per-sample Charbonnier losses bit-equal to the scalar kernel's."""
    from dynavsr_amd import engine, hipops
    net = make_net(0)
    x = synth.clip(61, k, 5, h, w).cuda().requires_grad_()
    tgt = synth.clip(62, k, 1, 4 * h, 4 * w)[:, 0].cuda()
    want = []
    for i in range(k):
        xi = x[i:i + 1].detach().requires_grad_()
        yi = net(xi)
        li = hipops.charbonnier(yi, tgt[i:i + 1])
        gs = torch.autograd.grad(li, [xi] + net.ordered_parameters())
        want.append((yi.detach(), li.detach(), gs))
    stacked = _stack(net.ordered_parameters(), k)
    y = engine.EdvrStackedFunction.apply(x, net._cfg(), False, *stacked)
    losses = hipops.charbonnier_per_sample(y, tgt)
    losses.sum().backward()
    for i in range(k):
        assert relerr(y[i:i + 1], want[i][0]) < 1e-6
        assert abs(float(losses[i]) - float(want[i][1])) <= 1e-6 * abs(float(want[i][1]))
        assert relerr(x.grad[i:i + 1], want[i][2][0]) < 1e-2      # through every ReLU / max-pool / floor() kink of the net
        bad = [(n, relerr(s.grad[i], g)) for n, s, g in zip(net._names, stacked, want[i][2][1:]) if relerr(s.grad[i], g) > 2e-3]
        assert not bad, (i, bad[:6])
    # scalar and per-sample loss kernels reduce a sample identically
    assert torch.equal(hipops.charbonnier_per_sample(y.detach()[:1], tgt[:1])[0], hipops.charbonnier(y.detach()[:1], tgt[:1]))
    a, b = torch.randn(k, 3, 5, 11, 13, device="cuda"), torch.randn(k, 3, 5, 11, 13, device="cuda")
    base = torch.rand(k, device="cuda")
    ps = hipops.inner_loss_per_sample(base, a, b, 10.0)
    for i in range(k):
        assert torch.equal(ps[i], hipops.inner_loss(base[i], a[i], b[i], 10.0))
    with pytest.raises(RuntimeError, match="K="):
        engine.EdvrStackedFunction.apply(x, net._cfg(), False, *[s[:1] for s in stacked])


@pytest.mark.parametrize("k,h,w", [(3, 16, 24), (2, 44, 80), (2, 96, 128)])   # (96x128: the level-1 layers on the Winograd kernel)
def test_edvr_stacked_tape_per_slice_weights(k, h, w):
    """dvsr_edvr_plan_create_ex with weight_sets = K: clip k is convolved with ITS OWN copy of the weights (the private
    copies of K frames after they have diverged: later inner steps, adapted forwards).  Every slice is a differently
    perturbed network; output, d/dx and slice k of every parameter gradient must equal a B = 1 pass of clip k through a
    network holding slice k -- forward, data gradients and weight gradients all index their weights by the clip."""
    from dynavsr_amd import engine, hipops
    net = make_net(0)
    base = [p.detach().clone() for p in net.ordered_parameters()]
    g = torch.Generator(device="cuda").manual_seed(5)
    stacked = []
    for p in base:
        s_ = p.unsqueeze(0).repeat((k,) + (1,) * p.dim())
        s_ = s_ * (1.0 + 0.05 * torch.randn(s_.shape, device="cuda", generator=g))
        stacked.append(s_.contiguous().requires_grad_())
    x = synth.clip(63, k, 5, h, w).cuda().requires_grad_()
    tgt = synth.clip(64, k, 1, 4 * h, 4 * w)[:, 0].cuda()
    y = engine.EdvrStackedFunction.apply(x, net._cfg(), True, *stacked)
    losses = hipops.charbonnier_per_sample(y, tgt)
    losses.sum().backward()
    for i in range(k):
        with torch.no_grad():
            for p, s_ in zip(net.ordered_parameters(), stacked):
                p.copy_(s_[i])
        xi = x[i:i + 1].detach().requires_grad_()
        yi = net(xi)
        gs = torch.autograd.grad(hipops.charbonnier(yi, tgt[i:i + 1]), [xi] + net.ordered_parameters())
        assert relerr(y[i:i + 1], yi) < 1e-6, i
        assert relerr(x.grad[i:i + 1], gs[0]) < 1e-2
        bad = [(n, relerr(s_.grad[i], g_)) for n, s_, g_ in zip(net._names, stacked, gs[1:]) if relerr(s_.grad[i], g_) > 2e-3]
        assert not bad, (i, bad[:6])
    # forward only (the adapted forwards of a chunk): no gradient workspace
    with torch.no_grad():
        y2 = engine.EdvrStackedFunction.apply(x.detach(), net._cfg(), True, *[s_.detach() for s_ in stacked])
    # (bit-identical unless the no-grad forward takes the F(4x4, 3x3) kernel on some layer -- tag "w5" -- where the training
    # tape keeps F(2x2): then the same sums by another algorithm, fp32 round-off apart)
    plan = engine.get_plan(net._cfg(), k, h, w, grad_groups=k, weight_sets=k)
    if any(nm.endswith("w5]") for (_k, nm, _f, _b) in plan.op_info()):
        assert relerr(y2, y.detach()) < 5e-6
    else:
        assert torch.equal(y2, y.detach())


def test_no_grad_forward_of_a_trainable_network_takes_the_no_grad_arena_and_geometries():
    """`with torch.no_grad(): net(x)` -- Video_base_model.test(), the baseline / adapted forwards of test_dynavsr.py --
    on a network whose parameters require grad: ctx.needs_input_grad still says True there, so the tape function reads the
    caller's grad mode instead.  The forward then allocates the activation arena only (the engine reads "no-grad" off the
    workspace size) and runs the no-grad launch geometries: at 180x320 that is the F(4x4, 3x3) kernel (tag "w5") on the
    large layers, and the result is that of dvsr_edvr_forward_timed on a no-grad arena -- bit for bit -- and of the
    recording forward (F(2x2) kernels) to fp32 round-off."""
    from dynavsr_amd import engine
    net = make_net(0)
    assert all(p.requires_grad for p in net.parameters())
    x = synth.clip(77, 1, 5, 180, 320).cuda()
    net._debug_ws = []
    with torch.no_grad():
        y = net(x)
    plan, ws = net._debug_ws[-1]
    assert ws.numel() == plan.workspace_bytes(False) < plan.workspace_bytes(True)
    assert not y.requires_grad
    tags = [nm for (_k, nm, _f, _b) in plan.op_info()]
    assert sum(t.endswith("w5]") for t in tags) >= 10, tags
    out = torch.empty_like(y)
    ws2 = torch.empty(plan.workspace_bytes(False), dtype=torch.uint8, device="cuda")
    plan.forward_timed([p.detach() for p in net.ordered_parameters()], x, out, ws2)
    assert torch.equal(out, y)
    y_rec = net(x)                                     # grad mode on: the recording tape, full workspace
    assert y_rec.requires_grad and net._debug_ws[-1][1].numel() == plan.workspace_bytes(True)
    assert relerr(y_rec.detach(), y) < 5e-6
    net.requires_grad_(False)                          # nothing to record either: frozen weights, data input
    z = net(x)
    assert net._debug_ws[-1][1].numel() == plan.workspace_bytes(False) and torch.equal(z, y)


def test_frozen_weights_scope_packs_once_per_stream_and_repacks_after_an_update(monkeypatch):
    """engine.FrozenWeights / dvsr_edvr_forward_packed (the clips of a video through one network, super_resolve_video): the
    first no-grad forward of a plan on a stream packs the weights, the following ones skip the packing launches and give the
    bits of a forward outside the scope; an in-place update of a weight (an optimiser step, load_state_dict) bumps the
    version counter and the next forward packs again; a recording forward inside the scope is untouched by it."""
    from dynavsr_amd import engine
    from dynavsr_amd.adapt import super_resolve_video
    net = make_net(3)
    xs = [synth.clip(80 + i, 1, 5, 64, 96).cuda() for i in range(4)]
    with torch.no_grad():
        want = [net(x).clone() for x in xs]
    calls = []
    real = engine.Plan.forward

    def spy(self, params, x, out, ws, packed=False):
        calls.append(bool(packed))
        return real(self, params, x, out, ws, packed=packed)

    monkeypatch.setattr(engine.Plan, "forward", spy)
    frozen = engine.FrozenWeights()
    with torch.no_grad(), frozen:
        got = [net(x).clone() for x in xs]
    assert calls == [False, True, True, True]
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    with frozen:                                         # grad mode on: the recording forward takes its own workspace
        y = net(xs[0])
    assert calls[-1] is False and y.requires_grad and relerr(y.detach(), want[0]) < 5e-6   # (the tape's own geometries)
    del calls[:]
    with torch.no_grad():
        net.conv_first.weight.mul_(1.25)                 # in place: the version counter moves
        want2 = net(xs[1]).clone()
        with frozen:
            got2 = [net(xs[1]).clone(), net(xs[1]).clone()]
    assert calls == [False, False, True]
    assert torch.equal(got2[0], want2) and torch.equal(got2[1], want2) and not torch.equal(want2, want[1])
    # the generator: two clips in flight = two streams, each packs once
    del calls[:]
    out = [y_.clone() for y_ in super_resolve_video({"network_G": {"which_model_G": "EDVR"}}, net, [xs[i % 4] for i in range(6)],
                                                     in_flight=2)]
    assert calls == [False, False, True, True, True, True]
    with torch.no_grad():
        assert all(torch.equal(out[i], net(xs[i % 4])) for i in range(6))


@pytest.mark.parametrize("optimizer,overlap", [("Adam", True), ("SGD", False)])
def test_adapt_video_batched_frames_equal_the_per_frame_loop(optimizer, overlap):
    """adapt_video(frames_per_batch=K): the inner steps of K consecutive frames as ONE batch with per-frame parameter
    gradients (FrameBatch) must give every frame what the reference's per-frame loop gives it (test_dynavsr.py:208-283,
    adapt_iter = 1): baseline SR bit-equal, loss, SLR clip, adapted SR and the adapted copies' weights.  5 clips with
    K = 2 -> chunks of 2, 2, 1; a clip of another size closes a chunk early."""
    from dynavsr_amd.adapt import FrameBatch, adapt_frame, adapt_video
    from dynavsr_amd.models import create_model
    opt = _gpu_opt(optimizer)
    assert opt["train"]["maml"]["adapt_iter"] == 1
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    assert FrameBatch.supported(opt, model, est)
    sizes = [(32, 48)] * 3 + [(48, 32)] + [(32, 48)]            # chunks: [0,1] [2] (size change) [3] (size change) [4]
    clips = [{"LQs": synth.clip(50 + i, 1, 5, h, w).cuda()} for i, (h, w) in enumerate(sizes)]
    want = []
    for c in clips:
        model.feed_data(c, need_GT=False); model.test()
        base = model.fake_H.clone()
        r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, c)
        want.append((base, r["sr"].clone(), float(r["losses"][0]), r["slr"].clone(),
                     [p.detach().clone() for p in modelcp.netG.ordered_parameters()],
                     [p.detach().clone() for p in estcp.netE.ordered_parameters()]))
    PG = {k: v.clone() for k, v in model.netG.state_dict().items()}
    n = 0
    for (base, r), w_ in zip(adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips, overlap=overlap,
                                         frames_per_batch=2), want):
        assert relerr(base, w_[0]) < 1e-6               # (the chunk's baselines are one batched forward: launch geometry may differ)
        assert abs(float(r["losses"][0]) - w_[2]) <= 2e-6 * abs(w_[2])
        assert relerr(r["slr"], w_[3]) < 1e-6
        assert relerr(r["sr"], w_[1]) < 1e-4
        # the frame's adapted copies: the update is lr * sign-like for Adam (a flipped tiny gradient moves an element by 2 lr)
        for name, p, q, p0 in zip(model.netG._names, modelcp.netG.ordered_parameters(), w_[4], model.netG.ordered_parameters()):
            assert relerr(p - p0, q - p0) < (2e-2 if optimizer == "Adam" else 2e-3), name
        for p, q, p0 in zip(estcp.netE.ordered_parameters(), w_[5], est.netE.ordered_parameters()):
            assert relerr(p - p0, q - p0) < (2e-2 if optimizer == "Adam" else 2e-3)
        assert set(modelcp.netG.state_dict().keys()) == set(PG.keys())
        n += 1
    assert n == len(clips)
    for k, v in model.netG.state_dict().items():                 # the meta-parameters are untouched
        assert torch.equal(v, PG[k]), k
    # more than one inner step (BASELINE configs[2] takes 3): after the first step the K copies have diverged and the
    # batch runs with per-frame weight sets (dvsr_*_plan_create_ex) -- same results as the per-frame loop
    opt["train"]["maml"]["adapt_iter"] = 3
    assert FrameBatch.supported(opt, model, est)
    want = []
    for c in clips[:2]:
        r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, c)
        want.append((r["sr"].clone(), [float(v) for v in r["losses"]]))
    got = list(adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips[:2], overlap=overlap, frames_per_batch=2))
    assert len(got) == 2
    for (base, r), (sr_w, l_w) in zip(got, want):
        assert len(r["losses"]) == 3
        assert all(abs(float(a) - b) <= 1e-5 * abs(b) for a, b in zip(r["losses"], l_w)), (r["losses"], l_w)
        assert relerr(r["sr"], sr_w) < 2e-4
    # what the batched step does not cover takes the per-frame loop
    opt["train"]["maml"]["use_patch"] = True
    assert not FrameBatch.supported(opt, model, est)
    opt["train"]["maml"]["use_patch"] = False


def test_adapt_video_batched_at_the_north_star_size():
    """The batched pipeline at the size bench.py measures it on (LR 176x320, SLR 44x80; four frames per batch -> 20-image
    launches: the large-grid kernels -- DMA-halo, Winograd, FAST weight gradient, per-frame weight sets in the adapted
    forwards) against the per-frame loop: same baselines, losses, SLR clips and adapted frames."""
    from dynavsr_amd.adapt import adapt_frame, adapt_video
    from dynavsr_amd.models import create_model
    opt = _gpu_opt("Adam")
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    clips = [{"LQs": synth.clip(70 + i, 1, 5, 176, 320).cuda()} for i in range(4)]
    want = []
    for c in clips:
        model.feed_data(c, need_GT=False); model.test()
        base = model.fake_H.clone()
        r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, c)
        want.append((base.cpu(), r["sr"].cpu(), float(r["losses"][0]), r["slr"].cpu()))
    got = list(adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips, frames_per_batch=4))
    assert len(got) == 4
    for (base, r), w_ in zip(got, want):
        # (the batch's baseline forwards are 4-clip launches: the cost model may put a layer on F(4x4) that the one-clip
        # forward runs on F(2x2) -- the same sums by another algorithm, fp32 round-off apart)
        assert relerr(base.cpu(), w_[0]) < 5e-6
        assert abs(float(r["losses"][0]) - w_[2]) <= 2e-6 * abs(w_[2])
        assert relerr(r["slr"].cpu(), w_[3]) < 1e-6
        assert relerr(r["sr"].cpu(), w_[1]) < 1e-4
    # three inner steps (BASELINE configs[2]): steps 2 and 3 and the adapted forwards run on per-frame weight sets
    opt["train"]["maml"]["adapt_iter"] = 3
    want3 = []
    for c in clips[:2]:
        r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, c)
        want3.append((r["sr"].cpu(), [float(v) for v in r["losses"]]))
    got3 = list(adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips[:2], frames_per_batch=2))
    for (base, r), (sr_w, l_w) in zip(got3, want3):
        assert all(abs(float(a) - b) <= 1e-5 * abs(b) for a, b in zip(r["losses"], l_w)), (r["losses"], l_w)
        assert relerr(r["sr"].cpu(), sr_w) < 2e-4


def test_validate_video_psnr_vectors_match_the_host_definition():
    """adapt.validate_video (train_dynavsr.py:500-728 on one rank): PSNR of the un-adapted and of the adapted frame against
    GT, computed on the device, must be the reference's uint8 PSNR (utils/util.py:262-269 on tensor2img) of the very
    frames adapt_video yields; vectors have one entry per frame, in frame order."""
    from dynavsr_amd.adapt import adapt_video, validate_video
    from dynavsr_amd.models import create_model
    from dynavsr_amd.utils import util
    opt = _gpu_opt("Adam")
    model, est = create_model(opt)
    modelcp, estcp = create_model(opt)
    _, est_fixed = create_model(opt)
    model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
    est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
    clips = [{"LQs": synth.clip(80 + i, 1, 5, 32, 48).cuda()} for i in range(3)]
    gts = [synth.clip(90 + i, 1, 1, 128, 192)[0, 0].cuda() for i in range(3)]
    r = validate_video(opt, model, est, modelcp, estcp, est_fixed, clips, gts, frames_per_batch=2)
    assert r["frames"] == [0, 1, 2] and r["psnr_start"].shape == (3,) and r["psnr_final"].dtype == torch.float64
    for i, (base, res) in enumerate(adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips, frames_per_batch=2)):
        hr = util.tensor2img(gts[i], mode="rgb")
        assert abs(float(r["psnr_start"][i]) - util.calculate_psnr(util.tensor2img(base[0], mode="rgb"), hr)) < 1e-9   # same call path
        assert abs(float(r["psnr_final"][i]) - util.calculate_psnr(util.tensor2img(res["sr"][0], mode="rgb"), hr)) < 2e-3


# ---- gradient parity that kink flips cannot loosen (r04) -------------------------------------------------------------
def _kink_free_gradient_check(cfg, seed, b, h, w, xseed, tseed):
    """dvsr_edvr_backward against the fp64 oracle evaluated ON THE GPU'S OWN ACTIVATIONS (oracle/edvr_tape.py: every launch
    output of the GPU forward is teacher-forced into the oracle, (L)ReLU backwards key on the sign of the GPU's stored
    outputs, pools / the deformable sampler decide on the GPU's values): what is compared is the exact linearisation of the
    forward the GPU ran, so only summation-order round-off separates the two -- every parameter gradient is held to 1e-4
    rel-L2 (observed worst: 1.8e-5, EDVR-M and EDVR-L, Winograd and direct kernels alike) and the input gradient, which
    has passed through every layer's data gradient and the deformable sampler's fixed-point window sums, to 5e-4 (observed
    4e-5 .. 1.2e-4) -- where the unforced comparisons need 1e-2 .. 5e-2 for the flips."""
    from dynavsr_amd import hipops
    from oracle import edvr as oedvr
    from oracle import edvr_tape
    net = make_net(seed, **cfg)
    net._debug_ws = []
    s = cfg.get("scale", 4)
    x = synth.clip(xseed, b, cfg.get("nframes", 5), h, w)
    tgt = synth.clip(tseed, b, 1, s * h, s * w)[:, 0]
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    plan, ws = net._debug_ws[-1]
    names = [nm.split("[")[0] for (_k, nm, _f, _b) in plan.op_info()]
    two = ("tsa_gate", "tsa_pool1", "tsa_pool2")
    outs = []
    for i, nm in enumerate(names):     # read back BEFORE the backward runs over the workspace
        o0 = plan.op_output(ws, i, 0)
        o1 = plan.op_output(ws, i, 1) if nm in two else None
        outs.append((None if o0 is None else o0.detach().cpu().clone(), None if o1 is None else o1.detach().cpu().clone()))
    y_cpu = y.detach().cpu()
    loss = hipops.charbonnier(y, tgt.cuda())
    loss.backward()

    def force(i, name, which, v):
        assert names[i] == name, (i, names[i], name)
        f = outs[i][which]
        if f is None:
            assert name == "conv_last"
            f = y_cpu
        assert f.numel() == v.numel(), (name, f.numel(), tuple(v.shape))
        return f.view(v.shape)
    P = synth.edvr_state_dict(seed, **cfg)
    Pd = OrderedDict((k, v.double().clone().requires_grad_(True)) for k, v in P.items())
    xd = x.double().clone().requires_grad_(True)
    yo, onames = edvr_tape.edvr_forward_tape(Pd, xd, force=force, **cfg)
    assert onames == names
    lo = oedvr.charbonnier(yo, tgt.double())
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-6 * abs(float(lo.detach()))
    worst = ("", 0.0)
    bad = []
    for name, p in zip(net._names, net.ordered_parameters()):
        e = relerr(p.grad, Pd[name].grad)
        if e > worst[1]:
            worst = (name, e)
        if not e < 1e-4:
            bad.append((name, e))
    ex = relerr(xg.grad, xd.grad)
    print("kink-free gradient check %s B=%d %dx%d: worst parameter gradient %s %.2e, input gradient %.2e"
          % (cfg or "EDVR-M", b, h, w, worst[0], worst[1], ex))
    assert not bad, bad[:8]
    assert ex < 5e-4, ex


@pytest.mark.parametrize("wino", ["1", "0"])
@pytest.mark.parametrize("b,h,w", [(1, 32, 48), (2, 44, 80)])
def test_edvr_backward_kink_free_all_144_gradients(b, h, w, wino, monkeypatch):
    """EDVR-M x4, every one of the 144 parameter gradients and the input gradient, on the default tape (the large 3x3 layers of
    the 44x80 batch on the Winograd kernels) and on the direct kernels (DVSR_CONV_WINO=0)."""
    monkeypatch.setenv("DVSR_CONV_WINO", wino)
    _kink_free_gradient_check({}, 0, b, h, w, 21, 22)


def test_edvr_l_backward_kink_free_all_264_gradients():
    """BASELINE configs[4]'s network (EDVR-L x4: nf 128, 7 frames, 40 blocks) on its 1x7x3x64x64 tile."""
    _kink_free_gradient_check(dict(EDVR_L), 8, 1, 64, 64, 31, 32)
