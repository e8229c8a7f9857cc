"""Functional restatement of DynaVSR's test-time inner MAML step (TEST INFRASTRUCTURE, CPU).

Follows codes/test_dynavsr.py:208-283:
  * deep copies of netG / netE are adapted (:208), one optimizer over G u E params (:213-231),
  * per step: SLR = netE(LR) with grad (:238-241); loss = Charbonnier(netG(SLR), LR[:, center])
    (:262-264, models/Video_base_model.py:191-195, models/loss.py:26-30)
    + 10 * L1(SLR, netE_fixed(LR)) (:267-274); backward; optimizer step (:276-277),
  * then the adapted netG super-resolves the LR clip (:282-283).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .edvr import charbonnier, edvr_forward
from .mfdn import mfdn_forward


def crop(LR_seq, HR, num_patches_for_batch=4, patch_size=44):
    """test_dynavsr.py:118-145 with preprocessing.common_crop (:57-85) inlined: per patch py, px = randrange on the
    lowest-resolution grid, the same window (scaled) cut out of both tensors, patches stacked."""
    import random
    seq, hr = LR_seq[0], HR[0]
    min_h, min_w = min(seq.shape[-2], hr.shape[-2]), min(seq.shape[-1], hr.shape[-1])
    ps = patch_size // 2
    lrs, hrs = [], []
    for _ in range(num_patches_for_batch):
        py = random.randrange(0, min_h - ps + 1)
        px = random.randrange(0, min_w - ps + 1)
        s1, s2 = seq.shape[-2] // min_h, hr.shape[-2] // min_h
        lrs.append(seq[..., s1 * py:s1 * (py + ps), s1 * px:s1 * (px + ps)])
        hrs.append(hr[..., s2 * py:s2 * (py + ps), s2 * px:s2 * (px + ps)])
    return torch.stack(lrs, 0), torch.stack(hrs, 0)


def inner_adapt(PG, PE, PE_fixed, lqs, steps=1, optimizer="SGD", lr=1e-5, betas=(0.9, 0.99),
                edvr_cfg=None, scale=4, slr_weight=10.0, use_patch=False, num_patch=4, patch_size=44):
    """Returns (losses, PG_adapted, PE_adapted, sr) with sr = adapted netG applied to ``lqs``."""
    edvr_cfg = dict(edvr_cfg or {})
    PG = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in PG.items())
    PE = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in PE.items())
    params = list(PG.values()) + list(PE.values())
    opt = (torch.optim.Adam(params, lr=lr, betas=betas) if optimizer == "Adam"
           else torch.optim.SGD(params, lr=lr))
    center = lqs.shape[1] // 2
    target = lqs[:, center]
    with torch.no_grad():
        slr_fixed = mfdn_forward(PE_fixed, lqs, scale)
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        slr = mfdn_forward(PE, lqs, scale)
        if use_patch:                       # test_dynavsr.py:255-260
            p_lq, p_gt = crop(slr, target, num_patch, patch_size)
            loss = charbonnier(edvr_forward(PG, p_lq, scale=scale, **edvr_cfg), p_gt)
        else:
            loss = charbonnier(edvr_forward(PG, slr, scale=scale, **edvr_cfg), target)
        loss = loss + slr_weight * F.l1_loss(slr, slr_fixed)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    with torch.no_grad():
        sr = edvr_forward(PG, lqs, scale=scale, **edvr_cfg)
    return losses, PG, PE, sr


def tensor2img_rgb(t):
    """utils/util.py:112-142 for a 3-D CHW tensor, mode='rgb', uint8."""
    a = t.detach().float().cpu().clamp(0, 1).numpy().transpose(1, 2, 0)
    import numpy as np
    return (a * 255.0).round().astype(np.uint8)


def psnr_uint8(a, b):
    """utils/util.py:262-269."""
    import math
    import numpy as np
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
