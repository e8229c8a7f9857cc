#!/usr/bin/env python3
"""Synchronised section times of one inner MAML step (each section followed by a device sync, so the sum
overstates the pipelined step; it shows where the time is).  usage (GPU box): python tools/inner_breakdown.py"""
import os
import sys
import time
from copy import deepcopy

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.adapt import make_inner_optimizer  # noqa: E402
from dynavsr_amd.models import create_model  # noqa: E402
from dynavsr_amd.options import options as option  # noqa: E402

h, w = 176, 320
opt = option.dict_to_nonedict(option.parse(os.path.join(ROOT, "dynavsr_amd/options/test/EDVR/EDVR_M_S4.yml"),
                                           is_train=False))
opt["dist"] = False
for k in ("pretrain_model_G", "pretrain_model_E"):
    opt["path"][k] = None
model, est = create_model(opt)
modelcp, estcp = create_model(opt)
_, est_fixed = create_model(opt)
model.netG.load_state_dict(synth.edvr_state_dict(0))
est.netE.load_state_dict(synth.mfdn_state_dict(0))
est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
lqs = synth.clip(1, 1, 5, h, w, smooth=False).cuda()
data = {"LQs": lqs}
modelcp.netG, estcp.netE = deepcopy(model.netG), deepcopy(est.netE)
inner = make_inner_optimizer(opt, modelcp.netG, estcp.netE)
est_fixed.feed_data(data); est_fixed.test()
slr_fixed = est_fixed.fake_L
acc = {}


def sec(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r


def step():
    def a():
        estcp.feed_data(data); estcp.forward_without_optim(); return estcp.fake_L
    slr = sec("MFDN forward (grad)", a)
    sec("zero_grad", inner.zero_grad)

    def b():
        modelcp.feed_data({"LQs": slr, "GT": lqs[:, 2]})
        return modelcp.calculate_loss() + 10 * F.l1_loss(slr, slr_fixed)
    loss = sec("EDVR forward + losses", b)
    sec("backward (EDVR + MFDN)", loss.backward)
    sec("optimizer step", inner.step)


for _ in range(3):
    step()
acc.clear()
n = 10
for _ in range(n):
    step()
tot = 0.0
for k, v in acc.items():
    print("%-26s %7.3f ms" % (k, v / n)); tot += v / n
print("%-26s %7.3f ms (synchronised sum)" % ("total", tot))
