#!/usr/bin/env python3
"""Phases of the batched inner step (adapt.FrameBatch.adapt) with a device synchronisation between them, and the
un-synchronised whole; run it under `rocprofv3 --kernel-trace --stats` for the kernel table.
usage (GPU box): python tools/inner_batch_profile.py [K [iters [H W]]]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()
from dynavsr_amd import hipops, synth  # noqa: E402
from dynavsr_amd.adapt import FrameBatch  # noqa: E402
from dynavsr_amd.models import create_model  # noqa: E402
from dynavsr_amd.options import options as option  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h = int(sys.argv[3]) if len(sys.argv) > 4 else 176
w = int(sys.argv[4]) if len(sys.argv) > 4 else 320
opt = option.dict_to_nonedict(option.parse(os.path.join(ROOT, "dynavsr_amd", "options", "test", "EDVR", "EDVR_M_S4.yml"),
                                           is_train=False))
opt["dist"] = False
for k in ("pretrain_model_G", "pretrain_model_E"):
    opt["path"][k] = None
model, est = create_model(opt)
_, est_fixed = create_model(opt)
model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
fb = FrameBatch(opt, model.netG, est.netE, K)
lqs = synth.clip(2, K, 5, h, w, smooth=False).cuda()
for _ in range(5):
    fb.adapt(model, est, est_fixed, lqs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    fb.adapt(model, est, est_fixed, lqs)
torch.cuda.synchronize()
whole = (time.perf_counter() - t0) / iters * 1e3
print("batched inner step K=%d LR %dx%d: %.2f ms per batch, %.2f ms per frame" % (K, h, w, whole, whole / K))

acc = {}


def phase(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
    return r


for _ in range(iters):
    phase("refresh copies + optimiser reset", lambda: fb.refresh(model.netG, est.netE))
    center = lqs.size(1) // 2

    def fixed():
        est_fixed.feed_data({'LQs': lqs}); est_fixed.test()
        return est_fixed.fake_L
    slr_fixed = phase("frozen MFDN forward", fixed)
    est.feed_data({'LQs': lqs})
    y = phase("MFDN forward", lambda: est.netE.forward_stacked(est.var_H, fb.e_stack))
    slr = y.transpose(1, 2)
    sr = phase("EDVR forward (SLR)", lambda: model.netG.forward_stacked(slr, fb.g_stack))
    loss = phase("losses forward", lambda: hipops.inner_loss_per_sample(
        model.l_pix_w * hipops.charbonnier_per_sample(sr, lqs[:, center], model.cri_pix.eps), slr, slr_fixed, 10.0))
    phase("backward (losses, EDVR, MFDN)", lambda: loss.sum().backward())
    phase("optimiser step", lambda: fb.inner.step())
tot = sum(acc.values())
for k_, v in acc.items():
    print("  %-36s %8.3f ms per batch  %6.3f per frame  %5.1f %%" % (k_, v / iters, v / iters / K, 100 * v / tot))
print("  %-36s %8.3f ms per batch  %6.3f per frame (synchronised phases)" % ("sum", tot / iters, tot / iters / K))
