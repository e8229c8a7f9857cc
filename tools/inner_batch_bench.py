#!/usr/bin/env python3
"""Inner MAML step per frame: the per-frame loop (adapt_frame) against K frames as one batch with per-frame parameter
gradients (adapt.FrameBatch), and the per-frame pipeline (adapt_video) for both.
usage (GPU box): python tools/inner_batch_bench.py [H W]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.adapt import FrameBatch, adapt_frame, adapt_video  # noqa: E402
from dynavsr_amd.models import create_model  # noqa: E402
from dynavsr_amd.options import options as option  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 2 else 176
w = int(sys.argv[2]) if len(sys.argv) > 2 else 320
opt = option.dict_to_nonedict(option.parse(os.path.join(ROOT, "dynavsr_amd", "options", "test", "EDVR", "EDVR_M_S4.yml"),
                                           is_train=False))
opt["dist"] = False
for k in ("pretrain_model_G", "pretrain_model_E"):
    opt["path"][k] = None
model, est = create_model(opt)
modelcp, estcp = create_model(opt)
_, est_fixed = create_model(opt)
model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
GF = (3 * 973.6 * (h // 4) * (w // 4) + 4 * 53.1 * h * w) / (180.0 * 320.0)   # GFLOP per inner step


def timed(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


lq1 = {"LQs": synth.clip(1, 1, 5, h, w, smooth=False).cuda()}
ms = timed(lambda: adapt_frame(opt, model, est, modelcp, estcp, est_fixed, lq1, final_test=False), 40)
print("per-frame loop      : %.2f ms per frame-step  %.1f TFLOP/s (%.2f of the fp32 MFMA peak)" % (ms, GF / ms, GF / ms / 157.3))
for K in (2, 4, 8):
    fb = FrameBatch(opt, model.netG, est.netE, K)
    lqs = synth.clip(2, K, 5, h, w, smooth=False).cuda()
    ms = timed(lambda: fb.adapt(model, est, est_fixed, lqs), 20) / K
    print("batch of %d frames   : %.2f ms per frame-step  %.1f TFLOP/s (%.2f of the fp32 MFMA peak)" % (K, ms, GF / ms, GF / ms / 157.3))
    del fb
clips = [{"LQs": synth.clip(10 + i, 1, 5, h, w, smooth=False).cuda()} for i in range(16)]
for K in (1, 4, 8):
    for ov in (False, True):
        for _ in adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips[:8], overlap=ov, frames_per_batch=K):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips, overlap=ov, frames_per_batch=K):
            pass
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / len(clips) * 1e3
        print("adapt_video K=%d overlap=%d: %.2f ms per frame (%.1f frames/s)" % (K, ov, ms, 1e3 / ms))
