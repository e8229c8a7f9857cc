#!/usr/bin/env python3
"""Times the inner MAML step (BASELINE north_star: >= 50 clips/s at LR 176x320 -> SLR 44x80) and the
full per-frame pipeline of test_dynavsr.py through the wrapper API on synthetic data.
usage (GPU box): python tools/inner_bench.py [H W [steps]]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.adapt import adapt_frame  # noqa: E402
from dynavsr_amd.models import create_model  # noqa: E402
from dynavsr_amd.options import options as option  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 2 else 176
w = int(sys.argv[2]) if len(sys.argv) > 2 else 320
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
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


def sync():
    torch.cuda.synchronize()


def timeit(fn, n):
    fn(); fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e3


# (ii) inner step only: MFDN fwd (grad) + EDVR fwd/bwd on SLR + losses + optimizer step
from copy import deepcopy  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from dynavsr_amd.adapt import make_inner_optimizer  # noqa: E402

modelcp.netG, estcp.netE = deepcopy(model.netG), deepcopy(est.netE)
inner = make_inner_optimizer(opt, modelcp.netG, estcp.netE)
est_fixed.feed_data(data); est_fixed.test()
slr_fixed = est_fixed.fake_L


def inner_step():
    estcp.feed_data(data); estcp.forward_without_optim()
    slr = estcp.fake_L
    inner.zero_grad()
    modelcp.feed_data({"LQs": slr, "GT": lqs[:, 2]})
    loss = modelcp.calculate_loss() + 10 * F.l1_loss(slr, slr_fixed)
    loss.backward()
    inner.step()


def edvr_fwd_bwd():
    slr = estcp.fake_L.detach()
    modelcp.feed_data({"LQs": slr, "GT": lqs[:, 2]})
    modelcp.calculate_loss().backward()


def mfdn_fwd():
    estcp.feed_data(data); estcp.test()


def full_frame():
    modelcp.feed_data(data, need_GT=False); modelcp.test()          # baseline forward
    adapt_frame(opt, model, est, modelcp, estcp, est_fixed, data)   # deepcopy + inner step + adapted forward


gt = synth.clip(7, 1, 1, 4 * h, 4 * w, smooth=False)[0, 0].cuda()
from dynavsr_amd.utils import util  # noqa: E402


def full_frame_with_metrics():   # test_dynavsr.py:197-305 per frame: both super-resolved frames are scored
    modelcp.feed_data(data, need_GT=False); modelcp.test()
    m0 = util.frame_metrics(modelcp.fake_H, gt, need_img=True)
    r = adapt_frame(opt, model, est, modelcp, estcp, est_fixed, data)
    m1 = util.frame_metrics(r["sr"], gt, need_img=True)
    return m0, m1


def host_metrics():              # the reference's way: fp32 frame to the host, tensor2img, calculate_psnr
    img = util.tensor2img(modelcp.fake_H, mode="rgb")
    return util.calculate_psnr(img, util.tensor2img(gt, mode="rgb"))


print("LR %dx%d (SLR %dx%d), EDVR-M x4 + MFDN, fp32" % (h, w, h // 4, w // 4))
t = timeit(inner_step, steps); print("inner step            %8.2f ms  -> %6.1f clips/s" % (t, 1e3 / t))
t = timeit(edvr_fwd_bwd, steps); print("  EDVR fwd+bwd on SLR   %8.2f ms" % t)
t = timeit(mfdn_fwd, steps); print("  MFDN fwd (no grad)    %8.2f ms" % t)
t = timeit(full_frame, max(2, steps // 3)); print("full per-frame pipeline %8.2f ms -> %6.1f frames/s" % (t, 1e3 / t))
t = timeit(full_frame_with_metrics, max(2, steps // 3)); print("  + PSNR/SSIM + uint8 image of both frames on the GPU %8.2f ms -> %6.1f frames/s" % (t, 1e3 / t))
t = timeit(host_metrics, 3); print("  (host path of ONE frame's tensor2img x2 + PSNR, no SSIM: %8.2f ms)" % t)


def t_deepcopy():
    modelcp.netG, estcp.netE = deepcopy(model.netG), deepcopy(est.netE)


def t_optim():
    make_inner_optimizer(opt, modelcp.netG, estcp.netE)


def t_fixed():
    est_fixed.feed_data(data); est_fixed.test()


def t_test():
    modelcp.feed_data(data, need_GT=False); modelcp.test()


print("per-frame pipeline pieces: deepcopy(netG, netE) %.2f ms | inner optimizer construction %.2f ms | frozen estimator %.2f ms | "
      "EDVR test() at LR size %.2f ms" % (timeit(t_deepcopy, 5), timeit(t_optim, 5), timeit(t_fixed, 5), timeit(t_test, 5)))

from dynavsr_amd.adapt import adapt_video  # noqa: E402
clips = [{"LQs": synth.clip(10 + i, 1, 5, h, w, smooth=False).cuda()} for i in range(12)]
for ov in (False, True):
    for _ in adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips[:3], overlap=ov):
        pass
    sync()
    t0 = time.perf_counter()
    outs = [(a, r["sr"]) for a, r in adapt_video(opt, model, est, modelcp, estcp, est_fixed, clips, overlap=ov)]
    sync()
    t = (time.perf_counter() - t0) / len(clips) * 1e3
    print("adapt_video over %d clips, baseline of clip i+1 %s: %6.2f ms per frame -> %5.1f frames/s"
          % (len(clips), "and adapted forward of clip i on side streams under the next adaptation" if ov else "sequential", t, 1e3 / t))
    if ov:
        print("  overlapped == sequential results: max |diff| baseline %.1e adapted %.1e"
              % (max(float((a - b).abs().max()) for (a, _), (b, _) in zip(outs, ref)),
                 max(float((a - b).abs().max()) for (_, a), (_, b) in zip(outs, ref))))
    ref = outs
