#!/usr/bin/env python3
"""Times one outer (meta-training) iteration, adapt.meta_train_step = train_dynavsr.py:265-438, at the shapes of
the shipped training YAMLs: B tasks of LR 5x3x64x64 (HR 256x256, SLR 16x16), EDVR-M x4 + MFDN, adapt_iter inner steps.
usage (GPU box): python tools/meta_bench.py [B [adapt_iter]]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.adapt import meta_train_step  # noqa: E402
from dynavsr_amd.models import create_model  # noqa: E402
from dynavsr_amd.options import options as option  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
opt = option.dict_to_nonedict(option.parse(os.path.join(ROOT, "dynavsr_amd/options/test/EDVR/EDVR_M_S4.yml"), is_train=False))
opt["dist"] = False
for k in ("pretrain_model_G", "pretrain_model_E"):
    opt["path"][k] = None
opt["train"]["maml"]["adapt_iter"] = steps
model, est = create_model(opt)
modelcp, estcp = create_model(opt)
model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
params = [p for p in model.netG.parameters()] + [p for p in est.netE.parameters()]
optimizer = torch.optim.Adam(params, lr=1e-5, betas=(0.9, 0.99))
data = {"LQs": synth.clip(1, B, 5, 64, 64, smooth=False).cuda(), "SuperLQs": synth.clip(2, B, 5, 16, 16, smooth=False).cuda(),
        "GT": synth.clip(3, B, 5, 256, 256, smooth=False).cuda()}
for mode in ("reference", "copies"):
    for _ in range(2):
        meta_train_step(opt, model, est, modelcp, estcp, data, optimizer, inner=mode)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        r = meta_train_step(opt, model, est, modelcp, estcp, data, optimizer, inner=mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("meta_train_step inner=%-9s B=%d tasks, adapt_iter=%d: %7.1f ms per outer iteration -> %5.1f tasks/s (loss_q %.4f)"
          % (mode, B, steps, dt * 1e3, B / dt, r["loss_q"]))
