#!/usr/bin/env python3
"""EDVR-L x4 (BASELINE configs[4]) forward+backward loop on 1x7x3x64x64 for rocprofv3 --kernel-trace --stats.
usage: python tools/edvr_l_step_profile.py [bf16_mfma mode [steps]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops, synth  # noqa: E402
from dynavsr_amd.models.archs.EDVR_arch import EDVR  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = dict(nf=128, nframes=7, groups=8, front_RBs=5, back_RBs=40, scale=4)
net = EDVR(bf16_mfma=mode, **cfg)
net.load_state_dict(synth.edvr_state_dict(8, **cfg))
net = net.cuda()
x = synth.clip(9, 1, 7, 64, 64, smooth=False).cuda()
tgt = synth.clip(109, 1, 1, 256, 256, smooth=False)[:, 0].cuda()


def step():
    for p in net.parameters():
        p.grad = None
    hipops.charbonnier(net(x), tgt).backward()


if os.environ.get("EDVR_L_EXACT_STEPS"):   # PMC passes: exactly `steps` + 2 steps in the process (tools/pmc_total.py divides)
    step(); step()
else:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("EDVR-L 1x7x3x64x64 bf16_mfma=%d: fwd+bwd %.2f ms" % (mode, (time.perf_counter() - t0) / steps * 1e3))
