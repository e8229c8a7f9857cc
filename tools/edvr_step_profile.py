#!/usr/bin/env python3
"""EDVR forward+backward loop at a given clip size (for rocprofv3 --kernel-trace --stats).
usage: python tools/edvr_step_profile.py H W steps"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops, synth  # noqa: E402
from dynavsr_amd.models.archs.EDVR_arch import EDVR  # noqa: E402

h, w, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
net = EDVR()
net.load_state_dict(synth.edvr_state_dict(0))
net = net.cuda()
x = synth.clip(1, 1, 5, h, w, smooth=False).cuda().requires_grad_(True)
tgt = synth.clip(2, 1, 1, 4 * h, 4 * w, smooth=False)[:, 0].cuda()


def step():
    for p in net.parameters():
        p.grad = None
    loss = hipops.charbonnier(net(x), tgt)
    loss.backward()


step(); step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e3
t1 = time.perf_counter()
for _ in range(steps):
    with torch.no_grad():
        net(x)
torch.cuda.synchronize()
print("EDVR %dx%d: fwd+bwd %.2f ms, fwd only %.2f ms" % (h, w, dt, (time.perf_counter() - t1) / steps * 1e3))
