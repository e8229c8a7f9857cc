#!/usr/bin/env python3
"""TOFlow loops on 1x7x3x256x448 for rocprofv3 --kernel-trace --stats.  usage: python tools/tof_profile.py [steps [train]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops, synth  # noqa: E402
from dynavsr_amd.models.archs import TOF_arch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
train = len(sys.argv) > 2 and sys.argv[2] == "train"
net = TOF_arch.TOFlow(adapt_official=True)
net.load_state_dict(synth.tof_state_dict(0))
net = net.cuda().train(train)
x = synth.clip(5, 1, 7, 256, 448, smooth=True).cuda()
tgt = torch.rand(1, 3, 256, 448, device="cuda")


def step():
    if not train:
        with torch.no_grad():
            net(x)
        return
    for p in net.parameters():
        p.grad = None
    hipops.charbonnier(net(x), tgt).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("TOFlow 1x7x3x256x448 %s %.2f ms" % ("train forward+backward" if train else "eval forward", (time.perf_counter() - t0) / steps * 1e3))
