#!/usr/bin/env python3
"""TOFlow eval forward loop on 1x7x3x256x448 for rocprofv3 --kernel-trace --stats.  usage: python tools/tof_profile.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.models.archs import TOF_arch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
net = TOF_arch.TOFlow(adapt_official=True)
net.load_state_dict(synth.tof_state_dict(0))
net = net.cuda().eval()
x = synth.clip(5, 1, 7, 256, 448, smooth=True).cuda()
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net(x)
    torch.cuda.synchronize()
print("TOFlow 1x7x3x256x448 eval forward %.2f ms" % ((time.perf_counter() - t0) / steps * 1e3))
