#!/usr/bin/env python3
"""DUF eval forward loop on 1x7x3x64x112 for rocprofv3 --kernel-trace --stats.  usage: python tools/duf_profile.py [layers [steps]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.models.archs import DUF_arch  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 52
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
net = {16: DUF_arch.DUF_16L, 28: DUF_arch.DUF_28L, 52: DUF_arch.DUF_52L}[layers](scale=4, adapt_official=True)
net.load_state_dict(synth.duf_state_dict(0, layers, scale=4))
net = net.cuda().eval()
x = synth.clip(5, 1, 7, 64, 112, smooth=True).cuda()
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net(x)
    torch.cuda.synchronize()
print("DUF-%dL 1x7x3x64x112 eval forward %.2f ms" % (layers, (time.perf_counter() - t0) / steps * 1e3))
