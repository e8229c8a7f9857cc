#!/usr/bin/env python3
"""MFDN x4 forward / forward+backward timing at the inner-step clip size (LR 176x320, 5 frames).
usage (GPU box): python tools/estimator_bench.py [H W [steps]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.models.archs.LRimg_estimator import DirectKernelEstimatorVideo  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 2 else 176
w = int(sys.argv[2]) if len(sys.argv) > 2 else 320
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
net = DirectKernelEstimatorVideo(64, 3, 4)
net.load_state_dict(synth.mfdn_state_dict(0))
net = net.cuda()
x = synth.clip(1, 1, 5, h, w, smooth=False).transpose(1, 2).contiguous().cuda()
go = torch.randn(1, 3, 5, h // 4, w // 4, device="cuda")


def fwd():
    with torch.no_grad():
        net(x)


def fwd_bwd():
    for p in net.parameters():
        p.grad = None
    net(x).backward(go)


for name, fn in (("forward", fwd), ("forward+backward", fwd_bwd)):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    print("MFDN x4 %dx%d %-18s %7.3f ms" % (h, w, name, (time.perf_counter() - t0) / steps * 1e3))
