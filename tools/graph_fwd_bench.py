#!/usr/bin/env python3
"""EDVR-M x4 forward at the headline size, eager launches vs one captured hipGraph (torch.cuda.CUDAGraph around the native
call): how much of the step is launch gaps.  usage (GPU box): python tools/graph_fwd_bench.py [H W [steps]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.models.archs.EDVR_arch import EDVR  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 2 else 180
w = int(sys.argv[2]) if len(sys.argv) > 2 else 320
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
net = EDVR()
net.load_state_dict(synth.edvr_state_dict(0))
net = net.cuda()
x = synth.clip(1, 1, 5, h, w, smooth=False).cuda()


def step():
    with torch.no_grad():
        return net(x)


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("eager   forward %dx%d: %.3f ms" % (h, w, timeit(step)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y_g = step()
torch.cuda.synchronize()
print("graph   forward %dx%d: %.3f ms" % (h, w, timeit(g.replay)))
y_e = step()
g.replay()
torch.cuda.synchronize()
print("graph vs eager: max abs diff %.2e" % float((y_g - y_e).abs().max()))
