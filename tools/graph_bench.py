#!/usr/bin/env python3
"""EDVR forward+backward at the inner-step clip size, eager launches vs one captured hipGraph
(torch.cuda.CUDAGraph around the two native C calls).  usage (GPU box): python tools/graph_bench.py [H W [steps]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops, synth  # noqa: E402
from dynavsr_amd.models.archs.EDVR_arch import EDVR  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 2 else 44
w = int(sys.argv[2]) if len(sys.argv) > 2 else 80
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
net = EDVR()
net.load_state_dict(synth.edvr_state_dict(0))
net = net.cuda()
x = synth.clip(1, 1, 5, h, w, smooth=False).cuda()
tgt = synth.clip(2, 1, 1, 4 * h, 4 * w, smooth=False)[:, 0].cuda()
params = list(net.parameters())


def step():
    loss = hipops.charbonnier(net(x), tgt)
    grads = torch.autograd.grad(loss, params)
    return loss, grads


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("eager   fwd+bwd %dx%d: %.2f ms" % (h, w, timeit(step)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss_g, grads_g = step()
torch.cuda.synchronize()
print("graph   fwd+bwd %dx%d: %.2f ms" % (h, w, timeit(g.replay)))
loss_e, grads_e = step()
g.replay()
torch.cuda.synchronize()
err = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(grads_g, grads_e))
print("graph vs eager: loss %.6e vs %.6e, max rel grad diff %.2e" % (float(loss_g), float(loss_e), err))
