#!/usr/bin/env python3
"""EDVR-M x4 forward at the headline size: S independent clips, one per HIP stream, against the same clips one after the
other on one stream -- how much of the forward is tail rounds and launch gaps that a second queue can fill.
usage (GPU box): python tools/fwd_concurrent.py [H W [steps [streams [clips per forward]]]]"""
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
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ns = int(sys.argv[4]) if len(sys.argv) > 4 else 2
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
net = EDVR()
net.load_state_dict(synth.edvr_state_dict(0))
net = net.cuda()
xs = [synth.clip(1 + i, B, 5, h, w, smooth=False).cuda() for i in range(ns)]
streams = [torch.cuda.Stream() for _ in range(ns)]


def serial():
    with torch.no_grad():
        return [net(x) for x in xs]


def concurrent():
    out = []
    with torch.no_grad():
        for s, x in zip(streams, xs):
            with torch.cuda.stream(s):
                out.append(net(x))
    return out


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


torch.cuda.synchronize()
a = timeit(serial)
b = timeit(concurrent)
print("%d x %d clips %dx%d one stream : %.3f ms (%.1f frames/s)" % (ns, B, h, w, a, ns * B * 1e3 / a))
print("%d x %d clips %dx%d %d streams : %.3f ms (%.1f frames/s)" % (ns, B, h, w, ns, b, ns * B * 1e3 / b))
ya, yb = serial(), concurrent()
torch.cuda.synchronize()
print("max abs diff: %.2e" % max(float((p - q).abs().max()) for p, q in zip(ya, yb)))
