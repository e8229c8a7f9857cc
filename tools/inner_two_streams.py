#!/usr/bin/env python3
"""Batched inner step (adapt.FrameBatch) of K frames as ONE batch on one stream against S batches of K / S frames, one per
HIP stream, in flight together.  usage (GPU box): python tools/inner_two_streams.py [K [S [reps]]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (configures the runtime)
from dynavsr_amd import synth  # noqa: E402
from dynavsr_amd.adapt import FrameBatch  # noqa: E402
from dynavsr_amd.models import create_model  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
h, w = 176, 320
opt = bench._opt()
model, est = create_model(opt)
_, est_fixed = create_model(opt)
model.netG.load_state_dict(synth.edvr_state_dict(0))
est.netE.load_state_dict(synth.mfdn_state_dict(0))
est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
lqs = synth.clip(3, K, 5, h, w, smooth=False).cuda()


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


fb = FrameBatch(opt, model.netG, est.netE, K)
one = timed(lambda: fb.adapt(model, est, est_fixed, lqs))
del fb
streams = [torch.cuda.Stream() for _ in range(S)]
fbs = []
for s in streams:
    with torch.cuda.stream(s):
        fbs.append(FrameBatch(opt, model.netG, est.netE, K // S))
parts = [lqs[i * (K // S):(i + 1) * (K // S)].contiguous() for i in range(S)]
torch.cuda.synchronize()


def many():
    for s, f, x in zip(streams, fbs, parts):
        with torch.cuda.stream(s):
            f.adapt(model, est, est_fixed, x)


two = timed(many)
print("inner step K=%d one batch, one stream : %.2f ms per batch, %.3f ms per frame" % (K, one, one / K))
print("inner step K=%d as %d x %d on %d streams: %.2f ms per batch, %.3f ms per frame" % (K, S, K // S, S, two, two / K))
