#!/usr/bin/env python3
"""Throughput of K independent inner MAML steps in flight at once (K host threads, K HIP streams, K sets of copies):
frames of a video are adapted independently, and one step at LR 176x320 is launch-latency bound on the 44x80 levels.
usage (GPU box): python tools/inner_concurrent.py [K ...]"""
import os
import sys
import threading
import time
from copy import deepcopy

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
import bench  # noqa: E402
from dynavsr_amd import hipops, synth  # noqa: E402
from dynavsr_amd.adapt import make_inner_optimizer  # noqa: E402
from dynavsr_amd.models import create_model  # noqa: E402


def make_worker(seed, stream):
    opt = bench._opt()
    with torch.cuda.stream(stream):
        model, est = create_model(opt)
        _, est_fixed = create_model(opt)
        model.netG.load_state_dict(synth.edvr_state_dict(0)); est.netE.load_state_dict(synth.mfdn_state_dict(0))
        est_fixed.netE.load_state_dict(synth.mfdn_state_dict(1))
        model.netG, est.netE = deepcopy(model.netG), deepcopy(est.netE)
        inner = make_inner_optimizer(opt, model.netG, est.netE)
        lqs = synth.clip(3 + seed, 1, 5, 176, 320, smooth=False).cuda()
        data = {"LQs": lqs}
        est_fixed.feed_data(data); est_fixed.test()
        slr_fixed = est_fixed.fake_L

    def step():
        est.feed_data(data); est.forward_without_optim()
        inner.zero_grad()
        model.feed_data({"LQs": est.fake_L, "GT": lqs[:, 2]})
        loss = hipops.inner_loss(model.calculate_loss(), est.fake_L, slr_fixed, 10.0)
        loss.backward()
        inner.step()
    return step


def run(k, steps=40):
    streams = [torch.cuda.Stream() for _ in range(k)]
    workers = [make_worker(i, s) for i, s in enumerate(streams)]
    torch.cuda.synchronize()

    def loop(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                workers[i]()

    def go(n):
        ts = [threading.Thread(target=loop, args=(i, n)) for i in range(k)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
    go(20)
    t0 = time.perf_counter()
    go(steps)
    dt = time.perf_counter() - t0
    print("K = %d concurrent inner steps: %6.1f clips/s  (%.2f ms per step per stream, %.2f ms per clip)"
          % (k, k * steps / dt, dt / steps * 1e3, dt / steps / k * 1e3))


for k in ([int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]):
    run(k)
