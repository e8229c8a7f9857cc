#!/usr/bin/env python3
"""Host-side (launch) time vs synchronised time of EDVR forward / backward at a small clip size."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops, synth
from dynavsr_amd.models.archs.EDVR_arch import EDVR
h, w = int(sys.argv[1]), int(sys.argv[2])
net = EDVR(); net.load_state_dict(synth.edvr_state_dict(0)); net = net.cuda()
x = synth.clip(1, 1, 5, h, w, smooth=False).cuda().requires_grad_(True)
tgt = synth.clip(2, 1, 1, 4 * h, 4 * w, smooth=False)[:, 0].cuda()
def run(n, sync_each):
    tf = tb = 0.0
    for _ in range(n):
        for p in net.parameters(): p.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = hipops.charbonnier(net(x), tgt)
        if sync_each: torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss.backward()
        if sync_each: torch.cuda.synchronize()
        t2 = time.perf_counter()
        tf += t1 - t0; tb += t2 - t1
    torch.cuda.synchronize()
    return tf / n * 1e3, tb / n * 1e3
run(3, True)
print("%dx%d  host-only: fwd %.2f ms, bwd %.2f ms | synced: fwd %.2f ms, bwd %.2f ms" % ((h, w) + run(20, False) + run(20, True)))
