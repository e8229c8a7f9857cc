#!/usr/bin/env python3
"""Per-launch profile of the EDVR forward tape (hipEvents around every launch).
usage (GPU box): python tools/op_profile.py [H W [reps [bf16_mfma]]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import engine, synth  # noqa: E402
from dynavsr_amd.models.archs.EDVR_arch import EDVR  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 2 else 180
w = int(sys.argv[2]) if len(sys.argv) > 2 else 320
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
net = EDVR(bf16_mfma=mode)
net.load_state_dict(synth.edvr_state_dict(0))
net = net.cuda()
x = synth.clip(1, 1, 5, h, w, smooth=False).cuda()
plan = engine.get_plan(net._cfg(), 1, h, w)
params = [p.detach().contiguous() for p in net.ordered_parameters()]
ws = torch.empty(plan.workspace_bytes(False), dtype=torch.uint8, device="cuda")
out = torch.empty(1, 3, 4 * h, 4 * w, device="cuda")
info = plan.op_info()
tot = [0.0] * len(info)
for r in range(reps + 1):
    ms = plan.forward_timed(params, x, out, ws)
    if r:
        tot = [a + b for a, b in zip(tot, ms)]
print("%-4s %-10s %-26s %9s %9s %9s" % ("#", "kind", "name", "us", "TFLOP/s", "GB/s"))
for i, ((kind, name, fl, by), t) in enumerate(zip(info, tot)):
    us = t / reps * 1e3
    print("%-4d %-10s %-26s %9.1f %9.2f %9.1f" % (i, kind, name, us, fl / us / 1e6, by / us / 1e3))
print("total %.3f ms" % (sum(tot) / reps))
