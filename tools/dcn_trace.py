#!/usr/bin/env python3
"""Cycle-stamp timeline of one mdcn_fwd_reg_kernel launch inside the EDVR forward (debug build:
python -m dynavsr_amd.build --trace).  usage (GPU box): python tools/dcn_trace.py [dcn_launch_index [H W]]
Stamps (thread 0 of each workgroup, groups 0 and 1 only): 0 start; per group g: 1+30g top, 2+30g window written,
3+30g barrier passed, then per tap t: 4+30g+2t tap top (offset loads of t+1 issued next), 5+30g+2t sampling done
(MFMAs follow until the next tap top); 62 loop done, 63 stores issued."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DVSR_HIP_LIB", os.path.join(HERE, "dynavsr_amd", "libdynavsr_hip_trace.so"))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dynavsr_amd import _lib, engine, synth  # noqa: E402
from dynavsr_amd.models.archs.EDVR_arch import EDVR  # noqa: E402

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
h = int(sys.argv[2]) if len(sys.argv) > 3 else 180
w = int(sys.argv[3]) if len(sys.argv) > 3 else 320
net = EDVR()
net.load_state_dict(synth.edvr_state_dict(0))
net = net.cuda()
x = synth.clip(1, 1, 5, h, w, smooth=False).cuda()
plan = engine.get_plan(net._cfg(), 1, h, w)
params = [p.detach().contiguous() for p in net.ordered_parameters()]
ws = torch.empty(plan.workspace_bytes(False), dtype=torch.uint8, device="cuda")
out = torch.empty(1, 3, 4 * h, 4 * w, device="cuda")
for _ in range(2):
    plan.forward(params, x, out, ws)
torch.cuda.synchronize()
NB = 1 << 14
buf = torch.zeros(NB * 64, dtype=torch.int64, device="cuda")
fn = _lib.lib().dvsr_debug_dcn_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn(buf.data_ptr(), idx)
plan.forward(params, x, out, ws)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NB, 64)
t = t[t[:, 0] != 0]
print("workgroups traced: %d" % len(t))
med = lambda v: float(np.median(v))
print("lifetime                      %8.0f cycles (median)" % med(t[:, 63] - t[:, 0]))
for g in range(2):
    b = 30 * g
    print("group %d: window fetch + LDS write %6.0f | barrier %6.0f" % (g, med(t[:, b + 2] - t[:, b + 1]), med(t[:, b + 3] - t[:, b + 2])))
    samp = [med(t[:, b + 5 + 2 * k] - t[:, b + 4 + 2 * k]) for k in range(9)]
    nxt = [t[:, b + 4 + 2 * (k + 1)] if k < 8 else (t[:, 31] if g == 0 else None) for k in range(9)]
    mf = [med(nxt[k] - t[:, b + 5 + 2 * k]) if nxt[k] is not None else float("nan") for k in range(9)]
    print("   sampling per tap:", " ".join("%5.0f" % v for v in samp))
    print("   A reads + MFMAs :", " ".join("%5.0f" % v for v in mf))
print("epilogue %6.0f" % med(t[:, 63] - t[:, 62]))
