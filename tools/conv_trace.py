#!/usr/bin/env python3
"""Cycle-stamp timeline of one conv2d_pipe_kernel launch inside the EDVR forward.

Needs the debug build (python -m dynavsr_amd.build --trace); run on the GPU box:
    python tools/conv_trace.py [launch_index [H W]]
Stamps (s_memtime, thread 0 of each workgroup): 0 start, 1 chunk 0 staged, per chunk k: 2+4k block top,
3+4k next chunk's DMA + halo loads issued, 4+4k 3/4 of the MFMAs issued, 5+4k next halo written to LDS
(then the rest of the MFMAs and the barrier), 40 loop done, 41 stores issued, 42 stores acknowledged.
"""
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

sel = sys.argv[1] if len(sys.argv) > 1 else "1"  # launch index among the packed convs, or an op name
h = int(sys.argv[2]) if len(sys.argv) > 3 else 180
w = int(sys.argv[3]) if len(sys.argv) > 3 else 320
net = EDVR(bf16_mfma=int(os.environ.get("DVSR_TRACE_MFMA_MODE", "0")))   # 2: the 3-way bf16 split kernel
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
NB = 1 << 16
buf = torch.zeros(NB * 64, dtype=torch.int64, device="cuda")
fn = _lib.lib().dvsr_debug_conv_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
if sel.isdigit():
    idx = int(sel)
else:
    kinds = [(k, nm) for (k, nm, _, _) in plan.op_info()]
    pos = next(i for i, (k, nm) in enumerate(kinds) if nm.split("[")[0] == sel)
    idx = sum(1 for k, nm in kinds[:pos] if k.startswith("conv"))
    print("op %s = tape position %d, packed-conv launch %d" % (sel, pos, idx))
fn(buf.data_ptr(), idx)
plan.forward(params, x, out, ws)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NB, 64)
used = t[:, 0] != 0
t = t[used]
print("workgroups traced: %d" % len(t))
t0 = t[:, 0].min()
hw = t[:, 63]
# HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ... ; XCC in a separate reg
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
start = t[:, 0] - t0
end = t[:, 42] - t0
print("kernel span (cycles): %d   [stamps are s_memtime ticks]" % end.max())


def stat(name, d):
    d = np.asarray(d, dtype=np.float64)
    print("%-34s median %8.0f  p10 %8.0f  p90 %8.0f  mean %8.0f" % (name, np.median(d), np.percentile(d, 10),
                                                                    np.percentile(d, 90), d.mean()))


stat("workgroup lifetime", t[:, 42] - t[:, 0])
rt = (t[:, 61] - t[:, 60]).astype(np.float64)  # s_memrealtime: constant 100 MHz
print("core clock from s_memtime / s_memrealtime: %.0f MHz (median over workgroups); kernel wall span %.1f us" %
      (np.median((t[:, 42] - t[:, 0]) / np.maximum(rt, 1)) * 100.0, (t[:, 61].max() - t[:, 60].min()) / 100.0))
stat("prologue: chunk 0 staged", t[:, 1] - t[:, 0])
nch = 8
for k in range(nch):
    b = 2 + 4 * k
    if not t[:, b].any():
        nch = k
        break
pref = np.stack([t[:, 3 + 4 * k] - t[:, 2 + 4 * k] for k in range(nch)], 1)
has_w = [t[:, 4 + 4 * k].any() for k in range(nch)]
m1 = np.stack([(t[:, 4 + 4 * k] if has_w[k] else t[:, 3 + 4 * k]) - t[:, 3 + 4 * k] for k in range(nch)], 1)
wr = np.stack([(t[:, 5 + 4 * k] - t[:, 4 + 4 * k]) if has_w[k] else 0 * t[:, 0] for k in range(nch)], 1)
nxt = [t[:, 2 + 4 * (k + 1)] if k + 1 < nch else t[:, 40] for k in range(nch)]
m2 = np.stack([nxt[k] - (t[:, 5 + 4 * k] if has_w[k] else t[:, 3 + 4 * k]) for k in range(nch)], 1)
for k in range(nch):
    print("chunk %d: prefetch issue %6.0f | MFMA part 1 %6.0f | wait+LDS write %6.0f | MFMA part 2 + barrier %6.0f   (medians)" %
          (k, np.median(pref[:, k]), np.median(m1[:, k]), np.median(wr[:, k]), np.median(m2[:, k])))
stat("sum prefetch issue / wg", pref.sum(1))
stat("sum MFMA part 1 / wg", m1.sum(1))
stat("sum wait+LDS write / wg", wr.sum(1))
stat("sum MFMA part 2 + barrier / wg", m2.sum(1))
stat("epilogue issue", t[:, 41] - t[:, 40])
stat("epilogue store ack", t[:, 42] - t[:, 41])
# occupancy per CU: workgroups that ran on the same CU (HW_ID se/sh/cu; the 8 XCDs have unrelated s_memtime
# bases, so a CU group is additionally split where start times jump by more than 1e8 ticks)
key = ((hw >> 8) & 0xFF).astype(np.int64)  # cu_id, sh_id, se_id
groups = {}
for i in range(len(t)):
    groups.setdefault(int(key[i]), []).append(i)
conc, gaps = [], []
for k, idxs in groups.items():
    idxs.sort(key=lambda i: t[i, 0])
    run = [idxs[0]]
    runs = []
    for a, b in zip(idxs[:-1], idxs[1:]):
        if t[b, 0] - t[a, 0] > 1e8:
            runs.append(run); run = []
        run.append(b)
    runs.append(run)
    for r in runs:
        if len(r) < 4:
            continue
        st, en = t[r, 0], t[r, 42]
        span = float(en.max() - st.min())
        busy = float((en - st).sum())
        conc.append(busy / span)
        # per-slot turnover: time from the k-th end to the (k+3)-th start when 3 workgroups share a CU
        ends = np.sort(en); starts = np.sort(st)
        if len(starts) > 3:
            gaps.extend(list(starts[3:] - ends[:len(starts) - 3]))
if conc:
    print("per-CU occupancy (sum of workgroup lifetimes / CU active span): median %.2f  p10 %.2f  p90 %.2f  (3 = always full)" %
          (np.median(conc), np.percentile(conc, 10), np.percentile(conc, 90)))
    print("slot turnover (next workgroup's start - a finished one's end, 3 slots per CU): median %.0f cycles  p90 %.0f" %
          (np.median(gaps), np.percentile(gaps, 90)))
# start-time histogram: how many rounds of workgroups
order = np.argsort(start)
print("start times (cycles) deciles:", np.percentile(start, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]).astype(int))
print("end   times (cycles) deciles:", np.percentile(end, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]).astype(int))
