#!/usr/bin/env python3
"""How many instructions hide behind a 64-cycle v_mfma_f32_32x32x2_f32 of the same wave (one wave per SIMD).
Prints shader cycles per MFMA for 0..16 v_fma_f32 / ds_read_b32 behind each MFMA, accumulators in VGPRs or AGPRs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import _lib as L  # noqa: E402

out = torch.zeros(1 << 20, device="cuda")
cyc = torch.zeros(4, dtype=torch.int64, device="cuda")
lib = L.lib()
iters = 2000
for kind, kname in ((0, "v_fma_f32"), (1, "ds_read_b32")):
    for acc, aname in ((0, "VGPR acc"), (1, "AGPR acc")):
        row = []
        for nv in (0, 4, 8, 12, 16):
            for _ in range(2):
                L.check(lib.dvsr_debug_mfma_shadow(cyc.data_ptr(), out.data_ptr(), 256, iters, nv, kind, acc, L.stream()), "shadow")
            torch.cuda.synchronize()
            row.append("%d: %6.1f" % (nv, cyc[0].item() / (iters * 16)))
        print("%-12s %-9s cycles per MFMA with n instructions behind it | %s" % (kname, aname, " | ".join(row)))
