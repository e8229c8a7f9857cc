#!/usr/bin/env python3
"""fp32 MFMA ceiling on this box for the conv kernel's launch shapes (register-only MFMA loop)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import _lib as L
out = torch.zeros(1 << 22, device="cuda")
lib = L.lib()
def run(blocks, iters, nacc, lds):
    for _ in range(2):
        n = lib.dvsr_debug_mfma_peak(out.data_ptr(), blocks, iters, nacc, lds, L.stream())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        n = lib.dvsr_debug_mfma_peak(out.data_ptr(), blocks, iters, nacc, lds, L.stream())
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    flops = blocks * 4 * n * 2 * 32 * 32 * 2
    print("blocks %5d iters %4d nacc %d lds %6d: %8.1f us  %6.1f TFLOP/s" % (blocks, iters, nacc, lds, ms * 1e3, flops / ms / 1e9))
for blocks, iters, nacc, lds in [(256, 2000, 4, 0), (512, 1000, 4, 0), (768, 1000, 2, 49920), (768, 1000, 2, 0), (2250, 36, 2, 49920),
                                 (2250, 36, 2, 0), (1024, 500, 1, 0), (2048, 500, 2, 0)]:
    run(blocks, iters, nacc, lds)
