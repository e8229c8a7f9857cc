#!/usr/bin/env python3
"""Per-frame metrics: dvsr_frame_metrics on the GPU vs the reference's host path
(device->host copy of the fp32 frame, tensor2img, calculate_psnr; the reference's SSIM needs cv2, which is not
installed -- its parity is tests/test_gpu_metrics.py's business, not this timing script's).
usage (GPU box): python tools/metrics_bench.py [H W]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd.utils import util  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 2 else 720
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
r = np.random.RandomState(0)
a = r.rand(3, h, w).astype(np.float32)
b = np.clip(a + 0.03 * r.standard_normal(a.shape).astype(np.float32), 0, 1).astype(np.float32)
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
for _ in range(3):
    util.frame_metrics(ta, tb, need_img=True)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    psnr, ssim, img = util.frame_metrics(ta, tb, need_img=True)      # includes the 2-double + uint8-image readback
gpu_ms = (time.perf_counter() - t0) / n * 1e3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
from dynavsr_amd import _lib as L  # noqa: E402
lib = L.lib()
ws = torch.empty(lib.dvsr_frame_metrics_workspace_bytes(3, h, w), dtype=torch.uint8, device="cuda")
out = torch.empty(2, dtype=torch.float64, device="cuda")
im = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
e0.record()
for _ in range(n):
    L.check(lib.dvsr_frame_metrics(L.ptr(ta), L.ptr(tb), 3, h, w, 0.0, 1.0, im.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), L.stream()))
e1.record()
torch.cuda.synchronize()
dev_us = e0.elapsed_time(e1) / n * 1e3
algo = 3 * h * w * (4 + 4 + 1 + 1 + 1 + 1 + 1)     # read 2 fp32, write 2 planar u8 + HWC u8, re-read the 2 planes
t0 = time.perf_counter()
ia = util.tensor2img(ta, mode="rgb")                 # the reference's path: D2H of fp32 + host quantisation
ib = util.tensor2img(tb, mode="rgb")
p_ref = util.calculate_psnr(ia, ib)
t1 = time.perf_counter()
print("frame 3x%dx%d: GPU kernels %.1f us (%.0f GB/s of %.1f MB algorithmic traffic), call incl. readback %.2f ms"
      % (h, w, dev_us, algo / dev_us / 1e3, algo / 1e6, gpu_ms))
print("host path (fp32 D2H + tensor2img x2 + PSNR, no SSIM) %.1f ms" % ((t1 - t0) * 1e3))
print("psnr gpu %.6f host %.6f | ssim gpu %.9f" % (psnr, p_ref, ssim))
