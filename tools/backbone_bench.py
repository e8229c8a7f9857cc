#!/usr/bin/env python3
"""Forward and forward+backward time of the two other video backbones behind define_G on synthetic clips:
TOFlow on 1x7x3x256x448 (it runs at the output resolution: the drivers up-sample the LR clip first,
test_dynavsr.py:244-250) and DUF-16L / 28L / 52L x4 on 1x7x3x64x112, with the algorithmic FLOPs of their convolutions.
usage (GPU box): python tools/backbone_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
from dynavsr_amd import hipops, synth  # noqa: E402
from dynavsr_amd.models.archs import DUF_arch, TOF_arch  # noqa: E402


def conv_flops(net, x):
    """2*MAC of every nn.Conv2d / nn.Conv3d parameter set at the spatial size it runs on (SpyNet: 4 pyramid levels x 6
    neighbours; DUF: T frames, T-2 after each reducing conv)."""
    b, t, c, h, w = x.shape
    fl = 0.0
    if isinstance(net, TOF_arch.TOFlow):
        for lvl, blk in enumerate(net.SpyNet.blocks):
            px = (h >> (3 - lvl)) * (w >> (3 - lvl)) * 6 * b
            for m in blk.block:
                if isinstance(m, torch.nn.Conv2d):
                    fl += 2.0 * px * m.weight.numel()
        for m in (net.conv_3x7_64_9x9, net.conv_64_64_9x9, net.conv_64_64_1x1, net.conv_64_3_1x1):
            fl += 2.0 * b * h * w * m.weight.numel()
        return fl
    frames = {"dense_block_2.conv3d_2": 5, "dense_block_2.conv3d_3": 5, "dense_block_2.conv3d_4": 3, "dense_block_2.conv3d_5": 3,
              "dense_block_2.conv3d_6": 1, "dense_block_2.conv3d_1": 7}
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.Conv3d):
            tt = frames.get(name, 1 if name.startswith(("conv3d_2", "conv3d_r", "conv3d_f")) else 7)
            fl += 2.0 * b * tt * h * w * m.weight.numel()
    return fl


def run(name, net, x, out_scale):
    net = net.cuda()
    x = x.cuda()
    tgt = torch.rand(x.shape[0], 3, out_scale * x.shape[-2], out_scale * x.shape[-1], device="cuda")

    def fwd():
        with torch.no_grad():
            net(x)

    def fwd_bwd():
        for p in net.parameters():
            p.grad = None
        hipops.charbonnier(net(x), tgt).backward()
    fl = conv_flops(net, x)
    res = []
    for mode, fn, mult in (("eval forward", fwd, 1.0), ("train forward+backward", fwd_bwd, 3.0)):
        net.train(mode.startswith("train"))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            fn()
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        res.append("%s %8.2f ms (%5.1f TFLOP/s)" % (mode, ms, mult * fl / ms / 1e9))
    print("%-10s %-18s %6.1f GFLOP fwd | %s" % (name, "x".join(map(str, x.shape)), fl / 1e9, " | ".join(res)))


def main():
    tof = TOF_arch.TOFlow(adapt_official=True)
    tof.load_state_dict(synth.tof_state_dict(0))
    run("TOFlow", tof, synth.clip(1, 1, 7, 256, 448, smooth=False), 1)
    for layers, cls in ((16, DUF_arch.DUF_16L), (28, DUF_arch.DUF_28L), (52, DUF_arch.DUF_52L)):
        net = cls(scale=4, adapt_official=True)
        net.load_state_dict(synth.duf_state_dict(0, layers, 4))
        run("DUF-%dL x4" % layers, net, synth.clip(2, 1, 7, 64, 112, smooth=False), 4)


if __name__ == "__main__":
    main()
