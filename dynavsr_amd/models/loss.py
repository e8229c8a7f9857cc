"""Pixel criteria of the hot path.  CharbonnierLoss (codes/models/loss.py:19-30, eps=1e-6, mean)
is the ``pixel_criterion: cb`` of every EDVR YAML and runs on the HIP reduction kernels; it is
GPU-only (no CPU fallback)."""
import torch.nn as nn

from dynavsr_amd import hipops


class CharbonnierLoss(nn.Module):
    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        return hipops.charbonnier(x.float(), y.float(), self.eps)
