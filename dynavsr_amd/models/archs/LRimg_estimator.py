"""Down-scaling estimators MFDN / SFDN (state-dict compatible with
codes/models/archs/LRimg_estimator.py:38-117).

First slice: these run on stock PyTorch-ROCm ops (MIOpen
Conv2d/Conv3d) on the GPU; native kernels for them are the declared next step.  The graph is
expressed with functional calls over the reference's parameter names (conv0..conv6).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def conv3d_k3_replicate(x, weight, bias):
    """nn.Conv3d(k=3, s=1) over ReplicationPad3d(1) input, computed as ONE batched 2-D convolution.

    out[:, :, t] = sum_dt conv2d(pad2d(x[:, :, clamp(t+dt-1)]), W[:, :, dt]): the three temporal taps
    are folded into the channel axis (3*Cin channels, weight [Cout, 3*Cin, 3, 3]).  Same arithmetic
    as LRimg_estimator.py:100,113 (summation order aside); it avoids MIOpen's naive Conv3d
    weight-gradient kernel, which took 43 ms per inner step on MI355X (profiles/r01_*inner*)."""
    b, c, t, h, w = x.shape
    xp = F.pad(x.transpose(1, 2).reshape(b * t, c, h, w), (1, 1, 1, 1), mode='replicate')
    xp = xp.view(b, t, c, h + 2, w + 2)
    idx = torch.arange(t, device=x.device)
    frames = torch.cat([xp[:, (idx - 1).clamp(min=0)], xp, xp[:, (idx + 1).clamp(max=t - 1)]], dim=2)
    w2 = weight.permute(0, 2, 1, 3, 4).reshape(weight.shape[0], 3 * c, 3, 3)
    y = F.conv2d(frames.reshape(b * t, 3 * c, h + 2, w + 2), w2, bias)
    return y.view(b, t, -1, h, w).transpose(1, 2)


class DirectKernelEstimatorVideo(nn.Module):
    """MFDN: multi-frame estimator, input B,C,T,H,W -> B,C,T,H/s,W/s."""

    def __init__(self, nf, in_nc=3, scale=2):
        super().__init__()
        if scale not in (2, 4):
            raise NotImplementedError()
        self.scale = scale
        self.conv0 = nn.Conv3d(in_nc, nf, 3, 1, 0)
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 0)
        self.conv2 = nn.Conv2d(nf, nf * 2, 4, 2, 0)
        self.conv3 = nn.Conv2d(nf * 2, nf, 3, 1, 0) if scale == 2 else nn.Conv2d(nf * 2, nf, 4, 2, 0)
        self.conv4 = nn.Conv2d(nf, nf, 3, 1, 0)
        self.conv5 = nn.Conv3d(nf, nf, 3, 1, 0)
        self.conv6 = nn.Conv2d(nf, in_nc, 1, 1, 0)

    @staticmethod
    def _rep3(x):
        return F.pad(x, (1, 1, 1, 1, 1, 1), mode='replicate')

    @staticmethod
    def _ref2(x):
        return F.pad(x, (1, 1, 1, 1), mode='reflect')

    def forward(self, x):
        b, c, t, h, w = x.shape
        s = self.scale
        mean = x.mean(-1, keepdim=True).mean(-2, keepdim=True)
        y = _lrelu(conv3d_k3_replicate(x - mean, self.conv0.weight, self.conv0.bias))
        y = y.transpose(1, 2).reshape(b * t, -1, h, w)
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4):
            y = _lrelu(conv(self._ref2(y)))
        y = y.reshape(b, t, -1, h // s, w // s).transpose(1, 2)
        y = _lrelu(conv3d_k3_replicate(y, self.conv5.weight, self.conv5.bias))
        y = self.conv6(y.transpose(1, 2).reshape(b * t, -1, h // s, w // s))
        return y.reshape(b, t, -1, h // s, w // s).transpose(1, 2) + mean


class DirectKernelEstimator_CMS(nn.Module):
    """SFDN: single-frame estimator (x2), input N,3,H,W -> N,3,H/2,W/2."""

    def __init__(self, nf):
        super().__init__()
        self.conv0 = nn.Conv2d(3, nf, 3, 1, 0)
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 0)
        self.conv2 = nn.Conv2d(nf, nf, 3, 1, 0)
        self.conv3 = nn.Conv2d(nf, nf * 2, 4, 2, 0)
        self.conv4 = nn.Conv2d(nf * 2, nf * 2, 3, 1, 0)
        self.conv5 = nn.Conv2d(nf * 2, nf, 3, 1, 0)
        self.conv6 = nn.Conv2d(nf, 3, 1, 1, 0)

    def forward(self, x):
        mean = x.mean(2, keepdim=True).mean(3, keepdim=True)
        y = x - mean
        for conv in (self.conv0, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            y = _lrelu(conv(F.pad(y, (1, 1, 1, 1), mode='reflect')))
        return self.conv6(y) + mean
