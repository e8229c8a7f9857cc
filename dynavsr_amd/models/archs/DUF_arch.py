"""DUF (Jo et al., CVPR 2018: dynamic upsampling filters, no explicit motion compensation) on the native ops --
drop-in for codes/models/archs/DUF_arch.py.

Same classes, attribute paths and state-dict keys as the reference (DUF_16L / DUF_28L / DUF_52L, DenseBlock,
DenseBlock_28L / _52L with their ``dense_blocks`` ModuleList), so its checkpoints load with strict=True; the
torch.nn modules only HOLD the parameters.  Execution (dynavsr_amd/tofops.py):
  * frames are the batch axis, [B*T,C,H,W]: Conv3d (1,3,3) and (1,1,1) are the MFMA conv2d kernels over the frames,
    BatchNorm3d(+ReLU) is the native BatchNorm over N = B*T;
  * Conv3d (3,3,3) = ``temporal_gather3`` (frames t-1, t, t+1 -> 3C channels; zero padding in time, or none in the
    T-reducing block) + a 3x3 conv2d whose weight is the Conv3d weight viewed as [Cout, 3C, 3, 3];
  * the tail -- softmax over the 25 filter taps, DynamicUpsamplingFilter_3C, the image residual with its
    adapt_official channel order, pixel_shuffle -- is ONE kernel (``dynamic_filter``); the reference materialises the
    [B,75,H,W] patch tensor, two permuted copies and the softmax.
"""
import torch
import torch.nn as nn

from dynavsr_amd import _lib as L
from dynavsr_amd import tofops as T


def _bn3(c):
    return nn.BatchNorm3d(c, eps=1e-3, momentum=1e-3)


def _conv1(x, m, act=L.ACT_NONE):
    """Conv3d (1,1,1) or (1,3,3) over frames-as-batch."""
    w = m.weight
    return T.conv(x, w.view(w.shape[0], w.shape[1], w.shape[3], w.shape[4]), m.bias, act=act)


def _conv333(x, m, b, t, pad_t):
    w = m.weight
    return T.conv(T.temporal_gather3(x, b, t, pad_t), w.view(w.shape[0], w.shape[1] * 3, 3, 3), m.bias)


def _crop_t(x, b, t):
    """x[:, :, 1:-1] of the reference's [B,C,T,H,W] in the frames-as-batch layout."""
    c, h, w = x.shape[1:]
    return x.view(b, t, c, h, w)[:, 1:-1].reshape(b * (t - 2), c, h, w)


class DenseBlock(nn.Module):
    """BN-ReLU-conv(1,1,1) / BN-ReLU-conv(3,3,3) x3 with dense concatenation (DUF_arch.py:32-83); t_reduce drops one
    frame at each end per (3,3,3) conv (7 -> 1)."""

    def __init__(self, nf=64, ng=32, t_reduce=False):
        super().__init__()
        self.t_reduce = t_reduce
        pad = (0, 1, 1) if t_reduce else (1, 1, 1)
        for i, c in enumerate((nf, nf + ng, nf + 2 * ng)):
            setattr(self, "bn3d_%d" % (2 * i + 1), _bn3(c))
            setattr(self, "conv3d_%d" % (2 * i + 1), nn.Conv3d(c, c, (1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), bias=True))
            setattr(self, "bn3d_%d" % (2 * i + 2), _bn3(c))
            setattr(self, "conv3d_%d" % (2 * i + 2), nn.Conv3d(c, ng, (3, 3, 3), stride=(1, 1, 1), padding=pad, bias=True))

    def forward(self, x, b, t):
        pad_t = 0 if self.t_reduce else 1
        for i in range(3):
            y = _conv1(T.batchnorm(x, getattr(self, "bn3d_%d" % (2 * i + 1)), relu=True), getattr(self, "conv3d_%d" % (2 * i + 1)))
            y = _conv333(T.batchnorm(y, getattr(self, "bn3d_%d" % (2 * i + 2)), relu=True),
                         getattr(self, "conv3d_%d" % (2 * i + 2)), b, t, pad_t)
            if self.t_reduce:
                x = _crop_t(x, b, t)
                t -= 2
            x = torch.cat((x, y), 1)
        return x, t


class _DenseStack(nn.Module):
    """DenseBlock_28L / _52L (:179-211, :289-320): `n` growth steps of BN-ReLU-conv(1,1,1)-BN-ReLU-conv(3,3,3), kept as
    the reference's flat ``dense_blocks`` ModuleList [BN, ReLU, Conv3d] x 2n (its state-dict keys)."""

    def __init__(self, nf, ng, n):
        super().__init__()
        layers = []
        for i in range(n):
            c = nf + i * ng
            layers += [_bn3(c), nn.ReLU(), nn.Conv3d(c, c, (1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), bias=True),
                       _bn3(c), nn.ReLU(), nn.Conv3d(c, ng, (3, 3, 3), stride=(1, 1, 1), padding=(1, 1, 1), bias=True)]
        self.dense_blocks = nn.ModuleList(layers)

    def forward(self, x, b, t):
        for i in range(0, len(self.dense_blocks), 6):
            y = _conv1(T.batchnorm(x, self.dense_blocks[i], relu=True), self.dense_blocks[i + 2])
            y = _conv333(T.batchnorm(y, self.dense_blocks[i + 3], relu=True), self.dense_blocks[i + 5], b, t, 1)
            x = torch.cat((x, y), 1)
        return x, t


class DenseBlock_28L(_DenseStack):
    def __init__(self, nf=64, ng=16):
        super().__init__(nf, ng, 9)


class DenseBlock_52L(_DenseStack):
    def __init__(self, nf=64, ng=16):
        super().__init__(nf, ng, 21)


class DynamicUpsamplingFilter_3C(nn.Module):
    """x [B,3,H,W], filters [B,25,R,H,W] -> [B,3R,H,W] (DUF_arch.py:86-110): the filters are applied AS GIVEN, whatever
    they sum to, and the gradient w.r.t. them is the plain one (the kernel's no-softmax mode).  Kept for the reference's
    module surface; DUF.forward uses the fused ``tofops.dynamic_filter`` (softmax folded in) instead."""

    def __init__(self, filter_size=(1, 5, 5)):
        super().__init__()
        if tuple(filter_size) != (1, 5, 5):
            raise NotImplementedError("DynamicUpsamplingFilter_3C: filter_size %s (the shipped networks use (1, 5, 5))" % (filter_size,))

    def forward(self, x, filters):
        b, nf, r, h, w = filters.shape
        s = int(round(r ** 0.5))
        out = T.dynamic_filter(x, filters.reshape(b, nf * r, h, w), x.new_zeros((b, 3 * r, h, w)), s, False,
                               taps_given=True)                                 # [B,3,sH,sW] pixel-shuffled
        return torch.nn.functional.pixel_unshuffle(out, s)


class _DUF(nn.Module):
    def __init__(self, scale, adapt_official, first, c1, ng2, c_out2):
        super().__init__()
        self.conv3d_1 = nn.Conv3d(3, 64, (1, 3, 3), stride=(1, 1, 1), padding=(0, 1, 1), bias=True)
        self.dense_block_1 = first
        self.dense_block_2 = DenseBlock(c1, ng2, t_reduce=True)
        self.bn3d_2 = _bn3(c_out2)
        self.conv3d_2 = nn.Conv3d(c_out2, 256, (1, 3, 3), stride=(1, 1, 1), padding=(0, 1, 1), bias=True)
        self.conv3d_r1 = nn.Conv3d(256, 256, (1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), bias=True)
        self.conv3d_r2 = nn.Conv3d(256, 3 * (scale ** 2), (1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), bias=True)
        self.conv3d_f1 = nn.Conv3d(256, 512, (1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), bias=True)
        self.conv3d_f2 = nn.Conv3d(512, 1 * 5 * 5 * (scale ** 2), (1, 1, 1), stride=(1, 1, 1), padding=(0, 0, 0), bias=True)
        self.dynamic_filter = DynamicUpsamplingFilter_3C((1, 5, 5))
        self.scale = scale
        self.adapt_official = adapt_official

    def forward(self, x):
        """x [B,7,3,H,W] -> [B,3,scale*H,scale*W] (DUF_arch.py:138-176)."""
        if not x.is_cuda:
            raise RuntimeError("dynavsr_amd DUF runs on the MI355X only (input is on %s); there is no CPU fallback" % x.device)
        b, t, c, h, w = x.shape
        if t != 7 or c != 3:
            raise RuntimeError("DUF expects [B,7,3,H,W], got %s" % (tuple(x.shape),))
        x = x.float().contiguous()
        x_center = x[:, t // 2].contiguous()
        y = _conv1(x.view(b * t, c, h, w), self.conv3d_1)
        y, t = self.dense_block_1(y, b, t)
        y, t = self.dense_block_2(y, b, t)               # T: 7 -> 1
        y = _conv1(T.batchnorm(y, self.bn3d_2, relu=True), self.conv3d_2, act=L.ACT_RELU)
        rx = _conv1(_conv1(y, self.conv3d_r1, act=L.ACT_RELU), self.conv3d_r2)     # image residual [B,3R,H,W]
        fx = _conv1(_conv1(y, self.conv3d_f1, act=L.ACT_RELU), self.conv3d_f2)     # filter logits [B,25R,H,W]
        return T.dynamic_filter(x_center, fx, rx, self.scale, self.adapt_official)


class DUF_16L(_DUF):
    """Official DUF structure with 16 layers (DUF_arch.py:113-176)."""

    def __init__(self, scale=4, adapt_official=False):
        super().__init__(scale, adapt_official, DenseBlock(64, 64 // 2, t_reduce=False), 160, 64 // 2, 256)


class DUF_28L(_DUF):
    def __init__(self, scale=4, adapt_official=False):
        super().__init__(scale, adapt_official, DenseBlock_28L(64, 16), 208, 16, 256)


class DUF_52L(_DUF):
    def __init__(self, scale=4, adapt_official=False):
        super().__init__(scale, adapt_official, DenseBlock_52L(64, 16), 400, 16, 448)
