"""TOFlow (Xue et al., IJCV 2018) on the native ops -- drop-in for codes/models/archs/TOF_arch.py.

Same modules, attribute paths and state-dict keys as the reference (``SpyNet.blocks.{0-3}.block.{0,1,3,4,...}``,
``conv_3x7_64_9x9``, ``conv_64_64_9x9``, ``conv_64_64_1x1``, ``conv_64_3_1x1``), so its checkpoints load with
strict=True; the torch.nn modules only HOLD the parameters -- every forward op is a native kernel
(dynavsr_amd/tofops.py): 7x7 / 9x9 / 1x1 convolutions with bias, ReLU and residual fused, BatchNorm2d + ReLU,
flow_warp, 2x2 average pooling, align_corners=True flow up-sampling, (de)normalisation.

MI355X-first differences to the reference's execution (results are the same):
  * the six neighbour frames go through SpyNet as ONE batch of 6B (the reference loops them in Python,
    TOF_arch.py:125-131; BatchNorm in training mode then sees the statistics of 6B samples instead of B six
    times -- the reference's per-call statistics are reproduced exactly with ``batch_neighbors=False``, the
    default, and the batched form is what eval mode uses since there the two coincide);
  * flows stay channel-first [N,2,H,W] (no permute);
  * torch.cat([ref, warped, flow]) is a channel-slice write, ``flow_up + block(...)`` the conv's residual input.
"""
import torch
import torch.nn as nn

from dynavsr_amd import _lib as L
from dynavsr_amd import tofops as T

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


class SpyNet_Block(nn.Module):
    """conv7(8->32) BN ReLU conv7(32->64) BN ReLU conv7(64->32) BN ReLU conv7(32->16) BN ReLU conv7(16->2)
    (TOF_arch.py:25-50)."""

    def __init__(self):
        super().__init__()
        chans = [(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)]
        layers = []
        for i, (ci, co) in enumerate(chans):
            layers.append(nn.Conv2d(ci, co, 7, 1, 3))
            if i < 4:
                layers += [nn.BatchNorm2d(co), nn.ReLU(inplace=True)]
        self.block = nn.Sequential(*layers)

    def forward(self, x, res=None):
        """x: [N,8,H,W] = [ref, warped neighbour, initial flow]; returns the flow update (+ ``res`` when given)."""
        for i in range(4):
            cv, bn = self.block[3 * i], self.block[3 * i + 1]
            x = T.batchnorm(T.conv(x, cv.weight, cv.bias), bn, relu=True)
        cv = self.block[12]
        return T.conv(x, cv.weight, cv.bias, res=res)


class SpyNet(nn.Module):
    """Coarse-to-fine flow over a 4-level average-pooled pyramid (TOF_arch.py:53-90)."""

    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([SpyNet_Block() for _ in range(4)])

    def forward(self, ref, nbr):
        """ref, nbr: [N,3,H,W] -> flow [N,2,H,W] that warps nbr onto ref."""
        n, c, h, w = ref.shape
        ref, nbr = [ref], [nbr]
        for _ in range(3):
            ref.insert(0, T.avg_pool2(ref[0]))
            nbr.insert(0, T.avg_pool2(nbr[0]))
        flow = ref[0].new_zeros((n, 2, h // 16, w // 16))
        for i in range(4):
            flow_up = T.resize_bilinear_ac(flow, nbr[i].shape[-2:], 2.0)
            x = torch.cat([ref[i], T.flow_warp(nbr[i], flow_up), flow_up], 1)
            flow = self.blocks[i](x, res=flow_up)
        return flow


class TOFlow(nn.Module):
    def __init__(self, adapt_official=False, batch_neighbors=False):
        super().__init__()
        self.SpyNet = SpyNet()
        self.conv_3x7_64_9x9 = nn.Conv2d(3 * 7, 64, 9, 1, 4)
        self.conv_64_64_9x9 = nn.Conv2d(64, 64, 9, 1, 4)
        self.conv_64_64_1x1 = nn.Conv2d(64, 64, 1)
        self.conv_64_3_1x1 = nn.Conv2d(64, 3, 1)
        self.relu = nn.ReLU(inplace=True)
        self.adapt_official = adapt_official  # True if using translated official weights else False
        self.batch_neighbors = batch_neighbors

    def _consts(self, x):
        mean = torch.tensor(_MEAN, device=x.device, dtype=torch.float32)
        std = torch.tensor(_STD, device=x.device, dtype=torch.float32)
        return mean, std

    def forward(self, x):
        """x: [B,7,3,H,W], H, W >= 16 -> [B,3,H,W].  Any size the reference takes: its pyramid floors (avg_pool2d), the
        flow starts as zeros of H//16 x W//16 and is resized to each level's own size (TOF_arch.py:69-90) -- the drivers feed
        180x320 (a 45x80 SLR clip x4), Vid4's 144x180, 22x22 patches."""
        if not x.is_cuda:
            raise RuntimeError("dynavsr_amd TOFlow runs on the MI355X only (input is on %s); there is no CPU fallback" % x.device)
        b, t, c, h, w = x.shape
        if t != 7 or c != 3:
            raise RuntimeError("TOFlow expects [B,7,3,H,W], got %s" % (tuple(x.shape),))
        if h < 16 or w < 16:
            raise RuntimeError("TOFlow: H=%d W=%d must be at least 16 (4-level SpyNet pyramid, flow of H//16 x W//16)" % (h, w))
        mean, std = self._consts(x)
        x = T.channel_affine(x.reshape(-1, c, h, w), 1.0 / std, -mean / std).view(b, t, c, h, w)   # normalize (:13-16)
        ref_idx = 3
        x_ref = x[:, ref_idx].contiguous()
        if self.adapt_official:          # the official weights take the reference frame first (:119-122)
            x = x[:, [3, 0, 1, 2, 4, 5, 6]]
            ref_idx = 0
        nbr_idx = [i for i in range(7) if i != ref_idx]
        batched = self.batch_neighbors or not self.training
        if batched:                      # one SpyNet pass over the 6B (reference, neighbour) pairs
            nbr = x[:, nbr_idx].reshape(-1, c, h, w)
            refs = x_ref[:, None].expand(b, 6, c, h, w).reshape(-1, c, h, w)
            warped = T.flow_warp(nbr, self.SpyNet(refs, nbr)).view(b, 6, c, h, w)
            frames = [None] * 7
            frames[ref_idx] = x_ref
            for k, i in enumerate(nbr_idx):
                frames[i] = warped[:, k]
        else:                            # the reference's loop: BatchNorm statistics per neighbour (:125-131)
            frames = []
            for i in range(7):
                if i == ref_idx:
                    frames.append(x_ref)
                else:
                    x_nbr = x[:, i].contiguous()
                    frames.append(T.flow_warp(x_nbr, self.SpyNet(x_ref, x_nbr)))
        y = torch.stack(frames, 1).view(b, -1, h, w)
        y = T.conv(y, self.conv_3x7_64_9x9.weight, self.conv_3x7_64_9x9.bias, act=L.ACT_RELU)
        y = T.conv(y, self.conv_64_64_9x9.weight, self.conv_64_64_9x9.bias, act=L.ACT_RELU)
        y = T.conv(y, self.conv_64_64_1x1.weight, self.conv_64_64_1x1.bias, act=L.ACT_RELU)
        y = T.conv(y, self.conv_64_3_1x1.weight, self.conv_64_3_1x1.bias, res=x_ref)
        return T.channel_affine(y, std, mean)                                                  # denormalize (:19-22)
