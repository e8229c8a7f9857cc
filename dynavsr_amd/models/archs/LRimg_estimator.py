"""Down-scaling estimators MFDN / SFDN (state-dict compatible with
codes/models/archs/LRimg_estimator.py:38-117).

The modules only hold the parameters under the reference's names (conv0..conv6, so reference
`*_E.pth` checkpoints load with strict=True); forward and the parameter gradients run as one
native launch tape (`dvsr_estimator_*`, csrc/engine.hip + pad.hip) on the MI355X.  There is no CPU
path: a CPU input raises.
"""
import torch.nn as nn

from dynavsr_amd import engine


class DirectKernelEstimatorVideo(nn.Module):
    """MFDN: multi-frame estimator, input B,C,T,H,W -> B,C,T,H/s,W/s (LRimg_estimator.py:70-117)."""

    def __init__(self, nf, in_nc=3, scale=2):
        super().__init__()
        if scale not in (2, 4):
            raise NotImplementedError()
        self.nf, self.in_nc, self.scale = nf, in_nc, scale
        self.conv0 = nn.Conv3d(in_nc, nf, 3, 1, 0)
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 0)
        self.conv2 = nn.Conv2d(nf, nf * 2, 4, 2, 0)
        self.conv3 = nn.Conv2d(nf * 2, nf, 3, 1, 0) if scale == 2 else nn.Conv2d(nf * 2, nf, 4, 2, 0)
        self.conv4 = nn.Conv2d(nf, nf, 3, 1, 0)
        self.conv5 = nn.Conv3d(nf, nf, 3, 1, 0)
        self.conv6 = nn.Conv2d(nf, in_nc, 1, 1, 0)

    def ordered_parameters(self):
        return [p for conv in (self.conv0, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6)
                for p in (conv.weight, conv.bias)]

    def forward(self, x):
        cfg = (engine.MFDN, self.nf, self.in_nc, self.scale, x.shape[2])
        return engine.EstimatorFunction.apply(x, cfg, *self.ordered_parameters())

    def forward_stacked(self, x, stacked, per_slice=False):
        """K clips with per-clip parameter gradients: `stacked` = the parameters as [K, *shape] tensors; per_slice=False:
        equal slices, slice 0 is read; True: clip k runs on slice k (engine.EstimatorStackedFunction)."""
        cfg = (engine.MFDN, self.nf, self.in_nc, self.scale, x.shape[2])
        return engine.EstimatorStackedFunction.apply(x, cfg, bool(per_slice), *stacked)


class DirectKernelEstimator_CMS(nn.Module):
    """SFDN: single-frame estimator (x2), input N,3,H,W -> N,3,H/2,W/2 (LRimg_estimator.py:38-67)."""

    def __init__(self, nf):
        super().__init__()
        self.nf = nf
        self.conv0 = nn.Conv2d(3, nf, 3, 1, 0)
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 0)
        self.conv2 = nn.Conv2d(nf, nf, 3, 1, 0)
        self.conv3 = nn.Conv2d(nf, nf * 2, 4, 2, 0)
        self.conv4 = nn.Conv2d(nf * 2, nf * 2, 3, 1, 0)
        self.conv5 = nn.Conv2d(nf * 2, nf, 3, 1, 0)
        self.conv6 = nn.Conv2d(nf, 3, 1, 1, 0)

    def ordered_parameters(self):
        return [p for conv in (self.conv0, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6)
                for p in (conv.weight, conv.bias)]

    def forward(self, x):
        cfg = (engine.SFDN, self.nf, 3, 2, 1)
        return engine.EstimatorFunction.apply(x, cfg, *self.ordered_parameters())

    def forward_stacked(self, x, stacked, per_slice=False):
        cfg = (engine.SFDN, self.nf, 3, 2, 1)
        return engine.EstimatorStackedFunction.apply(x, cfg, bool(per_slice), *stacked)
