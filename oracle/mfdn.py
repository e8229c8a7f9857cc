"""Pure-torch functional restatement of MFDN (DirectKernelEstimatorVideo), TEST INFRASTRUCTURE.

Follows codes/models/archs/LRimg_estimator.py:92-117 and the layout handling of
codes/models/LRestimator_model.py:96-128 (feed_data transposes B,T,C,H,W -> B,C,T,H,W; the
output is transposed back).
"""
import torch
import torch.nn.functional as F


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def _c2(P, name, x, stride=1):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), P[name + ".weight"], P[name + ".bias"],
                    stride=stride)


def _c3(P, name, x):
    return F.conv3d(F.pad(x, (1, 1, 1, 1, 1, 1), mode="replicate"), P[name + ".weight"],
                    P[name + ".bias"])


def mfdn_forward(P, lqs, scale=4):
    """lqs: [B, T, 3, H, W] -> super-LR clip [B, T, 3, H/scale, W/scale]."""
    x = lqs.transpose(1, 2)                                   # LRestimator_model.py:103
    b, c, t, h, w = x.shape
    m = x.mean(-1, keepdim=True).mean(-2, keepdim=True)       # LRimg_estimator.py:99
    x = _lrelu(_c3(P, "conv0", x - m))
    fea = x.transpose(1, 2).reshape(b * t, -1, h, w)
    fea = _lrelu(_c2(P, "conv1", fea))
    fea = _lrelu(_c2(P, "conv2", fea, stride=2))
    fea = _lrelu(_c2(P, "conv3", fea, stride=2 if scale == 4 else 1))
    fea = _lrelu(_c2(P, "conv4", fea))
    hs, ws = h // scale, w // scale
    fea = fea.reshape(b, t, -1, hs, ws).transpose(1, 2)
    fea = _lrelu(_c3(P, "conv5", fea))
    fea = fea.transpose(1, 2).reshape(b * t, -1, hs, ws)
    fea = F.conv2d(fea, P["conv6.weight"], P["conv6.bias"])
    fea = fea.reshape(b, t, -1, hs, ws).transpose(1, 2)
    return (fea + m).transpose(1, 2)                          # LRestimator_model.py:128


def sfdn_forward(P, x):
    """SFDN (DirectKernelEstimator_CMS.forward, LRimg_estimator.py:55-67): x [N,3,H,W] -> [N,3,H/2,W/2]."""
    m = x.mean(2, keepdim=True).mean(3, keepdim=True)
    fea = _lrelu(_c2(P, "conv0", x - m))
    fea = _lrelu(_c2(P, "conv1", fea))
    fea = _lrelu(_c2(P, "conv2", fea))
    fea = _lrelu(_c2(P, "conv3", fea, stride=2))
    fea = _lrelu(_c2(P, "conv4", fea))
    fea = _lrelu(_c2(P, "conv5", fea))
    return F.conv2d(fea, P["conv6.weight"], P["conv6.bias"]) + m
