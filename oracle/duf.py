"""Functional restatement of DUF (TEST INFRASTRUCTURE, CPU): codes/models/archs/DUF_arch.py:17-388 over a plain state
dict, in the reference's own [B,C,T,H,W] layout with F.conv3d / F.batch_norm.

Pinned by tests/golden/duf_*.npz, which oracle/gen_golden.py produces by running the REFERENCE's DUF modules on CPU
(plain torch) and asserting this file against them.  BatchNorm3d: eps 1e-3, momentum 1e-3 (:41); training -> batch
statistics and an in-place update of the running estimates inside ``P``.
"""
import torch
import torch.nn.functional as F


def _bn(P, pre, x, training):
    return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"], P[pre + ".bias"],
                        training, 1e-3, 1e-3)


def _c(P, pre, x, pad):
    return F.conv3d(x, P[pre + ".weight"], P[pre + ".bias"], padding=pad)


def dense_block(P, pre, x, t_reduce, training):
    pad = (0, 1, 1) if t_reduce else (1, 1, 1)
    for i in range(3):
        y = _c(P, "%s.conv3d_%d" % (pre, 2 * i + 1), F.relu(_bn(P, "%s.bn3d_%d" % (pre, 2 * i + 1), x, training)), 0)
        y = _c(P, "%s.conv3d_%d" % (pre, 2 * i + 2), F.relu(_bn(P, "%s.bn3d_%d" % (pre, 2 * i + 2), y, training)), pad)
        x = torch.cat((x[:, :, 1:-1] if t_reduce else x, y), 1)
    return x


def dense_stack(P, pre, x, n, training):
    for i in range(n):
        k = 6 * i
        y = _c(P, "%s.dense_blocks.%d" % (pre, k + 2), F.relu(_bn(P, "%s.dense_blocks.%d" % (pre, k), x, training)), 0)
        y = _c(P, "%s.dense_blocks.%d" % (pre, k + 5), F.relu(_bn(P, "%s.dense_blocks.%d" % (pre, k + 3), y, training)), (1, 1, 1))
        x = torch.cat((x, y), 1)
    return x


def dynamic_filter_3c(x, filters):
    """DynamicUpsamplingFilter_3C (:86-110): the 5x5 patch of each colour plane times the per-pixel filters."""
    b, nf, r, h, w = filters.shape
    patches = F.unfold(x, 5, padding=2).view(b, 3, 25, h, w)             # [B,3,25,H,W]: patch tap f = (dy, dx) row-major
    return torch.einsum("bcfhw,bfrhw->bcrhw", patches, filters).reshape(b, 3 * r, h, w)


def duf_forward(P, x, layers=16, scale=4, adapt_official=True, training=False, taps=None):
    """x [B,7,3,H,W] -> [B,3,scale*H,scale*W]."""
    b, t, c, h, w = x.shape
    x = x.permute(0, 2, 1, 3, 4)
    x_center = x[:, :, t // 2]
    y = _c(P, "conv3d_1", x, (0, 1, 1))
    if layers == 16:
        y = dense_block(P, "dense_block_1", y, False, training)
    else:
        y = dense_stack(P, "dense_block_1", y, 9 if layers == 28 else 21, training)
    y = dense_block(P, "dense_block_2", y, True, training)
    y = F.relu(_c(P, "conv3d_2", F.relu(_bn(P, "bn3d_2", y, training)), (0, 1, 1)))
    rx = _c(P, "conv3d_r2", F.relu(_c(P, "conv3d_r1", y, 0)), 0)
    fx = _c(P, "conv3d_f2", F.relu(_c(P, "conv3d_f1", y, 0)), 0)
    fx = F.softmax(fx.view(b, 25, scale ** 2, h, w), dim=1)
    if adapt_official:                       # :17-29, written without the in-place aliasing
        r = scale ** 2
        rx = torch.cat((rx[:, 0::3], rx[:, 1::3], rx[:, 2::3]), 1)
        assert rx.shape[1] == 3 * r
    if taps is not None:
        taps["features"], taps["rx"], taps["fx"] = y, rx, fx
    out = dynamic_filter_3c(x_center, fx) + rx.squeeze(2)
    return F.pixel_shuffle(out, scale)
