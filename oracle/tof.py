"""Functional restatement of TOFlow (TEST INFRASTRUCTURE, CPU): codes/models/archs/TOF_arch.py:13-140 and
arch_util.flow_warp (:55-79) over a plain state dict, so that tests and goldens need no nn.Module.

Pinned by tests/golden/tof_*.npz, which oracle/gen_golden.py produces by running the REFERENCE's own TOFlow module
on CPU (it has no custom op) in training and in eval mode and asserting this file against it.
BatchNorm follows nn.BatchNorm2d: training -> batch statistics (biased variance) and an in-place update of the
running estimates passed in ``P`` (momentum 0.1, unbiased variance); eval -> the running estimates.
"""
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def normalize(x):
    m = torch.tensor(MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    s = torch.tensor(STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - m) / s


def denormalize(x):
    m = torch.tensor(MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    s = torch.tensor(STD, dtype=x.dtype).view(1, 3, 1, 1)
    return x * s + m


def flow_warp(x, flow):
    """arch_util.py:55-79; flow [N,H,W,2] in pixels; grid normalised by (size - 1), grid_sample with its default
    align_corners=False (torch >= 1.3), zeros padding."""
    n, c, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
    grid = torch.stack((gx, gy), 2).to(x.dtype)
    v = grid + flow
    vx = 2.0 * v[..., 0] / max(w - 1, 1) - 1.0
    vy = 2.0 * v[..., 1] / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((vx, vy), 3), mode="bilinear", padding_mode="zeros", align_corners=False)


def _bn(P, pre, x, training):
    return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"], P[pre + ".bias"],
                        training, 0.1, 1e-5)


def spynet_block(P, pre, x, training):
    for i in range(4):
        x = F.conv2d(x, P["%s.block.%d.weight" % (pre, 3 * i)], P["%s.block.%d.bias" % (pre, 3 * i)], padding=3)
        x = F.relu(_bn(P, "%s.block.%d" % (pre, 3 * i + 1), x, training))
    return F.conv2d(x, P[pre + ".block.12.weight"], P[pre + ".block.12.bias"], padding=3)


def spynet(P, ref, nbr, training, taps=None):
    n, c, h, w = ref.shape
    ref, nbr = [ref], [nbr]
    for _ in range(3):
        ref.insert(0, F.avg_pool2d(ref[0], 2, 2, count_include_pad=False))
        nbr.insert(0, F.avg_pool2d(nbr[0], 2, 2, count_include_pad=False))
    flow = ref[0].new_zeros((n, 2, h // 16, w // 16))
    for i in range(4):
        up = F.interpolate(flow, size=nbr[i].shape[-2:], mode="bilinear", align_corners=True) * 2.0
        flow = up + spynet_block(P, "SpyNet.blocks.%d" % i,
                                 torch.cat([ref[i], flow_warp(nbr[i], up.permute(0, 2, 3, 1)), up], 1), training)
        if taps is not None:
            taps.setdefault("flow_l%d" % i, []).append(flow)
    return flow


def toflow_forward(P, x, adapt_official=True, training=False, taps=None):
    """x [B,7,3,H,W] -> [B,3,H,W].  In training mode the running estimates inside ``P`` are updated in place, six
    times per BatchNorm layer (one SpyNet call per neighbour), like the module."""
    b, t, c, h, w = x.shape
    x = normalize(x.reshape(-1, c, h, w)).view(b, t, c, h, w)
    ref_idx = 3
    x_ref = x[:, ref_idx]
    if adapt_official:
        x = x[:, [3, 0, 1, 2, 4, 5, 6]]
        ref_idx = 0
    frames = []
    for i in range(7):
        if i == ref_idx:
            frames.append(x_ref)
        else:
            nbr = x[:, i]
            flow = spynet(P, x_ref, nbr, training, taps)
            frames.append(flow_warp(nbr, flow.permute(0, 2, 3, 1)))
    y = torch.stack(frames, 1).view(b, -1, h, w)
    if taps is not None:
        taps["warped"] = y
    y = F.relu(F.conv2d(y, P["conv_3x7_64_9x9.weight"], P["conv_3x7_64_9x9.bias"], padding=4))
    y = F.relu(F.conv2d(y, P["conv_64_64_9x9.weight"], P["conv_64_64_9x9.bias"], padding=4))
    y = F.relu(F.conv2d(y, P["conv_64_64_1x1.weight"], P["conv_64_64_1x1.bias"]))
    y = F.conv2d(y, P["conv_64_3_1x1.weight"], P["conv_64_3_1x1.bias"]) + x_ref
    return denormalize(y)
