"""Deterministic synthetic weights and clips (SURVEY.md §8d "Config 1/2").

There are no datasets or checkpoints in the build environment, so benchmarks, parity tests and
golden vectors all use tensors generated here.  ``numpy.random.RandomState`` (the frozen legacy
MT19937 stream) is used rather than ``torch.manual_seed`` so that the same seed gives bit-identical
tensors across torch versions and devices; fixtures therefore store seeds, not 13 MB of weights.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

from .spec import edvr_param_spec, mfdn_param_spec, sfdn_param_spec, tof_param_spec


def _rs(seed, name):
    return np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))


def _fan_in(shape):
    f = 1
    for d in shape[1:]:
        f *= d
    return f


def edvr_state_dict(seed=0, offset_std=0.05, dtype=torch.float32, **cfg):
    """Kaiming-normal weights (x0.1 inside residual blocks, as arch_util.py:46 does), small random
    biases, and conv_offset_mask ~ N(0, offset_std^2) so that the DCN offsets are non-trivial
    (the reference zero-initialises them, deform_conv.py:270-272, which would not exercise sampling).
    """
    sd = OrderedDict()
    for name, shape in edvr_param_spec(**cfg).items():
        r = _rs(seed, name)
        if "conv_offset_mask" in name:
            a = r.standard_normal(shape) * offset_std
        elif name.endswith(".bias"):
            a = r.standard_normal(shape) * 0.01
        else:
            gain = 0.1 if (".conv1." in name or ".conv2." in name) else 1.0
            a = r.standard_normal(shape) * (gain * np.sqrt(2.0 / _fan_in(shape)))
        sd[name] = torch.from_numpy(a).to(dtype)
    return sd


def mfdn_state_dict(seed=0, dtype=torch.float32, **cfg):
    sd = OrderedDict()
    for name, shape in mfdn_param_spec(**cfg).items():
        r = _rs(seed + 7919, name)
        if name.endswith(".bias"):
            a = r.standard_normal(shape) * 0.01
        else:
            a = r.standard_normal(shape) * np.sqrt(2.0 / _fan_in(shape))
        sd[name] = torch.from_numpy(a).to(dtype)
    return sd


def sfdn_state_dict(seed=0, dtype=torch.float32, **cfg):
    sd = OrderedDict()
    for name, shape in sfdn_param_spec(**cfg).items():
        r = _rs(seed + 104729, name)
        if name.endswith(".bias"):
            a = r.standard_normal(shape) * 0.01
        else:
            a = r.standard_normal(shape) * np.sqrt(2.0 / _fan_in(shape))
        sd[name] = torch.from_numpy(a).to(dtype)
    return sd


def tof_state_dict(seed=0, dtype=torch.float32):
    """TOFlow weights: Kaiming-normal convolutions (the last conv of every SpyNet block x0.2, so that the flows stay
    within a few pixels on [0,1] images), BatchNorm gamma ~ 1 +- 0.1, beta ~ +-0.05, running_mean ~ N(0, 0.1^2),
    running_var ~ U[0.5, 1.5]: eval mode then differs from training mode (batch statistics)."""
    sd = OrderedDict()
    for name, shape in tof_param_spec().items():
        r = _rs(seed + 15485863, name)
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        if name.endswith("running_mean"):
            a = r.standard_normal(shape) * 0.1
        elif name.endswith("running_var"):
            a = r.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and ".block." in name and int(name.split(".block.")[1].split(".")[0]) % 3 == 1:
            a = (1.0 + 0.1 * r.standard_normal(shape)) if name.endswith(".weight") else 0.05 * r.standard_normal(shape)
        elif name.endswith(".bias"):
            a = r.standard_normal(shape) * 0.01
        else:
            gain = 0.2 if ".block.12." in name else 1.0
            a = r.standard_normal(shape) * (gain * np.sqrt(2.0 / _fan_in(shape)))
        sd[name] = torch.from_numpy(np.asarray(a)).to(dtype)
    return sd


def duf_state_dict(seed=0, layers=16, scale=4, dtype=torch.float32):
    """DUF weights by the names and shapes of the module's own state dict (DUF_arch.py; 101 / 185 / 353 tensors):
    Kaiming-normal convolutions (the two output convolutions x0.3 so that the filter logits and the residual stay
    O(1)), BatchNorm gamma ~ 1 +- 0.1, beta ~ +-0.05, running_mean ~ N(0, 0.1^2), running_var ~ U[0.5, 1.5]."""
    from .models.archs import DUF_arch
    cls = {16: DUF_arch.DUF_16L, 28: DUF_arch.DUF_28L}.get(layers, DUF_arch.DUF_52L)
    spec = OrderedDict((k, tuple(v.shape)) for k, v in cls(scale=scale, adapt_official=True).state_dict().items())
    sd = OrderedDict()
    for name, shape in spec.items():
        r = _rs(seed + 32452843, name)
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        if name.endswith("running_mean"):
            a = r.standard_normal(shape) * 0.1
        elif name.endswith("running_var"):
            a = r.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and (name + "@").replace(".weight@", ".running_var").replace(".bias@", ".running_var") in spec:
            a = (1.0 + 0.1 * r.standard_normal(shape)) if name.endswith(".weight") else 0.05 * r.standard_normal(shape)
        elif name.endswith(".bias"):
            a = r.standard_normal(shape) * 0.01
        else:
            gain = 0.3 if name.startswith(("conv3d_r2", "conv3d_f2")) else 1.0
            a = r.standard_normal(shape) * (gain * np.sqrt(2.0 / _fan_in(shape)))
        sd[name] = torch.from_numpy(np.asarray(a)).to(dtype)
    return sd


def clip(seed, b, n, h, w, dtype=torch.float32, smooth=True):
    """A [b, n, 3, h, w] clip in [0,1].  ``smooth`` gives low-frequency content with a global
    per-frame shift (so alignment has something to align); otherwise i.i.d. U[0,1)."""
    r = np.random.RandomState(seed)
    if not smooth:
        return torch.from_numpy(r.random_sample((b, n, 3, h, w))).to(dtype)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    out = np.zeros((b, n, 3, h, w))
    for bi in range(b):
        comps = [(r.uniform(0.02, 0.6), r.uniform(0.02, 0.6), r.uniform(0, 6.28), r.uniform(0.3, 1))
                 for _ in range(6)]
        dx, dy = r.uniform(-1.5, 1.5, 2)
        for t in range(n):
            for c in range(3):
                img = np.zeros((h, w))
                for fy, fx, ph, amp in comps:
                    img += amp * np.sin(fy * (yy + dy * t) + fx * (xx + dx * t) + ph + c)
                out[bi, t, c] = img
    out = (out - out.min()) / (out.max() - out.min() + 1e-12)
    out = 0.9 * out + 0.1 * r.random_sample(out.shape)
    return torch.from_numpy(out).to(dtype)


def sr_pair(seed, n, h, w, scale=4, blur=7, dtype=torch.float32):
    """A synthetic super-resolution PAIR at a realistic operating point: a smooth HR clip [1, n, 3, scale h, scale w] (the seeded
    clip, box-blurred `blur` x `blur` with reflection: the detail a x`scale` network can plausibly restore), its area-downscaled
    LR clip [1, n, 3, h, w] and the HR centre frame as ground truth [3, scale h, scale w].  Bilinear up-sampling of the LR
    centre frame -- what EDVR adds its residual to (EDVR_arch.py:311-312) -- scores 30.1 dB PSNR against the ground truth at
    blur = 7, the range of a trained video SR network on REDS; the accuracy gates of the reduced-precision modes are taken
    there (with the residual branch damped: damp_residual_branch) instead of on unrelated noise images at 8 dB."""
    import torch.nn.functional as F
    hr = clip(seed, 1, n, scale * h, scale * w, dtype=torch.float32)[0]
    if blur > 1:
        hr = F.avg_pool2d(F.pad(hr, (blur // 2,) * 4, mode="reflect"), blur, 1)
    lr = F.avg_pool2d(hr, scale)
    return lr.unsqueeze(0).to(dtype).contiguous(), hr[n // 2].to(dtype).contiguous()


def damp_residual_branch(sd, gain=0.02):
    """conv_last's weight and bias times `gain`: a randomly initialised EDVR adds O(1) noise to the up-sampled base frame
    (8 dB PSNR against any target); a trained one adds a small correction.  With the branch damped the network sits at the
    operating point of a trained model (sr_pair: ~30 dB), where an arithmetic perturbation of the trunk is weighed against
    the ground-truth error the way north_star's 0.02 dB gate means it."""
    sd = type(sd)((k, v.clone()) for k, v in sd.items())
    for k in ("conv_last.weight", "conv_last.bias"):
        sd[k] = sd[k] * gain
    return sd
