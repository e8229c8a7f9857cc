"""EDVR backbone behind the reference's constructor / state-dict contract, executed by the
native MI355X engine.

Mirrors the interface of codes/models/archs/EDVR_arch.py:206-313 (class EDVR): same ctor
arguments, same parameter names/shapes (144 tensors for EDVR-M, dynavsr_amd/spec.py), same
forward signature ``netG(x[B,N,3,H,W]) -> [B,3,sH,sW]``.  There are no torch.nn compute layers:
the module tree below only *holds* the parameters under the reference's attribute paths, and
``forward`` hands them to csrc/engine.hip in one call.

Not supported (never enabled by a shipped YAML): predeblur, HR_in, w_TSA=False.
"""
import math

import torch
import torch.nn as nn

from dynavsr_amd.engine import EdvrFunction, EdvrStackedFunction
from dynavsr_amd.spec import edvr_param_spec


class ParamHolder(nn.Module):
    """A node of the parameter tree; children/parameters are attached by name."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the network runs in the native engine, call EDVR(x)")


def _attach(root, dotted, tensor):
    node = root
    parts = dotted.split(".")
    for part in parts[:-1]:
        if part not in node._modules:
            node.add_module(part, ParamHolder())
        node = node._modules[part]
    node.register_parameter(parts[-1], nn.Parameter(tensor))


def _init_tensor(name, shape):
    """Same initial distribution family as the reference: nn.Conv2d default (kaiming-uniform,
    a=sqrt(5)) everywhere, kaiming-normal x0.1 + zero bias inside residual blocks
    (arch_util.py:7-24,46), uniform(+-1/sqrt(fan_in)) + zero bias for DCN weights
    (deform_conv.py:245-252), zeros for conv_offset_mask (deform_conv.py:270-272)."""
    t = torch.empty(shape)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    is_w = name.endswith(".weight")
    if "conv_offset_mask" in name:
        return t.zero_()
    if "dcnpack" in name:
        return t.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in)) if is_w else t.zero_()
    if ("feature_extraction." in name or "recon_trunk." in name):
        return t.normal_(0, math.sqrt(2.0 / fan_in)).mul_(0.1) if is_w else t.zero_()
    if is_w:
        return nn.init.kaiming_uniform_(t, a=math.sqrt(5))
    return t  # bias: filled below from the matching weight's fan_in


class EDVR(nn.Module):
    def __init__(self, nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, center=None,
                 predeblur=False, HR_in=False, w_TSA=True, scale=4, bf16_mfma=False):
        """bf16_mfma (not a reference option): run the 3x3 stride-1 convolutions on the bf16 MFMA with fp32
        accumulation (network_G.bf16_mfma in the YAML; BASELINE configs[4])."""
        super().__init__()
        self.bf16_mfma = int(bf16_mfma)  # 0 fp32 MFMA | 1 bf16 operands | 2 three-way bf16 split (experimental)
        if predeblur or HR_in or not w_TSA:
            raise NotImplementedError("dynavsr_amd EDVR supports predeblur=False, HR_in=False, "
                                      "w_TSA=True (the only configuration DynaVSR ships)")
        self.nf, self.nframes, self.groups = nf, nframes, groups
        self.front_RBs, self.back_RBs, self.scale = front_RBs, back_RBs, scale
        self.center = nframes // 2 if center is None else center
        spec = edvr_param_spec(nf, nframes, groups, front_RBs, back_RBs, scale)
        self._names = list(spec.keys())
        for name, shape in spec.items():
            t = _init_tensor(name, shape)
            if name.endswith(".bias") and "dcnpack" not in name and "feature_extraction." not in name \
                    and "recon_trunk." not in name:
                wshape = spec[name[:-5] + ".weight"]
                fan_in = wshape[1] * wshape[2] * wshape[3]
                t.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            _attach(self, name, t)
        self._debug_ws = None  # set to a list to capture (plan, workspace) of each forward

    def _cfg(self):
        return (self.nf, self.nframes, self.groups, self.front_RBs, self.back_RBs, self.scale,
                self.center, int(self.bf16_mfma))

    def ordered_parameters(self):
        d = dict(self.named_parameters())
        return [d[n] for n in self._names]

    def forward(self, x):
        return EdvrFunction.apply(x, self._cfg(), self._debug_ws, *self.ordered_parameters())

    def forward_stacked(self, x, stacked, per_slice=False):
        """K clips [K,N,3,H,W] as one batch with PER-CLIP parameter gradients: `stacked` = this network's parameters
        (ordered_parameters order) as [K, *shape] leaf tensors -- the private copies of K frames.  per_slice=False: the K
        slices are equal (before the first inner step), slice 0 is read; True: clip k runs on slice k
        (engine.EdvrStackedFunction)."""
        return EdvrStackedFunction.apply(x, self._cfg(), bool(per_slice), *stacked)
