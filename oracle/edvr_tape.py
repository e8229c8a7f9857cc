"""EDVR forward written in the ORDER OF THE NATIVE LAUNCH TAPE (TEST INFRASTRUCTURE, CPU).

Same arithmetic as oracle/edvr.py (the reference-pinned restatement of EDVR_arch.py:95-313) -- checked against it by
tests/test_oracle_edvr_tape.py -- but with the N frames batched and one torch op per launch of csrc/engine.hip:build_plan,
in that order and under the launches' names, so that every launch output of a GPU forward can be put in the place of
the oracle's own value ("teacher forcing"):

    y  <-  y + (y_gpu - y).detach()          (value of the GPU, gradient of the oracle)

With every launch output forced, each non-smooth decision of the backward -- the sign an (L)ReLU backward keys on, the
arg-max of the 3x3 max pools, the floor() cells and the (-1, H) x (-1, W) gate of the deformable sampler -- is taken on
exactly the activations the GPU kernels saw, and the oracle's backward is the exact (fp64) linearisation of the GPU's own
forward: the comparison with dvsr_edvr_backward is free of the kink flips that make two correct fp32 evaluations of this
network differ by ~1e-3 in a gradient tensor, and can be held to summation-order round-off.
"""
import torch
import torch.nn.functional as F

from . import dcn as _dcn


class _ForcedAct(torch.autograd.Function):
    """(L)ReLU whose VALUE is the GPU's output and whose backward keys on the sign of that output, as the GPU's activation
    backward does (it masks by the stored, activated tensor): slope = 0.1 (LeakyReLU) or 0 (ReLU) where y_gpu <= 0."""

    @staticmethod
    def forward(ctx, pre, y_gpu, slope):
        ctx.save_for_backward(y_gpu > 0)
        ctx.slope = slope
        return y_gpu.to(pre.dtype).clone()

    @staticmethod
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        return torch.where(pos, g, g * ctx.slope), None, None


class Tape:
    """Collects the launch names in launch order; `force(index, name, which, value)` may return the GPU tensor that takes
    the place of `value` (same shape) or None."""

    def __init__(self, force=None):
        self.names = []
        self.force = force

    def forced(self, name, which, like):
        return None if self.force is None else self.force(len(self.names), name, which, like)

    @staticmethod
    def _sub(y, f):
        return y if f is None else y + (f.to(y.dtype) - y).detach()

    def out(self, name, y, y2=None):
        f = self.forced(name, 0, y)
        f2 = self.forced(name, 1, y2) if y2 is not None else None
        self.names.append(name)
        return self._sub(y, f) if y2 is None else (self._sub(y, f), self._sub(y2, f2))


def _act(y, act):
    return F.leaky_relu(y, 0.1) if act == "L" else (F.relu(y) if act == "R" else y)


def _forced_act(t, name, pre, act):
    """act(pre) -- with the GPU's launch output in hand: its value, and its sign as the backward's mask."""
    f = t.forced(name, 0, pre) if act != "N" else None
    if f is None:
        return _act(pre, act)
    return _ForcedAct.apply(pre, f, 0.1 if act == "L" else 0.0)


def _conv(t, P, name, pname, x0, x1=None, stride=1, act="N", res=None, ps=0, y_add=None):
    """One conv launch: y = act(conv(cat(x0, x1)) + b) [pixel-shuffled] [+ res] (engine.hip Builder::conv; a launch has a
    residual or an activation, never both)."""
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    w = P[pname + ".weight"]
    y = F.conv2d(x, w, P[pname + ".bias"], stride=stride, padding=w.shape[-1] // 2)
    if ps:
        y = F.pixel_shuffle(y, ps)
    assert act == "N" or (res is None and y_add is None)
    y = _forced_act(t, name, y, act)
    if res is not None:
        y = y + res
    if y_add is not None:
        y = y + y_add
    return t.out(name, y)


def _up(t, name, x, s=2, mul=1.0):
    y = F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
    return t.out(name, y * mul if mul != 1.0 else y)


def _dcn_launch(t, P, name, pname, x, om, groups, act):
    n_off = groups * 2 * 9
    offset, mask = om[:, :n_off], torch.sigmoid(om[:, n_off:])
    y = _dcn.modulated_deform_conv(x, offset.contiguous(), mask.contiguous(), P[pname + ".weight"], P[pname + ".bias"],
                                   1, 1, 1, 1, groups)
    return t.out(name, _forced_act(t, name, y, act))


def edvr_forward_tape(P, x, nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, scale=4, center=None, force=None):
    """x: [B, N, 3, H, W] -> ([B, 3, sH, sW], launch names).  Launch order and names = build_plan (engine.hip)."""
    t = Tape(force)
    b, n, c, h, w = x.shape
    ctr = n // 2 if center is None else center
    C = nf
    xin = x.reshape(b * n, c, h, w)

    def ref_of(f):   # the reference frame of every clip, broadcast over its N frames (the tape's x1_bdiv view)
        return f.view(b, n, *f.shape[1:])[:, ctr].repeat_interleave(n, 0)

    f1 = _conv(t, P, "conv_first", "conv_first", xin, act="L")
    for i in range(front_RBs):
        a = _conv(t, P, "fe_rb_a", "feature_extraction.%d.conv1" % i, f1, act="R")
        f1 = _conv(t, P, "fe_rb_b", "feature_extraction.%d.conv2" % i, a, res=f1)
    f2 = _conv(t, P, "fea_L2_conv1", "fea_L2_conv1", f1, stride=2, act="L")
    f2 = _conv(t, P, "L2_fea", "fea_L2_conv2", f2, act="L")
    f3 = _conv(t, P, "fea_L3_conv1", "fea_L3_conv1", f2, stride=2, act="L")
    f3 = _conv(t, P, "L3_fea", "fea_L3_conv2", f3, act="L")
    p = "pcd_align."
    # L3
    o3 = _conv(t, P, "L3_offset_conv1", p + "L3_offset_conv1", f3, ref_of(f3), act="L")
    o3 = _conv(t, P, "L3_offset", p + "L3_offset_conv2", o3, act="L")
    om3 = _conv(t, P, "L3_om", p + "L3_dcnpack.conv_offset_mask", o3)
    fe3 = _dcn_launch(t, P, "L3_aligned", p + "L3_dcnpack", f3, om3, groups, "L")
    # L2
    o2 = _conv(t, P, "L2_offset_conv1", p + "L2_offset_conv1", f2, ref_of(f2), act="L")
    u3 = _up(t, "L3_offset_up", o3, 2, 2.0)
    o2 = _conv(t, P, "L2_offset_conv2", p + "L2_offset_conv2", o2, u3, act="L")
    o2 = _conv(t, P, "L2_offset", p + "L2_offset_conv3", o2, act="L")
    om2 = _conv(t, P, "L2_om", p + "L2_dcnpack.conv_offset_mask", o2)
    d2 = _dcn_launch(t, P, "L2_dcn", p + "L2_dcnpack", f2, om2, groups, "N")
    uf3 = _up(t, "L3_aligned_up", fe3)
    fe2 = _conv(t, P, "L2_aligned", p + "L2_fea_conv", d2, uf3, act="L")
    # L1
    o1 = _conv(t, P, "L1_offset_conv1", p + "L1_offset_conv1", f1, ref_of(f1), act="L")
    u2 = _up(t, "L2_offset_up", o2, 2, 2.0)
    o1 = _conv(t, P, "L1_offset_conv2", p + "L1_offset_conv2", o1, u2, act="L")
    o1 = _conv(t, P, "L1_offset", p + "L1_offset_conv3", o1, act="L")
    om1 = _conv(t, P, "L1_om", p + "L1_dcnpack.conv_offset_mask", o1)
    d1 = _dcn_launch(t, P, "L1_dcn", p + "L1_dcnpack", f1, om1, groups, "N")
    uf2 = _up(t, "L2_aligned_up", fe2)
    fe1 = _conv(t, P, "L1_aligned", p + "L1_fea_conv", d1, uf2)
    # cascade
    oc = _conv(t, P, "cas_offset_conv1", p + "cas_offset_conv1", fe1, ref_of(f1), act="L")
    oc = _conv(t, P, "cas_offset", p + "cas_offset_conv2", oc, act="L")
    omc = _conv(t, P, "cas_om", p + "cas_dcnpack.conv_offset_mask", oc)
    aligned = _dcn_launch(t, P, "aligned", p + "cas_dcnpack", fe1, omc, groups, "L")   # [B*N, C, H, W]
    # TSA
    q = "tsa_fusion."
    al5 = aligned.view(b, n, C, h, w)
    emb_ref = _conv(t, P, "tsa_emb_ref", q + "tAtt_2", al5[:, ctr])
    emb = _conv(t, P, "tsa_emb", q + "tAtt_1", aligned)
    cor = torch.sigmoid((emb.view(b, n, C, h, w) * emb_ref.unsqueeze(1)).sum(2))   # [B, N, H, W]
    gated = (al5 * cor.unsqueeze(2)).reshape(b, n * C, h, w)
    cor, gated = t.out("tsa_gate", cor, gated)
    fea = _conv(t, P, "tsa_fea", q + "fea_fusion", gated, act="L")
    att = _conv(t, P, "tsa_att1", q + "sAtt_1", gated, act="L")
    pmx, pav = t.out("tsa_pool1", F.max_pool2d(att, 3, 2, 1), F.avg_pool2d(att, 3, 2, 1))
    att = _conv(t, P, "tsa_att2", q + "sAtt_2", pmx, pav, act="L")
    attL = _conv(t, P, "tsa_attL1", q + "sAtt_L1", att, act="L")
    qmx, qav = t.out("tsa_pool2", F.max_pool2d(attL, 3, 2, 1), F.avg_pool2d(attL, 3, 2, 1))
    attL = _conv(t, P, "tsa_attL2", q + "sAtt_L2", qmx, qav, act="L")
    attL = _conv(t, P, "tsa_attL3", q + "sAtt_L3", attL, act="L")
    attLu = _up(t, "tsa_attL_up", attL)
    a3 = _conv(t, P, "tsa_att3", q + "sAtt_3", att, act="L")
    att3 = t.out("tsa_att3_add", a3 + attLu)
    a4 = _conv(t, P, "tsa_att4", q + "sAtt_4", att3, act="L")
    a4u = _up(t, "tsa_att4_up", a4)
    a5 = _conv(t, P, "tsa_att", q + "sAtt_5", a4u)
    ad = _conv(t, P, "tsa_add1", q + "sAtt_add_1", a5, act="L")
    ad = _conv(t, P, "tsa_add2", q + "sAtt_add_2", ad)
    out = t.out("tsa_blend", fea * torch.sigmoid(a5) * 2 + ad)
    # reconstruction
    for i in range(back_RBs):
        a = _conv(t, P, "rc_rb_a", "recon_trunk.%d.conv1" % i, out, act="R")
        out = _conv(t, P, "rc_rb_b", "recon_trunk.%d.conv2" % i, a, res=out)
    if scale == 4:
        out = _conv(t, P, "upconv1", "upconv1", out, act="L", ps=2)
    out = _conv(t, P, "upconv2", "upconv2", out, act="L", ps=2)
    out = _conv(t, P, "HRconv", "HRconv", out, act="L")
    xc = x[:, ctr]
    base = torch.cat([_up(t, "base_up", xc[i:i + 1], scale) for i in range(b)], 0)
    y = _conv(t, P, "conv_last", "conv_last", out, y_add=base)
    return y, t.names
