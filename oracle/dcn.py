"""ctypes binding + autograd wrapper for the C oracle of modulated deformable conv.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Mirrors the call contract of the reference's
``ModulatedDeformConvFunction`` (codes/models/archs/dcn/deform_conv.py:97-154): positional
signature ``(input, offset, mask, weight, bias, stride, padding, dilation, groups,
deformable_groups)``, first-order only (``once_differentiable`` there, :123).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_dcn.so")
_lib = None


def build(force=False):
    """Compile oracle/dcn_oracle.c with gcc (Makefile next to this file)."""
    src = os.path.join(_HERE, "dcn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for sfx in ("f32", "f64"):
            getattr(_lib, "dcn_oracle_forward_" + sfx).restype = ctypes.c_int
            getattr(_lib, "dcn_oracle_backward_" + sfx).restype = ctypes.c_int
    return _lib


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _sfx(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError("oracle DCN supports float32/float64, got %s" % t.dtype)


def _geom(x, w, stride, pad, dil, groups, dg):
    n, c, h, wd = x.shape
    return [ctypes.c_int(v) for v in (n, c, h, wd, w.shape[0], w.shape[2], w.shape[3], stride, pad,
                                      dil, groups, dg)]


def forward(x, offset, mask, w, b, stride, pad, dil, groups, dg):
    x, offset, mask, w = (t.contiguous() for t in (x, offset, mask, w))
    b = None if b is None else b.contiguous()
    n, _, h, wd = x.shape
    kh, kw = w.shape[2:]
    ho = (h + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    wo = (wd + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    out = x.new_empty((n, w.shape[0], ho, wo))
    rc = getattr(lib(), "dcn_oracle_forward_" + _sfx(x))(
        _ptr(x), _ptr(offset), _ptr(mask), _ptr(w), _ptr(b), _ptr(out),
        *_geom(x, w, stride, pad, dil, groups, dg))
    if rc:
        raise RuntimeError("dcn_oracle_forward failed rc=%d" % rc)
    return out


def backward(x, offset, mask, w, has_bias, gout, stride, pad, dil, groups, dg):
    x, offset, mask, w, gout = (t.contiguous() for t in (x, offset, mask, w, gout))
    gx, goff, gmask, gw = (torch.zeros_like(t) for t in (x, offset, mask, w))
    gb = x.new_zeros(w.shape[0]) if has_bias else None
    rc = getattr(lib(), "dcn_oracle_backward_" + _sfx(x))(
        _ptr(x), _ptr(offset), _ptr(mask), _ptr(w), _ptr(gout), _ptr(gx), _ptr(goff), _ptr(gmask),
        _ptr(gw), _ptr(gb), *_geom(x, w, stride, pad, dil, groups, dg))
    if rc:
        raise RuntimeError("dcn_oracle_backward failed rc=%d" % rc)
    return gx, goff, gmask, gw, gb


class _OracleDCN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, offset, mask, w, b, stride, pad, dil, groups, dg):
        ctx.cfg = (stride, pad, dil, groups, dg)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, offset, mask, w)
        return forward(x, offset, mask, w, b, stride, pad, dil, groups, dg)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, offset, mask, w = ctx.saved_tensors
        gx, goff, gmask, gw, gb = backward(x, offset, mask, w, ctx.has_bias, gout, *ctx.cfg)
        return gx, goff, gmask, gw, gb, None, None, None, None, None


def modulated_deform_conv(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                          groups=1, deformable_groups=1):
    return _OracleDCN.apply(x, offset, mask, weight, bias, stride, padding, dilation, groups,
                            deformable_groups)


def gather_reference(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                     groups=1, deformable_groups=1):
    """Independent pure-torch formulation (index gathers + autograd) used to cross-check the C
    oracle: same sampling rule (joint (-1,H)x(-1,W) gate, per-corner zero padding), no shared code.
    """
    n, c, h, w = x.shape
    cout, cg, kh, kw = weight.shape
    k = kh * kw
    ho = (h + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    dg = deformable_groups
    cpg = c // dg
    dev, dt = x.device, x.dtype
    ys = (torch.arange(ho, device=dev) * stride - padding).view(1, 1, 1, ho, 1).to(dt)
    xs = (torch.arange(wo, device=dev) * stride - padding).view(1, 1, 1, 1, wo).to(dt)
    ki = (torch.arange(k, device=dev) // kw * dilation).view(1, 1, k, 1, 1).to(dt)
    kj = (torch.arange(k, device=dev) % kw * dilation).view(1, 1, k, 1, 1).to(dt)
    off = offset.view(n, dg, k, 2, ho, wo)
    py = ys + ki + off[:, :, :, 0]          # n, dg, k, ho, wo
    px = xs + kj + off[:, :, :, 1]
    inside = (py > -1) & (px > -1) & (py < h) & (px < w)
    y0 = torch.floor(py)
    x0 = torch.floor(px)
    ly, lx = py - y0, px - x0
    y0, x0 = y0.long(), x0.long()
    xg = x.view(n, dg, cpg, h * w)

    def corner(yy, xx, wt):
        ok = inside & (yy >= 0) & (yy <= h - 1) & (xx >= 0) & (xx <= w - 1)
        idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).view(n, dg, 1, -1).expand(-1, -1, cpg, -1)
        v = torch.gather(xg, 3, idx).view(n, dg, cpg, k, ho, wo)
        return v * (wt * ok.to(dt)).unsqueeze(2)

    samp = (corner(y0, x0, (1 - ly) * (1 - lx)) + corner(y0, x0 + 1, (1 - ly) * lx)
            + corner(y0 + 1, x0, ly * (1 - lx)) + corner(y0 + 1, x0 + 1, ly * lx))
    col = samp * mask.view(n, dg, 1, k, ho, wo)             # n, dg, cpg, k, ho, wo
    col = col.reshape(n, groups, (c // groups) * k, ho * wo)
    wmat = weight.view(groups, cout // groups, cg * k)
    out = torch.einsum('gor,ngrp->ngop', wmat, col).reshape(n, cout, ho, wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
