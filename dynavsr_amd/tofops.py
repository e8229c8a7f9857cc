"""Autograd faces of the op-level C ABI used by the TOFlow backbone.

The reference builds TOFlow from torch.nn modules (codes/models/archs/TOF_arch.py:25-140) and arch_util.flow_warp
(:55-79); here every op of that graph is a native kernel behind a ``torch.autograd.Function``:
convolutions (7x7 / 9x9 / 1x1 with bias, ReLU and residual fused) on the MFMA conv kernels, BatchNorm2d(+ReLU),
flow_warp, the 2x2 average pool of the SpyNet pyramids, the align_corners=True flow up-sampling and the
(de)normalisation.  fp32 CUDA(HIP) tensors only; there is no CPU path.
"""
import ctypes

import torch

from . import _lib as L


def _chk(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dynavsr_amd TOFlow ops run on the MI355X only (tensor on %s); there is no CPU fallback" % t.device)


def _f(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _packed_ok(d, ks):
    """1x1 / 3x3 always have a packed kernel; 7x7 / 9x9 only the row-split DMA kernel (W % 4 == 0, aligned, Cout >= 16)."""
    if ks in (1, 3):
        return True
    if ks not in (7, 9):
        return False
    geo = (ctypes.c_int * 4)()
    L.check(L.lib().dvsr_conv2d_packed_geometry(d, ctypes.byref(geo)), "dvsr_conv2d_packed_geometry")
    return geo[3] == 2


class _Conv(torch.autograd.Function):
    """y = act(conv(x, w) + b) [+ res]; stride 1, pad ks // 2."""

    @staticmethod
    def forward(ctx, x, w, b, res, act):
        _chk(x, w)
        x, w, b = _f(x), _f(w), _f(b)
        res = _f(res) if res is not None else None
        n, c, h, wd = x.shape
        cout, ctot, ks, _ = w.shape
        if ctot != c:
            raise RuntimeError("conv: weight expects %d input channels, got %d" % (ctot, c))
        y = x.new_empty((n, cout, h, wd))
        d = L.Conv2dDesc(L.ptr(x), None, L.ptr(w), L.ptr(b), L.ptr(res), L.ptr(y), n, c, 0, h, wd, cout, ks, 1, ks // 2,
                         act, 0, 1, 0, 0)
        if _packed_ok(d, ks):   # pipelined / small-grid / DMA kernels over a packed weight image
            ws = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)), 16), dtype=torch.uint8, device=x.device)
            L.check(L.lib().dvsr_conv2d_forward_packed(d, ws.data_ptr(), ws.numel(), L.stream()), "dvsr_conv2d_forward_packed")
        else:
            L.check(L.lib().dvsr_conv2d_forward(d, L.stream()), "dvsr_conv2d_forward")
        # the activation mask is the sign of the activated value, i.e. of y BEFORE the residual was added
        ctx.save_for_backward(x, w, None if act == L.ACT_NONE else (y if res is None else y - res))
        ctx.act, ctx.has_res = act, res is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = _f(gy)
        g_res = gy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        if ctx.act == L.ACT_RELU:
            gy = gy * (y > 0)           # (the residual is added AFTER the activation: its gradient is the raw gy)
        elif ctx.act == L.ACT_LRELU:
            gy = torch.where(y > 0, gy, 0.1 * gy)
        n, c, h, wd = x.shape
        cout, _, ks, _ = w.shape
        d = L.Conv2dDesc(L.ptr(x), None, L.ptr(w), None, None, None, n, c, 0, h, wd, cout, ks, 1, ks // 2, 0, 0, 1, 0, 0)
        ws = torch.empty(max(int(L.lib().dvsr_conv2d_backward_workspace_bytes(d)), 16), dtype=torch.uint8, device=x.device)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw, gb = torch.empty_like(w), w.new_empty(cout)
        packed = False
        if gx is not None:   # the data gradient is the forward kernel over gy with Cout and Cin swapped
            dg = L.Conv2dDesc(L.ptr(gy), None, None, None, None, None, n, cout, 0, h, wd, c, ks, 1, ks // 2, 0, 0, 1, 0, 0)
            packed = _packed_ok(dg, ks)
        L.check(L.lib().dvsr_conv2d_backward(d, L.ptr(gy), None if packed else L.ptr(gx), None, L.ptr(gw), L.ptr(gb),
                                             ws.data_ptr(), ws.numel(), L.stream()), "dvsr_conv2d_backward")
        if packed:
            pw = torch.empty(max(int(L.lib().dvsr_conv2d_packed_workspace_bytes(d)), 16), dtype=torch.uint8, device=x.device)
            L.check(L.lib().dvsr_conv2d_dgrad_packed(d, L.ptr(gy), L.ptr(gx), pw.data_ptr(), pw.numel(), L.stream()),
                    "dvsr_conv2d_dgrad_packed")
        return gx, gw, gb, g_res, None


def conv(x, weight, bias, res=None, act=L.ACT_NONE):
    return _Conv.apply(x, weight, bias, res, act)


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        _chk(x, gamma)
        x = _f(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        mean, rstd = x.new_empty(c), x.new_empty(c)
        ws = torch.empty(int(L.lib().dvsr_batchnorm_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
        L.check(L.lib().dvsr_batchnorm_forward(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var),
                                               L.ptr(y), L.ptr(mean), L.ptr(rstd), n, c, h * w, int(training),
                                               float(momentum), float(eps), int(relu), ws.data_ptr(), ws.numel(), L.stream()),
                "dvsr_batchnorm_forward")
        ctx.save_for_backward(x, y if relu else None, gamma, mean, rstd)
        ctx.training, ctx.relu = bool(training), bool(relu)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        gy = _f(gy)
        n, c, h, w = x.shape
        gx, gg, gb = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(int(L.lib().dvsr_batchnorm_workspace_bytes(c)), dtype=torch.uint8, device=x.device)
        L.check(L.lib().dvsr_batchnorm_backward(L.ptr(x), L.ptr(gy), L.ptr(y), L.ptr(gamma), L.ptr(mean), L.ptr(rstd),
                                                L.ptr(gx), L.ptr(gg), L.ptr(gb), n, c, h * w, int(ctx.training),
                                                int(ctx.relu), ws.data_ptr(), ws.numel(), L.stream()),
                "dvsr_batchnorm_backward")
        return gx, gg, gb, None, None, None, None, None, None


def batchnorm(x, bn, relu=True):
    """nn.BatchNorm2d module `bn` (its parameters / running estimates) applied natively, ReLU fused."""
    if bn.training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    use_batch = bn.training or not bn.track_running_stats
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _BatchNorm.apply(x, bn.weight, bn.bias, bn.running_mean.detach() if bn.track_running_stats else None,
                            bn.running_var.detach() if bn.track_running_stats else None, use_batch, momentum, bn.eps, relu)


class _FlowWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow):
        _chk(x, flow)
        x, flow = _f(x), _f(flow)
        n, c, h, w = x.shape
        if tuple(flow.shape) != (n, 2, h, w):
            raise RuntimeError("flow_warp: flow must be [N,2,H,W] matching x %s, got %s" % (tuple(x.shape), tuple(flow.shape)))
        out = torch.empty_like(x)
        L.check(L.lib().dvsr_flow_warp_forward(L.ptr(x), L.ptr(flow), L.ptr(out), n, c, h, w, 0, L.stream()),
                "dvsr_flow_warp_forward")
        ctx.save_for_backward(x, flow)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, flow = ctx.saved_tensors
        g = _f(g)
        n, c, h, w = x.shape
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gf = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        if gx is None and gf is None:
            return None, None
        L.check(L.lib().dvsr_flow_warp_backward(L.ptr(x), L.ptr(flow), L.ptr(g), L.ptr(gx), L.ptr(gf), n, c, h, w, 0,
                                                L.stream()), "dvsr_flow_warp_backward")
        return gx, gf


def flow_warp(x, flow):
    """arch_util.flow_warp(x, flow.permute(0, 2, 3, 1)) with ``flow`` kept channel-first [N,2,H,W]."""
    return _FlowWarp.apply(x, flow)


class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = _f(x)
        n, c, h, w = x.shape
        y = x.new_empty((n, c, h // 2, w // 2))
        L.check(L.lib().dvsr_avgpool2_forward(L.ptr(x), L.ptr(y), n * c, h, w, L.stream()), "dvsr_avgpool2_forward")
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        g = _f(g)
        gx = g.new_empty(ctx.shape)
        L.check(L.lib().dvsr_avgpool2_backward(L.ptr(g), L.ptr(gx), n * c, h, w, 0, L.stream()), "dvsr_avgpool2_backward")
        return gx


def avg_pool2(x):
    return _AvgPool2.apply(x)


class _ResizeAC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ho, wo, mul):
        _chk(x)
        x = _f(x)
        n, c, h, w = x.shape
        y = x.new_empty((n, c, ho, wo))
        L.check(L.lib().dvsr_resize_bilinear_ac_forward(L.ptr(x), L.ptr(y), n * c, h, w, ho, wo, float(mul), 0, L.stream()),
                "dvsr_resize_bilinear_ac_forward")
        ctx.geo = (n, c, h, w, ho, wo, float(mul))
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        n, c, h, w, ho, wo, mul = ctx.geo
        g = _f(g)
        gx = g.new_empty((n, c, h, w))
        L.check(L.lib().dvsr_resize_bilinear_ac_backward(L.ptr(g), L.ptr(gx), n * c, h, w, ho, wo, mul, 0, L.stream()),
                "dvsr_resize_bilinear_ac_backward")
        return gx, None, None, None


def resize_bilinear_ac(x, size, mul=1.0):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=True) * mul."""
    return _ResizeAC.apply(x, int(size[0]), int(size[1]), mul)


class _BicubicAC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        _chk(x)
        x = _f(x)
        lead, (h, w) = x.shape[:-2], x.shape[-2:]
        planes = 1
        for d in lead:
            planes *= d
        y = x.new_empty(tuple(lead) + (h * scale, w * scale))
        L.check(L.lib().dvsr_upsample_bicubic_ac_forward(L.ptr(x), L.ptr(y), planes, h, w, scale, L.stream()),
                "dvsr_upsample_bicubic_ac_forward")
        ctx.geo = (tuple(x.shape), planes, h, w, scale)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        shape, planes, h, w, scale = ctx.geo
        g = _f(g)
        gx = g.new_empty(shape)
        L.check(L.lib().dvsr_upsample_bicubic_ac_backward(L.ptr(g), L.ptr(gx), planes, h, w, scale, L.stream()),
                "dvsr_upsample_bicubic_ac_backward")
        return gx, None


def upsample_bicubic_ac(x, scale):
    """F.interpolate(x, scale_factor=scale, mode='bicubic', align_corners=True) over the last two dims."""
    return _BicubicAC.apply(x, int(scale))


class _ChannelAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        _chk(x)
        x = _f(x)
        n, c = x.shape[0], x.shape[1]
        hw = x[0, 0].numel()
        y = torch.empty_like(x)
        L.check(L.lib().dvsr_channel_affine(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(y), n, c, hw, 0, 0, 0, L.stream()),
                "dvsr_channel_affine")
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        g = _f(g)
        gx = torch.empty_like(g)
        n, c = g.shape[0], g.shape[1]
        L.check(L.lib().dvsr_channel_affine(L.ptr(g), L.ptr(scale), None, L.ptr(gx), n, c, g[0, 0].numel(), 0, 0, 0,
                                            L.stream()), "dvsr_channel_affine")
        return gx, None, None


def channel_affine(x, scale, shift):
    """x * scale[c] + shift[c] over dim 1 (constants: no gradient to scale / shift)."""
    return _ChannelAffine.apply(x, scale, shift)


# ---- DUF (codes/models/archs/DUF_arch.py) ----------------------------------------------------------------
class _TemporalGather3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, t, pad_t):
        _chk(x)
        x = _f(x)
        bt, c, h, w = x.shape
        if bt != b * t:
            raise RuntimeError("temporal_gather3: %d frames, expected B*T = %d*%d" % (bt, b, t))
        to = t + 2 * pad_t - 2
        y = x.new_empty((b * to, 3 * c, h, w))
        L.check(L.lib().dvsr_temporal_gather3_forward(L.ptr(x), L.ptr(y), b, t, c, h * w, pad_t, L.stream()),
                "dvsr_temporal_gather3_forward")
        ctx.geo = (b, t, c, h, w, pad_t)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        b, t, c, h, w, pad_t = ctx.geo
        g = _f(g)
        gx = g.new_empty((b * t, c, h, w))
        L.check(L.lib().dvsr_temporal_gather3_backward(L.ptr(g), L.ptr(gx), b, t, c, h * w, pad_t, L.stream()),
                "dvsr_temporal_gather3_backward")
        return gx, None, None, None


def temporal_gather3(x, b, t, pad_t):
    """[B*T,C,H,W] -> [B*To,3C,H,W] (channel c*3+kt = frame t+kt-pad_t, zero outside): the input of a (3,3,3) Conv3d
    run as a 3x3 conv2d.  pad_t 1 keeps T, 0 gives To = T - 2."""
    return _TemporalGather3.apply(x, b, t, pad_t)


class _DynamicFilter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xc, logits, res, scale, adapt):
        _chk(xc, logits, res)
        xc, logits, res = _f(xc), _f(logits), _f(res)
        b, c, h, w = xc.shape
        r = scale * scale
        if c != 3 or tuple(logits.shape) != (b, 25 * r, h, w) or tuple(res.shape) != (b, 3 * r, h, w):
            raise RuntimeError("dynamic_filter: x_center %s, logits %s, residual %s do not fit scale %d"
                               % (tuple(xc.shape), tuple(logits.shape), tuple(res.shape), scale))
        out = xc.new_empty((b, 3, scale * h, scale * w))
        L.check(L.lib().dvsr_dynamic_filter_forward(L.ptr(xc), L.ptr(logits), L.ptr(res), L.ptr(out), b, h, w, scale,
                                                    int(adapt), L.stream()), "dvsr_dynamic_filter_forward")
        ctx.save_for_backward(xc, logits)
        ctx.cfg = (scale, int(adapt))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        xc, logits = ctx.saved_tensors
        scale, adapt = ctx.cfg
        g = _f(g)
        b, _, h, w = xc.shape
        gl = torch.empty_like(logits)
        gr = g.new_empty((b, 3 * scale * scale, h, w))
        gx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        L.check(L.lib().dvsr_dynamic_filter_backward(L.ptr(xc), L.ptr(logits), L.ptr(g), L.ptr(gl), L.ptr(gr), L.ptr(gx), b, h,
                                                     w, scale, adapt, L.stream()), "dvsr_dynamic_filter_backward")
        return gx, gl, gr, None, None


def dynamic_filter(x_center, filter_logits, residual, scale, adapt_official, taps_given=False):
    """softmax over the 25 taps + DynamicUpsamplingFilter_3C + residual (adapt_official order) + pixel_shuffle.
    taps_given: `filter_logits` holds the filter taps themselves, applied as they are (no softmax)."""
    return _DynamicFilter.apply(x_center, filter_logits, residual, scale, int(bool(adapt_official)) | (2 if taps_given else 0))
