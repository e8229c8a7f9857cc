"""Thin functional wrappers over the op-level C ABI (forward ops; no autograd here).

Used by the op-level drop-in (models/archs/dcn) and by the parity tests.  Arguments mirror the
reference call sites; tensors must be fp32 contiguous CUDA(HIP) tensors.
"""
import torch

from . import _lib as L


def mdcn_forward(x, offset, mask, weight, bias, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, act=L.ACT_NONE):
    n, c, h, w = x.shape
    cout, _, kh, kw = weight.shape
    ho = (h + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (w + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    out = x.new_empty((n, cout, ho, wo))
    L.check(L.lib().dvsr_mdcn_forward(L.ptr(x), L.ptr(offset), L.ptr(mask), L.ptr(weight),
                                      L.ptr(bias), L.ptr(out), n, c, h, w, cout, kh, kw, stride,
                                      padding, dilation, groups, deformable_groups, act, L.stream()),
            "dvsr_mdcn_forward")
    return out


def mdcn_forward_fast(x, offset, mask, weight, bias, deformable_groups, act=L.ACT_NONE):
    """LDS-sampler kernel (3x3, stride=pad=dil=1, C/dg=8)."""
    n, c, h, w = x.shape
    cout = weight.shape[0]
    out = x.new_empty((n, cout, h, w))
    nbytes = int(L.lib().dvsr_mdcn_forward_fast_workspace_bytes(c, cout, deformable_groups))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    L.check(L.lib().dvsr_mdcn_forward_fast(L.ptr(x), L.ptr(offset), L.ptr(mask), L.ptr(weight), L.ptr(bias),
                                           L.ptr(out), n, c, h, w, cout, deformable_groups, act,
                                           ws.data_ptr(), nbytes, L.stream()), "dvsr_mdcn_forward_fast")
    return out


def mdcn_pack_forward(x, om, weight, bias, deformable_groups, act=L.ACT_NONE):
    n, c, h, w = x.shape
    cout = weight.shape[0]
    out = x.new_empty((n, cout, h, w))
    L.check(L.lib().dvsr_mdcn_pack_forward(L.ptr(x), L.ptr(om), L.ptr(weight), L.ptr(bias),
                                           L.ptr(out), n, c, h, w, cout, 3, 3, 1, 1, 1, 1,
                                           deformable_groups, act, L.stream()),
            "dvsr_mdcn_pack_forward")
    return out


def conv2d_forward(x0, weight, bias=None, stride=1, act=L.ACT_NONE, x1=None, res=None,
                   pixel_shuffle=0, x1_bdiv=1):
    n, c0, h, w = x0.shape
    cout, ctot, ks, _ = weight.shape
    c1 = 0 if x1 is None else x1.shape[1]
    assert ctot == c0 + c1
    pad = ks // 2
    ho = (h + 2 * pad - ks) // stride + 1
    wo = (w + 2 * pad - ks) // stride + 1
    if pixel_shuffle:
        y = x0.new_empty((n, cout // 4, 2 * ho, 2 * wo))
    else:
        y = x0.new_empty((n, cout, ho, wo))
    d = L.Conv2dDesc(L.ptr(x0), L.ptr(x1), L.ptr(weight), L.ptr(bias), L.ptr(res), L.ptr(y), n, c0,
                     c1, h, w, cout, ks, stride, pad, act, pixel_shuffle, x1_bdiv, 0, 0)
    L.check(L.lib().dvsr_conv2d_forward(d, L.stream()), "dvsr_conv2d_forward")
    return y


def upsample_bilinear(x, scale, mul=1.0):
    n, c, h, w = x.shape
    y = x.new_empty((n, c, h * scale, w * scale))
    L.check(L.lib().dvsr_upsample_bilinear_forward(L.ptr(x), L.ptr(y), n * c, h, w, scale, mul,
                                                   L.stream()), "dvsr_upsample_bilinear_forward")
    return y


def upsample_bilinear_backward(gy, scale, mul=1.0):
    n, c, ho, wo = gy.shape
    h, w = ho // scale, wo // scale
    gx = gy.new_empty((n, c, h, w))
    L.check(L.lib().dvsr_upsample_bilinear_backward(L.ptr(gy), L.ptr(gx), n * c, h, w, scale, mul, 0,
                                                    L.stream()), "dvsr_upsample_bilinear_backward")
    return gx


def pool3s2(x):
    n, c, h, w = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ymax, yavg = x.new_empty((n, c, ho, wo)), x.new_empty((n, c, ho, wo))
    L.check(L.lib().dvsr_pool3s2_forward(L.ptr(x), L.ptr(ymax), L.ptr(yavg), n * c, h, w, L.stream()),
            "dvsr_pool3s2_forward")
    return ymax, yavg


def pool3s2_backward(x, gmax, gavg):
    n, c, h, w = x.shape
    gx = torch.empty_like(x)
    L.check(L.lib().dvsr_pool3s2_backward(L.ptr(x), L.ptr(gmax), L.ptr(gavg), L.ptr(gx), n * c, h, w,
                                          L.stream()), "dvsr_pool3s2_backward")
    return gx


def tsa_gate(emb, emb_ref, aligned):
    b, n, c, h, w = aligned.shape
    cor = aligned.new_empty((b, n, h, w))
    gated = aligned.new_empty((b, n * c, h, w))
    L.check(L.lib().dvsr_tsa_gate_forward(L.ptr(emb), L.ptr(emb_ref), L.ptr(aligned), L.ptr(cor),
                                          L.ptr(gated), b, n, c, h * w, L.stream()),
            "dvsr_tsa_gate_forward")
    return cor, gated


def tsa_gate_backward(emb, emb_ref, aligned, cor, g_gated):
    b, n, c, h, w = aligned.shape
    g_emb, g_ref, g_al = torch.empty_like(emb), torch.empty_like(emb_ref), torch.empty_like(aligned)
    L.check(L.lib().dvsr_tsa_gate_backward(L.ptr(emb), L.ptr(emb_ref), L.ptr(aligned), L.ptr(cor),
                                           L.ptr(g_gated), L.ptr(g_emb), L.ptr(g_ref), L.ptr(g_al),
                                           b, n, c, h * w, L.stream()), "dvsr_tsa_gate_backward")
    return g_emb, g_ref, g_al


def tsa_blend(fea, att, att_add):
    out = torch.empty_like(fea)
    L.check(L.lib().dvsr_tsa_blend_forward(L.ptr(fea), L.ptr(att), L.ptr(att_add), L.ptr(out),
                                           fea.numel(), L.stream()), "dvsr_tsa_blend_forward")
    return out


def tsa_blend_backward(fea, att, g, g_att_io):
    g_fea = torch.empty_like(fea)
    L.check(L.lib().dvsr_tsa_blend_backward(L.ptr(fea), L.ptr(att), L.ptr(g), L.ptr(g_fea),
                                            L.ptr(g_att_io), fea.numel(), L.stream()),
            "dvsr_tsa_blend_backward")
    return g_fea


def conv2d_backward(gy, x0, weight, stride=1, x1=None, need_gx=True):
    """gy = gradient w.r.t. the pre-activation output.  Returns (gx0, gx1, gw, gb)."""
    n, c0, h, w = x0.shape
    cout, ctot, ks, _ = weight.shape
    c1 = 0 if x1 is None else x1.shape[1]
    d = L.Conv2dDesc(L.ptr(x0), L.ptr(x1), L.ptr(weight), None, None, None, n, c0, c1, h, w, cout, ks, stride,
                     ks // 2, 0, 0, 1, 0, 0)
    ws = torch.empty(max(int(L.lib().dvsr_conv2d_backward_workspace_bytes(d)), 16), dtype=torch.uint8,
                     device=x0.device)
    gx0 = torch.empty_like(x0) if need_gx else None
    gx1 = torch.empty_like(x1) if (need_gx and x1 is not None) else None
    gw, gb = torch.empty_like(weight), weight.new_empty(cout)
    L.check(L.lib().dvsr_conv2d_backward(d, L.ptr(gy), L.ptr(gx0), L.ptr(gx1), L.ptr(gw), L.ptr(gb),
                                         ws.data_ptr(), ws.numel(), L.stream()), "dvsr_conv2d_backward")
    return gx0, gx1, gw, gb


def mdcn_backward(x, offset, mask, weight, gout, stride=1, padding=0, dilation=1, groups=1,
                  deformable_groups=1, with_bias=True):
    """Returns (gx, goffset, gmask, gw, gb) like the reference op's backward (deform_conv.py:122-142)."""
    n, c, h, w = x.shape
    cout, _, kh, kw = weight.shape
    nbytes = int(L.lib().dvsr_mdcn_backward_workspace_bytes(n, c, h, w, cout, kh, kw, stride, padding, dilation))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    gx = torch.zeros_like(x)
    goff, gmask, gw = torch.empty_like(offset), torch.empty_like(mask), torch.empty_like(weight)
    gb = weight.new_empty(cout) if with_bias else None
    L.check(L.lib().dvsr_mdcn_backward(L.ptr(x), L.ptr(offset), L.ptr(mask), L.ptr(weight), L.ptr(gout),
                                       L.ptr(gx), L.ptr(goff), L.ptr(gmask), L.ptr(gw), L.ptr(gb), n, c, h, w,
                                       cout, kh, kw, stride, padding, dilation, groups, deformable_groups,
                                       ws.data_ptr(), nbytes, L.stream()), "dvsr_mdcn_backward")
    return gx, goff, gmask, gw, gb


class _Charbonnier(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, eps):
        if x.shape != y.shape:      # the kernel walks both with one element count; torch would broadcast or raise
            raise RuntimeError("charbonnier: prediction %s and target %s differ in shape" % (tuple(x.shape), tuple(y.shape)))
        x, y = x.contiguous(), y.contiguous()
        ws = torch.empty(int(L.lib().dvsr_charbonnier_workspace_bytes()), dtype=torch.uint8, device=x.device)
        loss = x.new_empty(())
        L.check(L.lib().dvsr_charbonnier_forward(L.ptr(x), L.ptr(y), loss.data_ptr(), x.numel(), eps,
                                                 ws.data_ptr(), ws.numel(), L.stream()),
                "dvsr_charbonnier_forward")
        ctx.save_for_backward(x, y)
        ctx.eps = eps
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = torch.empty_like(x)
        g = g.contiguous().float()
        L.check(L.lib().dvsr_charbonnier_backward(L.ptr(x), L.ptr(y), g.data_ptr(), L.ptr(gx), x.numel(),
                                                  ctx.eps, L.stream()), "dvsr_charbonnier_backward")
        return (gx if ctx.needs_input_grad[0] else None, -gx if ctx.needs_input_grad[1] else None, None)


def charbonnier(x, y, eps=1e-6):
    """mean(sqrt((x-y)^2 + eps)) with autograd (models/loss.py:26-30)."""
    return _Charbonnier.apply(x, y, eps)


class _CharbonnierPerSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, eps):
        if x.shape != y.shape:
            raise RuntimeError("charbonnier: prediction %s and target %s differ in shape" % (tuple(x.shape), tuple(y.shape)))
        x, y = x.contiguous(), y.contiguous()
        k = x.shape[0]
        n = x.numel() // k
        ws = torch.empty(k * int(L.lib().dvsr_charbonnier_workspace_bytes()), dtype=torch.uint8, device=x.device)
        loss = x.new_empty((k,))
        L.check(L.lib().dvsr_charbonnier_forward_grouped(L.ptr(x), L.ptr(y), loss.data_ptr(), n, k, eps, ws.data_ptr(),
                                                         ws.numel(), L.stream()), "dvsr_charbonnier_forward_grouped")
        ctx.save_for_backward(x, y)
        ctx.eps = eps
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = torch.empty_like(x)
        g = g.contiguous().float()
        k = x.shape[0]
        L.check(L.lib().dvsr_charbonnier_backward_grouped(L.ptr(x), L.ptr(y), g.data_ptr(), L.ptr(gx), x.numel() // k, k,
                                                          ctx.eps, L.stream()), "dvsr_charbonnier_backward_grouped")
        return (gx if ctx.needs_input_grad[0] else None, -gx if ctx.needs_input_grad[1] else None, None)


def charbonnier_per_sample(x, y, eps=1e-6):
    """[K, ...] x 2 -> [K]: loss[k] = mean(sqrt((x[k]-y[k])^2 + eps)), each reduced exactly as `charbonnier` reduces a
    single sample (bit-identical values) -- the pixel losses of K frames adapted as one batch."""
    return _CharbonnierPerSample.apply(x, y, eps)


class _InnerLossPerSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, x, y, weight):
        if x.shape != y.shape or base.shape != (x.shape[0],):
            raise RuntimeError("inner_loss: %s vs %s, base %s" % (tuple(x.shape), tuple(y.shape), tuple(base.shape)))
        x, y = x.contiguous(), y.contiguous()
        k = x.shape[0]
        b = base.detach().contiguous().float()
        ws = torch.empty(k * int(L.lib().dvsr_charbonnier_workspace_bytes()), dtype=torch.uint8, device=x.device)
        loss = x.new_empty((k,))
        L.check(L.lib().dvsr_l1_tail_forward_grouped(L.ptr(x), L.ptr(y), b.data_ptr(), float(weight), loss.data_ptr(),
                                                     x.numel() // k, k, ws.data_ptr(), ws.numel(), L.stream()),
                "dvsr_l1_tail_forward_grouped")
        ctx.save_for_backward(x, y)
        ctx.weight = float(weight)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.contiguous().float()
        gx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gx = torch.empty_like(x)
            k = x.shape[0]
            L.check(L.lib().dvsr_l1_tail_backward_grouped(L.ptr(x), L.ptr(y), g.data_ptr(), ctx.weight, L.ptr(gx),
                                                          x.numel() // k, k, L.stream()), "dvsr_l1_tail_backward_grouped")
        return (g if ctx.needs_input_grad[0] else None, gx if ctx.needs_input_grad[1] else None,
                -gx if ctx.needs_input_grad[2] else None, None)


def inner_loss_per_sample(loss_pix, slr, slr_fixed, weight=10.0):
    """[K] + weight * per-sample F.l1_loss(slr[k], slr_fixed[k]) -> [K] (see inner_loss)."""
    return _InnerLossPerSample.apply(loss_pix, slr, slr_fixed, weight)


class _InnerLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, x, y, weight):
        if x.shape != y.shape:
            raise RuntimeError("inner_loss: %s vs %s" % (tuple(x.shape), tuple(y.shape)))
        x, y = x.contiguous(), y.contiguous()
        b = base.detach().reshape(()).float()
        ws = torch.empty(int(L.lib().dvsr_charbonnier_workspace_bytes()), dtype=torch.uint8, device=x.device)
        loss = x.new_empty(())
        L.check(L.lib().dvsr_l1_tail_forward(L.ptr(x), L.ptr(y), b.data_ptr(), float(weight), loss.data_ptr(), x.numel(),
                                             ws.data_ptr(), ws.numel(), L.stream()), "dvsr_l1_tail_forward")
        ctx.save_for_backward(x, y)
        ctx.weight = float(weight)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.contiguous().float()
        gx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gx = torch.empty_like(x)
            L.check(L.lib().dvsr_l1_tail_backward(L.ptr(x), L.ptr(y), g.data_ptr(), ctx.weight, L.ptr(gx), x.numel(),
                                                  L.stream()), "dvsr_l1_tail_backward")
        return (g if ctx.needs_input_grad[0] else None, gx if ctx.needs_input_grad[1] else None,
                -gx if ctx.needs_input_grad[2] else None, None)


def inner_loss(loss_pix, slr, slr_fixed, weight=10.0):
    """loss_pix + weight * F.l1_loss(slr, slr_fixed) (test_dynavsr.py:264-274) as one native reduction: the pixel loss
    stays a device scalar, its gradient passes through, the L1 gradient is one kernel."""
    return _InnerLoss.apply(loss_pix, slr, slr_fixed, weight)


class _PatchGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, py, px, edge, scale):
        import ctypes
        src = src.contiguous().float()
        lead, (h, w) = src.shape[:-2], src.shape[-2:]
        planes = 1
        for d in lead:
            planes *= d
        n = len(py)
        e = edge * scale
        dst = src.new_empty((n,) + tuple(lead) + (e, e))
        apy, apx = (ctypes.c_int * n)(*py), (ctypes.c_int * n)(*px)
        L.check(L.lib().dvsr_patch_gather_forward(L.ptr(src), L.ptr(dst), apy, apx, n, planes, h, w, edge, scale, L.stream()),
                "dvsr_patch_gather_forward")
        ctx.geo = (tuple(src.shape), apy, apx, n, planes, h, w, edge, scale)
        return dst

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        shape, apy, apx, n, planes, h, w, edge, scale = ctx.geo
        g = g.contiguous().float()
        gs = g.new_empty(shape)
        L.check(L.lib().dvsr_patch_gather_backward(L.ptr(g), L.ptr(gs), apy, apx, n, planes, h, w, edge, scale, L.stream()),
                "dvsr_patch_gather_backward")
        return gs, None, None, None, None


def patch_gather(src, py, px, edge, scale=1):
    """src [..., H, W] -> [P, ..., scale*edge, scale*edge]: patch p starts at (scale*py[p], scale*px[p])."""
    if not src.is_cuda:
        raise RuntimeError("patch_gather runs on the GPU (libdynavsr_hip); there is no CPU fallback")
    return _PatchGather.apply(src, [int(v) for v in py], [int(v) for v in px], int(edge), int(scale))
