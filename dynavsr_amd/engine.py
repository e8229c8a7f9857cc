"""Python face of the native EDVR execution plan (csrc/engine.hip).

One `torch.autograd.Function` spans the whole backbone: forward enqueues every kernel of
EDVR.forward with ONE C call, backward (first-order, like the reference's once_differentiable
DCN, deform_conv.py:123) enqueues the whole backward and returns d/dx and d/d(144 params).
Parameters stay ordinary leaf nn.Parameters, so external torch.optim objects, deepcopy and
`param.grad += g` in the DynaVSR drivers keep working.
"""
import ctypes
import threading

import torch

from . import _lib as L

_plans = {}
# Environment switches the native plan builder reads EVERY TIME it builds a plan (conv2_choose, Builder::conv, build_backward,
# dvsr_*_plan_create): they are part of the plan-cache key, so a plan built under one setting is never handed out under another.
_GEOMETRY_ENV = ("DVSR_CONV_WINO", "DVSR_CONV_WINO3", "DVSR_CONV_WINO5", "DVSR_CONV_V1", "DVSR_EST_SPLIT", "DVSR_EST_SPLIT2", "DVSR_FUSE_ACT_BWD",
                 "DVSR_BWD_STREAMS")
# ... and the ones the native side reads ONCE PER PROCESS (function-local statics): changing them after the first plan has no
# effect, so they are deliberately NOT in the key -- set them before the first call (the A/B tools run one process per value).
_PROCESS_ENV = ("DVSR_CONV_WINO_T16", "DVSR_CONV_DMA", "DVSR_CONV_DMAROW", "DVSR_CONV_TILE", "DVSR_CONV_LDS", "DVSR_CONV_LAST_MFMA",
                "DVSR_CONV_KSPLIT_BELOW", "DVSR_CONV_KSPLIT_NT", "DVSR_CONV_CC16_BELOW", "DVSR_SPLIT_TH8_FROM", "DVSR_WGRAD_SPLIT3",
                "DVSR_WGRAD_SPLITS", "DVSR_WGRAD_SIMPLE", "DVSR_WGRAD_WIDE", "DVSR_WGRAD_S3V", "DVSR_WGRAD_S3_KYS_BELOW", "DVSR_WGRAD_S3_WGS",
                "DVSR_WGRAD_S3W", "DVSR_WGRAD_KYS_BELOW", "DVSR_WGRAD_KYS_WGS", "DVSR_WGRAD_BF_WGS", "DVSR_DCN_FWD", "DVSR_DCN_BWD",
                "DVSR_EST_FUSE_PAD", "DVSR_FUSE_RES_BWD", "DVSR_BWD_FORK_EVERY", "DVSR_BWD_PROBE", "DVSR_TSA_DUAL", "DVSR_TSA_V",
                "DVSR_DEGRADE_GENERIC", "DVSR_UP2")
_process_env_seen = None   # their values when the first plan of the process was built


def _check_process_env():
    """The once-per-process switches are snapshot when the first plan is built; a later plan built under another value gets a
    warning instead of silence (the native side keeps the value it read first)."""
    import os
    import warnings
    global _process_env_seen
    now = tuple(os.environ.get(k) for k in _PROCESS_ENV)
    if _process_env_seen is None:
        _process_env_seen = now
        return
    for k, a, b in zip(_PROCESS_ENV, _process_env_seen, now):
        if a != b:
            warnings.warn("dynavsr_amd: %s=%r now, but the native library read %r when the first plan of this process was built "
                          "and keeps that value (switches of this kind are read once per process: set them before the first "
                          "plan, one process per value)" % (k, b, a), RuntimeWarning, stacklevel=3)
    _process_env_seen = now


# dvsr_edvr_plan_work's nine doubles (include/dynavsr_hip.h): *_executed = fp32 products as the kernels shape them,
# *_f32_pipe / *_bf16_pipe = FLOPs issued to either matrix pipe
_WORK_KEYS = ("fwd_algorithmic", "fwd_executed", "bwd_algorithmic", "bwd_executed", "fwd_bytes",
              "fwd_f32_pipe", "fwd_bf16_pipe", "bwd_f32_pipe", "bwd_bf16_pipe")


def _env_key():
    import os
    return tuple(os.environ.get(k) for k in _GEOMETRY_ENV)


class Plan:
    """Owns one dvsr_edvr_plan (shape-specialised launch tape)."""

    def __init__(self, cfg, b, h, w, grad_groups=1, weight_sets=1):
        self.key = (cfg, b, h, w, grad_groups, weight_sets)
        cfg = tuple(cfg) + (0,) * (8 - len(cfg))  # older 7-tuples: fp32 MFMA
        self.cfg = dict(zip(("nf", "nframes", "groups", "front_RBs", "back_RBs", "scale", "center", "bf16_mfma"), cfg))
        self.b, self.h, self.w, self.grad_groups, self.weight_sets = b, h, w, grad_groups, weight_sets
        self._h = ctypes.c_void_p()
        L.check(L.lib().dvsr_edvr_plan_create_ex(L.EdvrConfig(*cfg), b, h, w, grad_groups, weight_sets,
                                                 ctypes.byref(self._h)), "dvsr_edvr_plan_create_ex")
        self.n_params = L.lib().dvsr_edvr_num_params(self._h)
        self.n_launches = L.lib().dvsr_edvr_num_launches(self._h)
        self.n_backward_launches = L.lib().dvsr_edvr_num_backward_launches(self._h)

    def workspace_bytes(self, need_grad):
        return int(L.lib().dvsr_edvr_workspace_bytes(self._h, int(need_grad)))

    def forward(self, params, x, out, ws, packed=False):
        """packed: `ws` still holds the weight packs of an earlier forward of this plan with these parameter values
        (dvsr_edvr_forward_packed: the packing launches are skipped; FrozenWeights keeps that contract)."""
        arr = (ctypes.c_void_p * len(params))(*[L.ptr(p) for p in params])
        fn = L.lib().dvsr_edvr_forward_packed if packed else L.lib().dvsr_edvr_forward
        L.check(fn(self._h, arr, L.ptr(x), L.ptr(out), ws.data_ptr(), ws.numel() * ws.element_size(), L.stream()),
                "dvsr_edvr_forward_packed" if packed else "dvsr_edvr_forward")

    def backward(self, params, x, gout, gparams, gx, ws):
        arr = (ctypes.c_void_p * len(params))(*[L.ptr(p) for p in params])
        garr = (ctypes.c_void_p * len(gparams))(*[L.ptr(g) for g in gparams])
        L.check(L.lib().dvsr_edvr_backward(self._h, arr, L.ptr(x), L.ptr(gout), garr, L.ptr(gx),
                                           ws.data_ptr(), ws.numel() * ws.element_size(), L.stream()),
                "dvsr_edvr_backward")

    def forward_timed(self, params, x, out, ws):
        """Per-launch milliseconds (hipEvents on the current stream, stream-synchronising)."""
        arr = (ctypes.c_void_p * len(params))(*[L.ptr(p) for p in params])
        ms = (ctypes.c_float * self.n_launches)()
        L.check(L.lib().dvsr_edvr_forward_timed(self._h, arr, L.ptr(x), L.ptr(out), ws.data_ptr(),
                                                ws.numel() * ws.element_size(), L.stream(), ms),
                "dvsr_edvr_forward_timed")
        return list(ms)

    def work(self, nograd=False):
        """dvsr_edvr_plan_work's (nograd: dvsr_edvr_plan_work_nograd's -- the forward tape as a forward WITHOUT the gradient
        workspace runs it) nine figures by name (_WORK_KEYS): fwd / bwd _algorithmic (2 x MACs of the direct sums) and
        _executed (fp32 products as the kernels shape them: launches on the Winograd kernels issue 16/36 of their algorithmic
        multiplies), fwd_bytes (algorithmic bytes of the forward tape), and fwd / bwd _f32_pipe / _bf16_pipe (FLOPs issued to
        either matrix pipe: a launch on the exact 3-way bf16 split issues six bf16 products per fp32 product)."""
        out = (ctypes.c_double * 9)()
        fn = L.lib().dvsr_edvr_plan_work_nograd if nograd else L.lib().dvsr_edvr_plan_work
        L.check(fn(self._h, ctypes.byref(out)), "dvsr_edvr_plan_work")
        return dict(zip(_WORK_KEYS, out))

    def op_output(self, ws, index, which=0):
        """Flat view of what launch `index` wrote into the workspace (None when it writes the output tensor)."""
        ia, off, n = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_longlong()
        L.check(L.lib().dvsr_edvr_op_output(self._h, index, which, ctypes.byref(ia), ctypes.byref(off), ctypes.byref(n)),
                "dvsr_edvr_op_output")
        if not ia.value:
            return None
        return ws.view(torch.float32)[off.value:off.value + n.value]

    def op_info(self):
        """[(kind, name, flops, bytes)] per launch."""
        out = []
        kind, name = ctypes.create_string_buffer(32), ctypes.create_string_buffer(64)
        fl, by = ctypes.c_double(), ctypes.c_double()
        for i in range(self.n_launches):
            L.check(L.lib().dvsr_edvr_op_info(self._h, i, kind, 32, name, 64, ctypes.byref(fl),
                                              ctypes.byref(by)), "dvsr_edvr_op_info")
            out.append((kind.value.decode(), name.value.decode(), fl.value, by.value))
        return out

    def tensor(self, ws, name, shape=None):
        off, n = ctypes.c_longlong(), ctypes.c_longlong()
        L.check(L.lib().dvsr_edvr_tensor_info(self._h, name.encode(), ctypes.byref(off),
                                              ctypes.byref(n)), "dvsr_edvr_tensor_info")
        t = ws.view(torch.float32)[off.value:off.value + n.value]
        return t.view(shape) if shape is not None else t

    def __del__(self):
        try:
            if self._h:
                L.lib().dvsr_edvr_plan_destroy(self._h)
        except Exception:
            pass


def get_plan(cfg, b, h, w, device=None, grad_groups=1, weight_sets=1):
    """One plan per (config, shape, device): a plan owns a side stream and events of the device it was first
    used on.  A plan is single-threaded: two host threads must not drive the same plan concurrently.
    grad_groups > 1: per-group parameter gradients (dvsr_edvr_plan_create_grouped)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    # ... and per HIP stream: the plan's side stream and fork / join events belong to ONE in-flight backward, so two
    # clips adapted concurrently on two streams (adapt_video(concurrency=2)) must not share them
    key = (tuple(cfg), b, h, w, dev, torch.cuda.current_stream(dev).cuda_stream, grad_groups, weight_sets, _env_key())
    p = _plans.get(key)
    if p is None:
        _check_process_env()
        p = _plans[key] = Plan(tuple(cfg), b, h, w, grad_groups, weight_sets)
    return p


def _stash(ctx, plan, ws, x, params):
    """What backward needs.  The parameters are detached aliases of the live leaves (the tape re-packs the weights
    from them), so autograd's own saved-tensor check does not see an in-place update between forward and
    backward: the version counters -- an alias shares its base's -- are recorded here and compared in backward."""
    ctx.plan, ctx.ws, ctx.x, ctx.params = plan, ws, x, params
    ctx.versions = [p._version for p in params]


def _check_stash(ctx, what):
    if ctx.ws is None:
        raise RuntimeError("%s: backward called a second time -- the activation arena was released after the "
                           "first one (the tape is first-order and single-use, like once_differentiable in "
                           "deform_conv.py:123)" % what)
    for i, (p, v) in enumerate(zip(ctx.params, ctx.versions)):
        if p._version != v:
            raise RuntimeError("%s: parameter %d was modified in place between forward and backward (version %d -> "
                               "%d); the gradient would be taken at the wrong weights" % (what, i, v, p._version))


def _prep(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_grad_mode = threading.local()


class _TapeFunction(torch.autograd.Function):
    """autograd.Function whose forward knows whether the CALLER records a graph.  Inside Function.forward grad mode is
    always off and ctx.needs_input_grad reports requires_grad of the inputs even under torch.no_grad(), so a no-grad
    forward of a network with trainable parameters would otherwise size the activation arena for a backward that
    never comes (and miss the engine's no-grad launch geometries, which it selects from the workspace size)."""

    @classmethod
    def apply(cls, *args, **kwargs):
        _grad_mode.enabled = torch.is_grad_enabled()
        return super().apply(*args, **kwargs)


def _need_grad(ctx):
    return bool(getattr(_grad_mode, "enabled", True)) and any(ctx.needs_input_grad)


class FrozenWeights:
    """Scope for the no-grad forwards of ONE network over many clips (a video through `Video_base_model.test()`,
    test_dynavsr.py:200-204): inside `with frozen:` a no-grad EDVR forward keeps its workspace -- one per (launch plan, HIP
    stream) -- in this object, and with it the packed weights, so that the next forward of the same plan on the same stream
    skips the packing launches (dvsr_edvr_forward_packed; six launches, ~2 % of a 180x320 forward).  The packs are keyed on
    the parameters' storage and autograd version counters: an optimiser step, `load_state_dict` or any in-place update
    re-packs on the next call.  (Writes through `.data`, which do not bump the counter, are not seen: do not mix them
    with a live scope.)  Results are bit-identical to forwards outside the scope.  Dropping the object frees the workspaces."""

    def __init__(self):
        self._slots = {}

    def __enter__(self):
        self._outer = getattr(_grad_mode, "frozen", None)
        _grad_mode.frozen = self
        return self

    def __exit__(self, *exc):
        _grad_mode.frozen = self._outer
        return False

    def slot(self, plan, device, params):
        """-> (workspace, packs_valid) for a no-grad forward of `plan` on the current stream."""
        key = (plan.key, device.index, torch.cuda.current_stream(device).cuda_stream)
        sig = tuple((p.data_ptr(), p._version) for p in params)
        have = self._slots.get(key)
        if have is not None and have[1] == sig:
            return have[0], True
        ws = have[0] if have is not None else torch.empty(plan.workspace_bytes(False), dtype=torch.uint8, device=device)
        self._slots[key] = (ws, sig)
        return ws, False


class EdvrFunction(_TapeFunction):
    @staticmethod
    def forward(ctx, x, cfg, keep_ws, *params):
        if not x.is_cuda:
            raise RuntimeError("dynavsr_amd EDVR runs on the MI355X only (input is on %s); there is "
                               "no CPU fallback" % x.device)
        x = _prep(x)
        b, n, c, h, w = x.shape
        if n != cfg[1] or c != 3:
            raise RuntimeError("EDVR expects [B,%d,3,H,W], got %s" % (cfg[1], tuple(x.shape)))
        plan = get_plan(cfg, b, h, w, x.device)
        if len(params) != plan.n_params:
            raise RuntimeError("EDVR engine expects %d parameter tensors, got %d"
                               % (plan.n_params, len(params)))
        leaves = params                  # (version counters are read off the caller's tensors, not off converted copies)
        params = [_prep(p.detach()) for p in params]
        need_grad = _need_grad(ctx)
        frozen = None if (need_grad or keep_ws is not None) else getattr(_grad_mode, "frozen", None)
        if frozen is not None:
            ws, packed = frozen.slot(plan, x.device, leaves)
        else:
            ws, packed = torch.empty(plan.workspace_bytes(need_grad), dtype=torch.uint8, device=x.device), False
        out = x.new_empty((b, 3, cfg[5] * h, cfg[5] * w))
        plan.forward(params, x, out, ws, packed=packed)
        if need_grad:
            _stash(ctx, plan, ws, x, params)
        if keep_ws is not None:
            keep_ws.append((plan, ws))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        _check_stash(ctx, "EDVR")
        gout = _prep(gout)
        gparams = [torch.empty_like(p) for p in ctx.params]
        gx = torch.empty_like(ctx.x) if ctx.needs_input_grad[0] else None
        ctx.plan.backward(ctx.params, ctx.x, gout, gparams, gx, ctx.ws)
        ctx.ws = None  # release the activation arena
        return (gx, None, None) + tuple(gparams)


class EdvrStackedFunction(_TapeFunction):
    """K clips through ONE tape with PER-CLIP parameter gradients (dvsr_edvr_plan_create_grouped).

    x: [K,N,3,H,W]; every parameter arrives STACKED, [K, *shape]: slice k is frame k's private copy of the weights
    (test_dynavsr.py:208 deep-copies the networks for every frame).  per_slice = False: all K slices hold the SAME values
    when this runs (contract) -- the first inner step, where every copy still equals the un-adapted network -- and the
    tape reads slice 0.  per_slice = True: clip k is convolved with slice k (dvsr_edvr_plan_create_ex, weight_sets = K):
    the copies have diverged (later inner steps, the adapted forwards).
    backward returns d loss_k / d theta in slice k (the caller sums the per-clip losses, so the incoming gradient of
    out[k] is that of loss_k alone): an elementwise optimiser stepping the stacked tensors once performs the K
    independent inner updates of the sequential loop."""

    @staticmethod
    def forward(ctx, x, cfg, per_slice, *stacked):
        if not x.is_cuda:
            raise RuntimeError("dynavsr_amd EDVR runs on the MI355X only (input is on %s); there is no CPU fallback" % x.device)
        x = _prep(x)
        k, n, c, h, w = x.shape
        if n != cfg[1] or c != 3:
            raise RuntimeError("EDVR expects [K,%d,3,H,W], got %s" % (cfg[1], tuple(x.shape)))
        if any(p.shape[0] != k for p in stacked):
            raise RuntimeError("EDVR (stacked): every parameter must be [K=%d, ...]" % k)
        plan = get_plan(cfg, k, h, w, x.device, grad_groups=k, weight_sets=k if per_slice else 1)
        if len(stacked) != plan.n_params:
            raise RuntimeError("EDVR engine expects %d parameter tensors, got %d" % (plan.n_params, len(stacked)))
        stacked = [_prep(p.detach()) for p in stacked]
        params = stacked if per_slice else [p[0] for p in stacked]
        need_grad = _need_grad(ctx)
        ws = torch.empty(plan.workspace_bytes(need_grad), dtype=torch.uint8, device=x.device)
        out = x.new_empty((k, 3, cfg[5] * h, cfg[5] * w))
        plan.forward(params, x, out, ws)
        if need_grad:
            _stash(ctx, plan, ws, x, stacked)
            ctx.per_slice = per_slice
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        _check_stash(ctx, "EDVR (stacked)")
        gout = _prep(gout)
        gparams = [torch.empty_like(p) for p in ctx.params]          # [K, *shape]: group-major, as the plan writes them
        gx = torch.empty_like(ctx.x) if ctx.needs_input_grad[0] else None
        ctx.plan.backward(ctx.params if ctx.per_slice else [p[0] for p in ctx.params], ctx.x, gout, gparams, gx, ctx.ws)
        ctx.ws = None
        return (gx, None, None) + tuple(gparams)


# ---- down-scaling estimators (csrc/engine.hip: dvsr_estimator_*) ---------------------------------
MFDN, SFDN = 0, 1
_eplans = {}


class EstimatorPlan:
    """Owns one dvsr_estimator_plan: MFDN / SFDN forward + parameter-gradient tape for one shape."""

    def __init__(self, cfg, b, h, w, grad_groups=1, weight_sets=1):
        self.cfg = dict(zip(("kind", "nf", "in_nc", "scale", "nframes"), cfg))
        self.b, self.h, self.w, self.grad_groups, self.weight_sets = b, h, w, grad_groups, weight_sets
        self._h = ctypes.c_void_p()
        L.check(L.lib().dvsr_estimator_plan_create_ex(L.EstimatorConfig(*cfg), b, h, w, grad_groups, weight_sets,
                                                      ctypes.byref(self._h)), "dvsr_estimator_plan_create_ex")
        self.n_params = L.lib().dvsr_estimator_num_params(self._h)
        self.n_launches = L.lib().dvsr_estimator_num_launches(self._h, 0)
        self.n_backward_launches = L.lib().dvsr_estimator_num_launches(self._h, 1)

    def workspace_bytes(self, need_grad):
        return int(L.lib().dvsr_estimator_workspace_bytes(self._h, int(need_grad)))

    def forward(self, params, x, out, ws):
        arr = (ctypes.c_void_p * len(params))(*[L.ptr(p) for p in params])
        L.check(L.lib().dvsr_estimator_forward(self._h, arr, L.ptr(x), L.ptr(out), ws.data_ptr(),
                                               ws.numel() * ws.element_size(), L.stream()),
                "dvsr_estimator_forward")

    def backward(self, params, x, gout, gparams, ws):
        arr = (ctypes.c_void_p * len(params))(*[L.ptr(p) for p in params])
        garr = (ctypes.c_void_p * len(gparams))(*[L.ptr(g) for g in gparams])
        L.check(L.lib().dvsr_estimator_backward(self._h, arr, L.ptr(x), L.ptr(gout), garr, ws.data_ptr(),
                                                ws.numel() * ws.element_size(), L.stream()),
                "dvsr_estimator_backward")

    def work(self):
        """As Plan.work(), for the estimator's tapes."""
        out = (ctypes.c_double * 9)()
        L.check(L.lib().dvsr_estimator_plan_work(self._h, ctypes.byref(out)), "dvsr_estimator_plan_work")
        return dict(zip(_WORK_KEYS, out))

    def __del__(self):
        try:
            if self._h:
                L.lib().dvsr_estimator_plan_destroy(self._h)
        except Exception:
            pass


def get_estimator_plan(cfg, b, h, w, device=None, grad_groups=1, weight_sets=1):
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (tuple(cfg), b, h, w, dev, torch.cuda.current_stream(dev).cuda_stream, grad_groups, weight_sets, _env_key())
    p = _eplans.get(key)
    if p is None:
        _check_process_env()
        p = _eplans[key] = EstimatorPlan(tuple(cfg), b, h, w, grad_groups, weight_sets)
    return p


class EstimatorFunction(_TapeFunction):
    """x: [B,C,T,H,W] (MFDN) or [B,C,H,W] (SFDN) -> same rank, spatially / scale.  The clip is data:
    no gradient is produced for x (LRestimator_model.py feeds `LQs`, test_dynavsr.py:237-241)."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        if not x.is_cuda:
            raise RuntimeError("dynavsr_amd estimator (MFDN/SFDN) runs on the MI355X only (input is on %s); "
                               "there is no CPU fallback" % x.device)
        if ctx.needs_input_grad[0]:
            raise RuntimeError("the estimator's input clip is data: gradients w.r.t. it are not provided")
        x = _prep(x)
        kind, nf, in_nc, scale, nframes = cfg
        if kind == MFDN:
            b, c, t, h, w = x.shape
            if t != nframes or c != in_nc:
                raise RuntimeError("MFDN expects [B,%d,%d,H,W], got %s" % (in_nc, nframes, tuple(x.shape)))
            oshape = (b, c, t, h // scale, w // scale)
        else:
            b, c, h, w = x.shape
            oshape = (b, c, h // scale, w // scale)
        plan = get_estimator_plan(cfg, b, h, w, x.device)
        if len(params) != plan.n_params:
            raise RuntimeError("estimator engine expects %d parameter tensors, got %d" % (plan.n_params, len(params)))
        params = [_prep(p.detach()) for p in params]
        need_grad = _need_grad(ctx)
        ws = torch.empty(plan.workspace_bytes(need_grad), dtype=torch.uint8, device=x.device)
        out = x.new_empty(oshape)
        plan.forward(params, x, out, ws)
        if need_grad:
            _stash(ctx, plan, ws, x, params)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        _check_stash(ctx, "estimator")
        gout = _prep(gout)
        gparams = [torch.empty_like(p) for p in ctx.params]
        ctx.plan.backward(ctx.params, ctx.x, gout, gparams, ctx.ws)
        ctx.ws = None
        return (None, None) + tuple(gparams)


class EstimatorStackedFunction(_TapeFunction):
    """The estimator on K clips with per-clip parameter gradients: x [K,C,T,H,W] (MFDN) / [K*T,C,H,W] (SFDN, the frames
    of a clip consecutive), every parameter stacked [K, *shape] with K equal slices (see EdvrStackedFunction)."""

    @staticmethod
    def forward(ctx, x, cfg, per_slice, *stacked):
        if not x.is_cuda:
            raise RuntimeError("dynavsr_amd estimator (MFDN/SFDN) runs on the MI355X only (input is on %s); "
                               "there is no CPU fallback" % x.device)
        if ctx.needs_input_grad[0]:
            raise RuntimeError("the estimator's input clip is data: gradients w.r.t. it are not provided")
        x = _prep(x)
        kind, nf, in_nc, scale, nframes = cfg
        if kind == MFDN:
            b, c, t, h, w = x.shape
            if t != nframes or c != in_nc:
                raise RuntimeError("MFDN expects [K,%d,%d,H,W], got %s" % (in_nc, nframes, tuple(x.shape)))
            oshape = (b, c, t, h // scale, w // scale)
        else:
            b, c, h, w = x.shape
            oshape = (b, c, h // scale, w // scale)
        k = stacked[0].shape[0] if stacked else 0
        if k < 1 or b % k or any(p.shape[0] != k for p in stacked):   # (SFDN in 'image' mode: batch = K clips x T frames)
            raise RuntimeError("estimator (stacked): every parameter must be [K, ...] with K dividing the batch %d" % b)
        plan = get_estimator_plan(cfg, b, h, w, x.device, grad_groups=k, weight_sets=k if per_slice else 1)
        if len(stacked) != plan.n_params:
            raise RuntimeError("estimator engine expects %d parameter tensors, got %d" % (plan.n_params, len(stacked)))
        stacked = [_prep(p.detach()) for p in stacked]
        need_grad = _need_grad(ctx)
        ws = torch.empty(plan.workspace_bytes(need_grad), dtype=torch.uint8, device=x.device)
        out = x.new_empty(oshape)
        plan.forward(stacked if per_slice else [p[0] for p in stacked], x, out, ws)
        if need_grad:
            _stash(ctx, plan, ws, x, stacked)
            ctx.per_slice = per_slice
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        _check_stash(ctx, "estimator (stacked)")
        gout = _prep(gout)
        gparams = [torch.empty_like(p) for p in ctx.params]
        ctx.plan.backward(ctx.params if ctx.per_slice else [p[0] for p in ctx.params], ctx.x, gout, gparams, ctx.ws)
        ctx.ws = None
        return (None, None, None) + tuple(gparams)
