"""ctypes binding of libdynavsr_hip.so (the C ABI declared in include/dynavsr_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, a RuntimeError
carrying dvsr_last_error() is raised.  The product path never routes through torch CPU ops or
the oracle.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch

_runtime = {"configured": False, "hw_queues": None, "effective": None}


def configure_runtime(hw_queues=6):
    """Process-wide runtime settings the execution plans want, applied EXPLICITLY (never at import).

    ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and streams that share a queue run
    behind each other.  The plans put their weight gradients on a side stream; with the default, as soon as another
    component of the process holds streams (an initialised RCCL communicator does) that side stream shares a queue and
    the inner MAML step measures 10.0 instead of 8.2 ms.  Two more queues are enough.  The runtime reads
    the variable when HIP is initialised (the first device call), so this must run before that: `create_model`,
    `bench.py` and the tools call it first thing.  A value the user exported is never overridden.  Returns
    {"hw_queues": the value in force, "effective": False when HIP was already initialised and the value came too late}."""
    if not _runtime["configured"]:
        user = os.environ.get("GPU_MAX_HW_QUEUES")
        late = torch.cuda.is_initialized()
        if user is None and not late:
            os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
        _runtime.update(configured=True, hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"),
                        effective=(user is not None) or not late)
    return {"hw_queues": _runtime["hw_queues"], "effective": _runtime["effective"]}


def runtime_report(device=None):
    """What the process actually got: the hardware-queue setting, whether it came in time, and the MEASURED answer to the
    question it exists for -- does the plans' weight-gradient side stream run beside the current stream of `device`
    (dvsr_side_stream_overlaps: 1 / 0 / -1 unknown)?  bench.py records it in its line."""
    r = configure_runtime()
    overlaps = None
    if torch.cuda.is_available():
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            overlaps = int(lib().dvsr_side_stream_overlaps(stream()))
    return {"hw_queues": r["hw_queues"], "effective": r["effective"], "side_stream_overlaps": overlaps}


_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libdynavsr_hip.so")
_lib = None

ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2


class Conv2dDesc(Structure):
    _fields_ = [("x0", c_void_p), ("x1", c_void_p), ("w", c_void_p), ("bias", c_void_p),
                ("res", c_void_p), ("y", c_void_p),
                ("N", c_int), ("c0", c_int), ("c1", c_int), ("H", c_int), ("W", c_int),
                ("Cout", c_int), ("ks", c_int), ("stride", c_int), ("pad", c_int), ("act", c_int),
                ("pixel_shuffle", c_int), ("x1_bdiv", c_int),
                ("x0_bstride", c_longlong), ("x1_bstride", c_longlong)]


class EdvrConfig(Structure):
    _fields_ = [(k, c_int) for k in ("nf", "nframes", "groups", "front_RBs", "back_RBs", "scale",
                                     "center", "bf16_mfma")]


class EstimatorConfig(Structure):
    _fields_ = [(k, c_int) for k in ("kind", "nf", "in_nc", "scale", "nframes")]


def _declare(lib):
    P, I, F, LL = c_void_p, c_int, c_float, c_longlong
    sig = {
        "dvsr_last_error": (c_char_p, []),
        "dvsr_version": (I, []),
        "dvsr_mdcn_forward": (I, [P] * 6 + [I] * 13 + [P]),
        "dvsr_mdcn_forward_fast_workspace_bytes": (c_size_t, [I, I, I]),
        "dvsr_mdcn_forward_fast": (I, [P] * 6 + [I] * 7 + [P, c_size_t, P]),
        "dvsr_mdcn_pack_forward": (I, [P] * 5 + [I] * 13 + [P]),
        "dvsr_conv2d_forward": (I, [POINTER(Conv2dDesc), P]),
        "dvsr_upsample_bilinear_forward": (I, [P, P, LL, I, I, I, F, P]),
        "dvsr_upsample_bilinear_backward": (I, [P, P, LL, I, I, I, F, I, P]),
        "dvsr_pool3s2_forward": (I, [P, P, P, LL, I, I, P]),
        "dvsr_pool3s2_backward": (I, [P, P, P, P, LL, I, I, P]),
        "dvsr_tsa_gate_forward": (I, [P] * 5 + [I, I, I, LL, P]),
        "dvsr_tsa_gate_backward": (I, [P] * 8 + [I, I, I, LL, P]),
        "dvsr_tsa_blend_forward": (I, [P] * 4 + [LL, P]),
        "dvsr_tsa_blend_backward": (I, [P] * 5 + [LL, P]),
        "dvsr_edvr_plan_create": (I, [POINTER(EdvrConfig), I, I, I, POINTER(c_void_p)]),
        "dvsr_edvr_plan_create_grouped": (I, [POINTER(EdvrConfig), I, I, I, I, POINTER(c_void_p)]),
        "dvsr_edvr_plan_create_ex": (I, [POINTER(EdvrConfig), I, I, I, I, I, POINTER(c_void_p)]),
        "dvsr_edvr_plan_destroy": (None, [P]),
        "dvsr_edvr_num_params": (I, [P]),
        "dvsr_edvr_num_launches": (I, [P]),
        "dvsr_edvr_workspace_bytes": (c_size_t, [P, I]),
        "dvsr_edvr_forward": (I, [P, POINTER(c_void_p), P, P, P, c_size_t, P]),
        "dvsr_edvr_forward_packed": (I, [P, POINTER(c_void_p), P, P, P, c_size_t, P]),
        "dvsr_edvr_backward": (I, [P, POINTER(c_void_p), P, P, POINTER(c_void_p), P, P, c_size_t, P]),
        "dvsr_edvr_num_backward_launches": (I, [P]),
        "dvsr_estimator_plan_create": (I, [POINTER(EstimatorConfig), I, I, I, POINTER(c_void_p)]),
        "dvsr_estimator_plan_create_grouped": (I, [POINTER(EstimatorConfig), I, I, I, I, POINTER(c_void_p)]),
        "dvsr_estimator_plan_create_ex": (I, [POINTER(EstimatorConfig), I, I, I, I, I, POINTER(c_void_p)]),
        "dvsr_estimator_plan_destroy": (None, [P]),
        "dvsr_estimator_num_params": (I, [P]),
        "dvsr_estimator_num_launches": (I, [P, I]),
        "dvsr_estimator_workspace_bytes": (c_size_t, [P, I]),
        "dvsr_estimator_forward": (I, [P, POINTER(c_void_p), P, P, P, c_size_t, P]),
        "dvsr_estimator_backward": (I, [P, POINTER(c_void_p), P, P, POINTER(c_void_p), P, c_size_t, P]),
        "dvsr_conv2d_backward_workspace_bytes": (c_size_t, [POINTER(Conv2dDesc)]),
        "dvsr_conv2d_backward": (I, [POINTER(Conv2dDesc), P, P, P, P, P, P, c_size_t, P]),
        "dvsr_mdcn_backward_workspace_bytes": (c_size_t, [I] * 10),
        "dvsr_mdcn_backward": (I, [P] * 10 + [I] * 12 + [P, c_size_t, P]),
        "dvsr_adam_step": (I, [POINTER(c_void_p)] * 4 + [POINTER(LL), I, F, F, F, F, F, I, P]),
        "dvsr_sgd_step": (I, [POINTER(c_void_p)] * 2 + [POINTER(LL), I, F, F, P]),
        "dvsr_replicate_tensors": (I, [POINTER(c_void_p)] * 2 + [POINTER(LL), I, I, P]),
        "dvsr_degrade_apply": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
        "dvsr_frame_metrics_workspace_bytes": (c_size_t, [I, I, I]),
        "dvsr_frame_metrics": (I, [P, P, I, I, I, F, F, P, P, P, c_size_t, P]),
        "dvsr_charbonnier_workspace_bytes": (c_size_t, []),
        "dvsr_charbonnier_forward": (I, [P, P, P, LL, F, P, c_size_t, P]),
        "dvsr_conv2d_packed_workspace_bytes": (c_size_t, [POINTER(Conv2dDesc)]),
        "dvsr_conv2d_forward_packed": (I, [POINTER(Conv2dDesc), P, c_size_t, P]),
        "dvsr_conv2d_dgrad_packed": (I, [POINTER(Conv2dDesc), P, P, P, c_size_t, P]),
        "dvsr_conv2d_packed_geometry": (I, [POINTER(Conv2dDesc), POINTER(ctypes.c_int * 4)]),
        "dvsr_conv2d_wgrad_bf16": (I, [POINTER(Conv2dDesc), P, P, P, P, c_size_t, P]),
        "dvsr_conv2d_wgrad_split3": (I, [POINTER(Conv2dDesc), P, P, P, P, c_size_t, P]),
        "dvsr_flow_warp_forward": (I, [P, P, P, I, I, I, I, LL, P]),
        "dvsr_flow_warp_backward": (I, [P, P, P, P, P, I, I, I, I, LL, P]),
        "dvsr_avgpool2_forward": (I, [P, P, LL, I, I, P]),
        "dvsr_avgpool2_backward": (I, [P, P, LL, I, I, I, P]),
        "dvsr_resize_bilinear_ac_forward": (I, [P, P, LL, I, I, I, I, F, LL, P]),
        "dvsr_resize_bilinear_ac_backward": (I, [P, P, LL, I, I, I, I, F, LL, P]),
        "dvsr_upsample_bicubic_ac_forward": (I, [P, P, LL, I, I, I, P]),
        "dvsr_upsample_bicubic_ac_backward": (I, [P, P, LL, I, I, I, P]),
        "dvsr_channel_affine": (I, [P, P, P, P, I, I, LL, LL, LL, I, P]),
        "dvsr_batchnorm_workspace_bytes": (c_size_t, [I]),
        "dvsr_batchnorm_forward": (I, [P] * 8 + [I, I, LL, I, F, F, I, P, c_size_t, P]),
        "dvsr_batchnorm_backward": (I, [P] * 9 + [I, I, LL, I, I, P, c_size_t, P]),
        "dvsr_temporal_gather3_forward": (I, [P, P, I, I, I, LL, I, P]),
        "dvsr_temporal_gather3_backward": (I, [P, P, I, I, I, LL, I, P]),
        "dvsr_dynamic_filter_forward": (I, [P, P, P, P, I, I, I, I, I, P]),
        "dvsr_dynamic_filter_backward": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
        "dvsr_conv3x3_small_cout": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
        "dvsr_conv1x1_dual": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
        "dvsr_patch_gather_forward": (I, [P, P, POINTER(c_int), POINTER(c_int), I, I, I, I, I, I, P]),
        "dvsr_patch_gather_backward": (I, [P, P, POINTER(c_int), POINTER(c_int), I, I, I, I, I, I, P]),
        "dvsr_l1_tail_forward": (I, [P, P, P, F, P, LL, P, c_size_t, P]),
        "dvsr_l1_tail_backward": (I, [P, P, P, F, P, LL, P]),
        "dvsr_charbonnier_backward": (I, [P, P, P, P, LL, F, P]),
        "dvsr_charbonnier_forward_grouped": (I, [P, P, P, LL, I, F, P, c_size_t, P]),
        "dvsr_charbonnier_backward_grouped": (I, [P, P, P, P, LL, I, F, P]),
        "dvsr_l1_tail_forward_grouped": (I, [P, P, P, F, P, LL, I, P, c_size_t, P]),
        "dvsr_l1_tail_backward_grouped": (I, [P, P, P, F, P, LL, I, P]),
        "dvsr_edvr_op_info": (I, [P, I, c_char_p, I, c_char_p, I, POINTER(ctypes.c_double),
                                  POINTER(ctypes.c_double)]),
        "dvsr_edvr_forward_timed": (I, [P, POINTER(c_void_p), P, P, P, c_size_t, P, POINTER(c_float)]),
        "dvsr_debug_mfma_peak": (LL, [P, I, I, I, I, P]),
        "dvsr_debug_mfma_shadow": (I, [P, P, I, I, I, I, I, P]),
        "dvsr_edvr_tensor_info": (I, [P, c_char_p, POINTER(LL), POINTER(LL)]),
        "dvsr_edvr_plan_work": (I, [P, POINTER(ctypes.c_double * 9)]),
        "dvsr_edvr_plan_work_nograd": (I, [P, POINTER(ctypes.c_double * 9)]),
        "dvsr_side_stream_overlaps": (I, [P]),
        "dvsr_edvr_op_output": (I, [P, I, I, POINTER(c_int), POINTER(LL), POINTER(LL)]),
        "dvsr_estimator_plan_work": (I, [P, POINTER(ctypes.c_double * 9)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sig


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "libdynavsr_hip.so is missing (%s). Build it with `python -m dynavsr_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no fallback path." % SO_PATH)
        # DVSR_HIP_LIB: load another build of the same library (tools/ use it for the trace build)
        _lib = ctypes.CDLL(os.environ.get("DVSR_HIP_LIB", SO_PATH))
        _lib._signatures = _declare(_lib)
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().dvsr_last_error().decode("utf-8", "replace")
        raise RuntimeError("libdynavsr_hip %s failed (rc=%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a tensor that must already be fp32, contiguous and on a HIP device."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libdynavsr_hip needs a tensor on the GPU, got device %s" % t.device)
    if t.dtype != torch.float32:
        raise RuntimeError("libdynavsr_hip computes in fp32, got %s" % t.dtype)
    if not t.is_contiguous():
        raise RuntimeError("libdynavsr_hip needs contiguous tensors")  # deform_conv_cuda.cpp:493-494
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
