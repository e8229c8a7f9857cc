/*
 * dynavsr_hip.h -- C ABI of libdynavsr_hip.so: MI355X (gfx950) kernels for the EDVR hot path of
 * DynaVSR (PCD deformable alignment, TSA fusion, reconstruction) and its inner MAML step.
 *
 * Contract (all entry points):
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer on the current HIP device,
 *     fp32, contiguous NCHW unless stated; tensors are borrowed, never owned or freed;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream), no
 *     internal synchronisation, no allocation: scratch comes from a caller workspace;
 *   - returns 0 on success or a negative DVSR_ERR_* code and never throws; the message of the
 *     last failure on the calling thread is dvsr_last_error();
 *   - the stateless entry points are re-entrant (the only global mutable state is that thread-local error
 *     string).  PLAN objects (dvsr_edvr_plan, dvsr_estimator_plan) are NOT: a plan owns a side stream and
 *     fork/join events that its backward records on, so one plan must be driven by one host thread at a time
 *     (the reference's caller is single-threaded too: the autograd engine thread).  Different plans
 *     may be used from different threads concurrently.
 *
 * Each declaration cites the reference interface it replaces (paths under codes/ of
 * esw0116/DynaVSR).  The reference's only native boundary is the pybind11 module
 * `deform_conv_cuda` (models/archs/dcn/src/deform_conv_cuda.cpp:681-695); everything else the
 * reference gets from cuDNN/ATen through torch.nn, which these kernels replace one-for-one.
 */
#ifndef DYNAVSR_HIP_H_
#define DYNAVSR_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVSR_OK 0
#define DVSR_ERR_INVALID (-1)
#define DVSR_ERR_UNSUPPORTED (-2)
#define DVSR_ERR_WORKSPACE (-3)
#define DVSR_ERR_HIP (-4)

#define DVSR_ACT_NONE 0
#define DVSR_ACT_LRELU 1 /* LeakyReLU(0.1): models/archs/EDVR_arch.py:93,161,252 */
#define DVSR_ACT_RELU 2  /* models/archs/arch_util.py:50 */

typedef void* dvsr_stream_t; /* hipStream_t */

const char* dvsr_last_error(void);
int dvsr_version(void);

/* ---- modulated deformable convolution (DCNv2) ------------------------------------------------
 * Replaces modulated_deform_conv_cuda_forward (deform_conv_cuda.cpp:486-564; kernels
 * deform_conv_cuda_kernel.cu:466-496,569-632) with ONE fused kernel: bilinear sampling, mask
 * and the Cout x (C*kh*kw) contraction happen per tile in LDS/MFMA, no im2col buffer in HBM.
 *   x [N,C,H,W]  offset [N,dg*2*kh*kw,Ho,Wo] ((dy,dx) interleaved per tap)  mask [N,dg*kh*kw,Ho,Wo]
 *   w [Cout,C/groups,kh,kw]  b [Cout] or NULL  ->  out [N,Cout,Ho,Wo] (overwritten)
 * Supported: kh=kw=3, groups=1, C/dg in {4,8,16}; anything else returns DVSR_ERR_UNSUPPORTED
 * (the reference raises through AT_ERROR, cpp:506-511).  `act` is applied to the result
 * (EDVR_arch.py:103,126 apply LeakyReLU right after the DCN; pass DVSR_ACT_NONE for the op alone).
 */
int dvsr_mdcn_forward(const float* x, const float* offset, const float* mask, const float* w,
                      const float* b, float* out, int N, int C, int H, int W, int Cout, int kh,
                      int kw, int stride, int pad, int dil, int groups, int dg, int act,
                      dvsr_stream_t stream);

/* Fast path of dvsr_mdcn_forward for the EDVR configuration (3x3, stride = pad = dil = 1, groups = 1,
 * C/dg a multiple of 8 -- EDVR-M 64/8, EDVR-L 128/8): 8-channel slices of a deformable group's input planes
 * are staged in LDS and sampled from there (8 LDS
 * reads per (pixel, tap) instead of 32 global gathers); samples that leave the staged window
 * (|offset| > 4 px) fall back to exact global gathers.  The workspace receives the packed weights. */
size_t dvsr_mdcn_forward_fast_workspace_bytes(int C, int Cout, int dg);
int dvsr_mdcn_forward_fast(const float* x, const float* offset, const float* mask, const float* w,
                           const float* b, float* out, int N, int C, int H, int W, int Cout, int dg, int act,
                           void* workspace, size_t workspace_bytes, dvsr_stream_t stream);

/* Replaces modulated_deform_conv_cuda_backward (deform_conv_cuda.cpp:566-679).  grad_out = gradient
 * w.r.t. the op's output (act = NONE).  gx is ACCUMULATED into with fp32 atomics (zero it first, the
 * reference's caller does: deform_conv.py:128); goffset/gmask/gw/gb are overwritten; gx/gw/gb may
 * be NULL.  Workspace: the EDVR configuration (3x3, stride / pad / dilation 1, 8 channels per group) runs the fused
 * kernel, whose scratch is the per-workgroup [Cout][72] weight-gradient partials and (r06) the weights re-laid-out as the
 * kernel's first-phase operands, once per call (no column buffer: dcol and the sampled columns stay on chip; with Cout = 64,
 * W % 4 == 0 and 16-byte aligned x / grad_out both contractions run on the bf16 MFMA under the exact 3-way operand split:
 * fp32 results); other configurations take the three-kernel path, which also needs the [C*9, Ho*Wo] column buffer.
 * dvsr_mdcn_backward_workspace_bytes() returns the larger of the two. */
size_t dvsr_mdcn_backward_workspace_bytes(int N, int C, int H, int W, int Cout, int kh, int kw, int stride,
                                          int pad, int dil);
int dvsr_mdcn_backward(const float* x, const float* offset, const float* mask, const float* w,
                       const float* grad_out, float* gx, float* goffset, float* gmask, float* gw, float* gb,
                       int N, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                       int groups, int dg, void* workspace, size_t workspace_bytes, dvsr_stream_t stream);

/* Same op fed by the RAW output `om` [N, 3*dg*kh*kw, Ho, Wo] of ModulatedDeformConvPack's
 * conv_offset_mask (deform_conv.py:274-291): offset = om[:, :2*dg*K], mask = sigmoid(om[:, 2*dg*K:]);
 * the chunk/cat/sigmoid launches of the reference are folded into the sampler. */
int dvsr_mdcn_pack_forward(const float* x, const float* om, const float* w, const float* b,
                           float* out, int N, int C, int H, int W, int Cout, int kh, int kw,
                           int stride, int pad, int dil, int groups, int dg, int act,
                           dvsr_stream_t stream);

/* ---- dense convolution ------------------------------------------------------------------------
 * Replaces nn.Conv2d (+ the torch.cat feeding it, + the activation / residual add /
 * nn.PixelShuffle(2) following it) everywhere in EDVR_arch.py / arch_util.py:34-52.
 * LDS-staged implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), no im2col. */
typedef struct dvsr_conv2d_desc {
  const float* x0;   /* [N][c0][H][W] */
  const float* x1;   /* optional 2nd input, concatenated after x0 along C: [ceil(N/x1_bdiv)][c1][H][W]; NULL if c1==0 */
  const float* w;    /* [Cout][c0+c1][ks][ks] (OIHW) */
  const float* bias; /* [Cout] or NULL */
  const float* res;  /* optional tensor added AFTER the activation, shaped like y; NULL = none */
  float* y;          /* [N][Cout][Ho][Wo]; with pixel_shuffle=2: [N][Cout/4][2Ho][2Wo] */
  int N, c0, c1, H, W, Cout;
  int ks;            /* 1 or 3 */
  int stride;        /* 1 or 2 */
  int pad;           /* zero padding, must be ks/2 */
  int act;           /* DVSR_ACT_* applied to conv+bias */
  int pixel_shuffle; /* 0, or 2 = nn.PixelShuffle(2) fused into the store (EDVR_arch.py:247,303-305) */
  int x1_bdiv;       /* x1 batch index = n / x1_bdiv (>=1): one reference frame shared by N frames */
  long long x0_bstride; /* elements between batch items of x0; 0 = dense (c0*H*W) */
  long long x1_bstride; /* same for x1; 0 = dense (c1*H*W) */
} dvsr_conv2d_desc;

int dvsr_conv2d_forward(const dvsr_conv2d_desc* d, dvsr_stream_t stream);
/* Backward of the plain layout (pixel_shuffle=0, x1_bdiv<=1): `gy` = gradient w.r.t. the
 * PRE-activation output.  gx0/gx1/gw/gb may be NULL to skip; all are overwritten.  Data gradient =
 * the same MFMA kernel on the transposed, tap-mirrored weight view (zero-dilated for stride 2);
 * weight/bias gradient = split-K MFMA kernel + deterministic reduce (conv2d_wgrad.hip). */
size_t dvsr_conv2d_backward_workspace_bytes(const dvsr_conv2d_desc* d);
int dvsr_conv2d_backward(const dvsr_conv2d_desc* d, const float* gy, float* gx0, float* gx1, float* gw,
                         float* gb, void* workspace, size_t workspace_bytes, dvsr_stream_t stream);

/* ---- streaming (HBM-bound) ops ------------------------------------------------------------------
 * F.interpolate(scale_factor=S, mode='bilinear', align_corners=False) * mul
 * (EDVR_arch.py:107-108,111,116-117,120,192,197,311); x [planes,H,W] -> y [planes,S*H,S*W]. */
int dvsr_upsample_bilinear_forward(const float* x, float* y, long long planes, int H, int W,
                                   int scale, float mul, dvsr_stream_t stream);
int dvsr_upsample_bilinear_backward(const float* gy, float* gx, long long planes, int H, int W,
                                    int scale, float mul, int accumulate, dvsr_stream_t stream);
/* nn.MaxPool2d(3,2,1) and nn.AvgPool2d(3,2,1) of the same tensor in one pass
 * (EDVR_arch.py:149-150,184-185,189-190). */
int dvsr_pool3s2_forward(const float* x, float* ymax, float* yavg, long long planes, int H, int W,
                         dvsr_stream_t stream);
int dvsr_pool3s2_backward(const float* x, const float* gmax, const float* gavg, float* gx,
                          long long planes, int H, int W, dvsr_stream_t stream);
/* TSA temporal attention gate (EDVR_arch.py:169-176): cor = sigmoid(sum_c emb*emb_ref) [B,N,HW],
 * gated = aligned * cor [B,N*C,HW]. */
int dvsr_tsa_gate_forward(const float* emb, const float* emb_ref, const float* aligned, float* cor,
                          float* gated, int B, int N, int C, long long HW, dvsr_stream_t stream);
int dvsr_tsa_gate_backward(const float* emb, const float* emb_ref, const float* aligned,
                           const float* cor, const float* g_gated, float* g_emb, float* g_emb_ref,
                           float* g_aligned, int B, int N, int C, long long HW,
                           dvsr_stream_t stream);
/* out = fea * sigmoid(att) * 2 + att_add (EDVR_arch.py:200-202). */
int dvsr_tsa_blend_forward(const float* fea, const float* att, const float* att_add, float* out,
                           long long n, dvsr_stream_t stream);
int dvsr_tsa_blend_backward(const float* fea, const float* att, const float* g, float* g_fea,
                            float* g_att_io, long long n, dvsr_stream_t stream);

/* ---- whole-network execution plan ----------------------------------------------------------------
 * Replaces `netG(var_L)` = EDVR.forward (models/archs/EDVR_arch.py:254-313, incl. PCD_Align :95-128
 * and TSA_Fusion :163-203) as called from VideoBaseModel.calculate_loss/test
 * (models/Video_base_model.py:191-201): one call enqueues every kernel of the backbone on `stream`.
 * `params` is a HOST array of dvsr_edvr_num_params() device pointers in the reference's state-dict
 * order (conv_first.weight, conv_first.bias, feature_extraction.0.conv1.weight, ... conv_last.bias).
 * x [B,nframes,3,H,W] -> out [B,3,scale*H,scale*W].  The workspace keeps every activation
 * (it is what a later backward call reads), so it must stay untouched between the two calls. */
typedef struct dvsr_edvr_config {
  int nf, nframes, groups, front_RBs, back_RBs, scale, center;
  /* 0: every contraction on the exact-fp32 MFMA (default; the parity configuration).
   * 1: the 3x3 stride-1 convolutions (forward and data gradient) round their operands to bf16 and run on
   *    v_mfma_f32_32x32x16_bf16 with fp32 accumulation; activations, weights, gradients, the DCN, the 1x1 /
   *    stride-2 convs and all weight gradients stay fp32 (BASELINE configs[4], "EDVR-L, bf16 MFMA path").
   * 2: experimental.  The same convolutions with every fp32 operand split exactly into three bf16 pieces
   *    (8 + 8 + 8 mantissa bits) and the six partial products above 2^-24 accumulated in fp32 on the bf16
   *    MFMA: fp32-level results (held to the fp32 parity bars by the tests), 6 x 32 instead of 8 x 64
   *    matrix-pipe cycles per 16 channels.  Not the default and not what bench.py's `value` is measured on. */
  int bf16_mfma;
} dvsr_edvr_config;
typedef struct dvsr_edvr_plan dvsr_edvr_plan;

int dvsr_edvr_plan_create(const dvsr_edvr_config* cfg, int B, int H, int W, dvsr_edvr_plan** out);
/* The same plan with PER-GROUP parameter gradients: the batch is `grad_groups` groups of B / grad_groups consecutive
 * clips, and dvsr_edvr_backward writes one gradient per group -- every grad_params[i] then points to
 * [grad_groups][numel(param i)] floats, group-major.  This is how the frames of a video that all adapt from the SAME
 * weights (test_dynavsr.py:208-277 deep-copies the un-adapted networks for every frame; with adapt_iter = 1, the value of
 * every shipped YAML, the K forward + backward passes differ only in their data) run as ONE batch of K clips whose
 * launches fill the device, while each frame still gets the gradient its own sequential step would have seen; the meta
 * tasks of train_dynavsr.py:355-426 likewise.  Forward, activations and data gradients are per sample in any case;
 * grad_groups = 1 is dvsr_edvr_plan_create.  The workspace grows by the extra weight-gradient partial sums. */
int dvsr_edvr_plan_create_grouped(const dvsr_edvr_config* cfg, int B, int H, int W, int grad_groups,
                                  dvsr_edvr_plan** out);
/* ... and PER-GROUP WEIGHTS (weight_sets == grad_groups; 1 = dvsr_edvr_plan_create_grouped): every params[i] then points to
 * [weight_sets][numel(param i)] floats and the clips of batch group g are convolved with set g, in forward, data gradient
 * and (per group, as above) weight gradient.  This is the K private copies of test_dynavsr.py:208 AFTER they have diverged:
 * the second and later inner steps (adapt_iter > 1: BASELINE configs[2] takes 3) and the adapted forwards of K frames
 * (:279-283) still run as one batch.  Every launch indexes its packed weights and bias by n / (N / weight_sets). */
int dvsr_edvr_plan_create_ex(const dvsr_edvr_config* cfg, int B, int H, int W, int grad_groups, int weight_sets,
                             dvsr_edvr_plan** out);
void dvsr_edvr_plan_destroy(dvsr_edvr_plan* plan);
int dvsr_edvr_num_params(const dvsr_edvr_plan* plan);
int dvsr_edvr_num_launches(const dvsr_edvr_plan* plan);
size_t dvsr_edvr_workspace_bytes(const dvsr_edvr_plan* plan, int need_grad);
int dvsr_edvr_forward(const dvsr_edvr_plan* plan, const float* const* params, const float* x,
                      float* out, void* workspace, size_t workspace_bytes, dvsr_stream_t stream);
/* dvsr_edvr_forward without its weight-packing launches, for a FROZEN network run over many clips (Video_base_model.py:197-201
 * `test()` called per clip of a video, test_dynavsr.py:200-204): `workspace` must be the workspace of an earlier
 * dvsr_edvr_forward of THIS plan with the SAME parameter values and the same need_grad sizing, not written by anyone since
 * (the packed weights live in it and are a pure function of the parameters).  Same launches otherwise, bit-identical
 * results.  After the parameters change, call dvsr_edvr_forward once again.  The caller owns that contract: stale packs are
 * not detected here (dynavsr_amd.engine keys them on the parameters' storage and version counters). */
int dvsr_edvr_forward_packed(const dvsr_edvr_plan* plan, const float* const* params, const float* x,
                             float* out, void* workspace, size_t workspace_bytes, dvsr_stream_t stream);
/* Backward of dvsr_edvr_forward (first-order; replaces autograd through EDVR incl. the 20 calls of
 * modulated_deform_conv_cuda_backward per clip, deform_conv_cuda.cpp:566-679).  Must follow a forward
 * on the SAME workspace, which must have been sized with need_grad=1.  grad_out [B,3,sH,sW];
 * grad_params: HOST array of device pointers shaped like `params`, every entry is OVERWRITTEN;
 * grad_x [B,nframes,3,H,W] or NULL.  The DCN input gradient uses fp32 atomics (summation order
 * varies run to run, as in the reference kernel, deform_conv_cuda_kernel.cu:687). */
int dvsr_edvr_backward(const dvsr_edvr_plan* plan, const float* const* params, const float* x,
                       const float* grad_out, float* const* grad_params, float* grad_x, void* workspace,
                       size_t workspace_bytes, dvsr_stream_t stream);
int dvsr_edvr_num_backward_launches(const dvsr_edvr_plan* plan);
/* Measurement aids: per-launch description (kind = "conv3x3s1", "mdcn", ...; algorithmic FLOPs and
 * bytes of that launch) and a forward that brackets every launch with hipEvents on `stream`,
 * synchronises, and returns per-launch milliseconds in op_ms[dvsr_edvr_num_launches()]. */
int dvsr_edvr_op_info(const dvsr_edvr_plan* plan, int index, char* kind, int kind_cap, char* name,
                      int name_cap, double* flops, double* bytes);
/* Contraction work of the plan's two tapes: out9 (NINE doubles) = {forward algorithmic FLOPs, forward fp32 products as the
 * kernels shape them, backward algorithmic, backward shaped, algorithmic bytes of the forward tape (every launch's distinct
 * inputs once + its outputs once), forward FLOPs issued to the fp32 matrix pipe, forward FLOPs issued to the bf16 matrix pipe,
 * backward fp32-pipe, backward bf16-pipe}.  Algorithmic = 2 x MACs of the direct sums; launches on the Winograd
 * F(2x2,3x3) kernels do 16/36 of theirs; launches on the exact 3-way bf16 operand split issue six bf16 products per fp32
 * product.  bench.py's roofline fractions price the issued figures against the peak of the pipe they were issued to. */
int dvsr_edvr_plan_work(const dvsr_edvr_plan* plan, double* out9);
/* The same nine figures with the FORWARD tape priced as dvsr_edvr_forward runs it on a workspace WITHOUT the gradient region
 * (workspace_bytes < dvsr_edvr_workspace_bytes(plan, 1): test(), Video_base_model.py:197-201): such a forward may run the
 * Winograd F(4x4,3x3) kernel (36/144 of the direct sum's multiplies, six bf16 products each) on layers whose training tape
 * keeps F(2x2,3x3); dvsr_edvr_op_info's tags ("w5") and dvsr_edvr_forward_timed describe that forward. */
int dvsr_edvr_plan_work_nograd(const dvsr_edvr_plan* plan, double* out9);
/* Does the plans' weight-gradient side stream run BESIDE `stream` on the current device?  ROCm maps HIP streams onto
 * GPU_MAX_HW_QUEUES hardware queues and two streams on one queue serialise -- which queue a stream gets depends on every
 * stream the process created before (an initialised RCCL communicator holds some; train_dynavsr.py:23-30 creates it first).
 * Measured once per (device, stream) with a 150 us two-kernel probe and cached: 1 overlaps, 0 serialised (dvsr_edvr_backward /
 * dvsr_estimator_backward then launch their weight gradients on `stream` itself instead of forking), -1 unknown (the stream is
 * being captured, or DVSR_BWD_PROBE=0).  Synchronises `stream`. */
int dvsr_side_stream_overlaps(dvsr_stream_t stream);
/* Test aid: where launch `index` of the forward tape leaves its result (which = 0; 1 = the second output of the pool /
 * TSA-gate launches).  *in_arena = 1: offset_floats counts from the start of the workspace (every activation of a
 * need_grad forward stays there); 0: the launch writes the output tensor. */
int dvsr_edvr_op_output(const dvsr_edvr_plan* plan, int index, int which, int* in_arena, long long* offset_floats,
                        long long* numel);
int dvsr_edvr_forward_timed(const dvsr_edvr_plan* plan, const float* const* params, const float* x,
                            float* out, void* workspace, size_t workspace_bytes,
                            dvsr_stream_t stream, float* op_ms);
/* Measurement aid: register-only fp32 MFMA loop (256-thread workgroups, `nacc` accumulators per wave,
 * `lds_bytes` of dynamic LDS to reproduce an occupancy); returns MFMA instructions per wave or -1. */
long long dvsr_debug_mfma_peak(float* out, int blocks, int iters, int nacc, int lds_bytes, dvsr_stream_t stream);
/* Measurement aid (tools/mfma_shadow.py): shader cycles of a stream of fp32 MFMAs with nv v_fma_f32 (kind 0) or
 * ds_read_b32 (kind 1) behind each one, accumulators in VGPRs (acc 0) or AGPRs (acc 1). */
int dvsr_debug_mfma_shadow(long long* cycles, float* out, int blocks, int iters, int nv, int kind, int acc,
                           dvsr_stream_t stream);
/* Offset (in floats, into the workspace) and size of a named intermediate, for layer-by-layer
 * parity checks ("L1_fea", "aligned", "tsa_out", "recon", ...). */
int dvsr_edvr_tensor_info(const dvsr_edvr_plan* plan, const char* name, long long* offset_floats,
                          long long* numel);

/* ---- Down-scaling estimators MFDN / SFDN as one launch tape ------------------------------------
 * Replaces DirectKernelEstimatorVideo.forward (models/archs/LRimg_estimator.py:92-117, "MFDN":
 * Conv3d(k3)+ReplicationPad3d, ReflectionPad2d + 3x3 / 4x4-stride-2 Conv2d, Conv3d, 1x1, per-frame
 * mean subtraction / re-addition) and DirectKernelEstimator_CMS.forward (:55-67, "SFDN", x2) plus their
 * autograd backward w.r.t. the parameters -- the estimator sits inside the inner-step graph
 * (test_dynavsr.py:237-241) but its input clip is data, so no input gradient is produced.
 * x: [B][in_nc][T][H][W] fp32 for MFDN (the layout feed_data builds, LRestimator_model.py:103),
 * [B][in_nc][H][W] for SFDN; out: [B][in_nc][T][H/scale][W/scale] (SFDN: [B][in_nc][H/2][W/2]).
 * params / grad_params: host arrays of dvsr_estimator_num_params() (= 14) device pointers in
 * state-dict order conv0.weight, conv0.bias, ..., conv6.bias; gradients are overwritten.
 * Workspace protocol as for the EDVR plan (need_grad=1 before the forward whose backward follows). */
#define DVSR_ESTIMATOR_MFDN 0
#define DVSR_ESTIMATOR_SFDN 1
typedef struct dvsr_estimator_config {
  int kind;    /* DVSR_ESTIMATOR_* */
  int nf;      /* 64 in every shipped YAML */
  int in_nc;   /* 3 */
  int scale;   /* MFDN: 2 or 4; SFDN: 2 */
  int nframes; /* T (MFDN); ignored for SFDN */
} dvsr_estimator_config;
typedef struct dvsr_estimator_plan dvsr_estimator_plan;
int dvsr_estimator_plan_create(const dvsr_estimator_config* cfg, int B, int H, int W, dvsr_estimator_plan** out);
/* Per-group parameter gradients, as dvsr_edvr_plan_create_grouped: grad_params[i] = [grad_groups][numel(param i)]. */
int dvsr_estimator_plan_create_grouped(const dvsr_estimator_config* cfg, int B, int H, int W, int grad_groups,
                                       dvsr_estimator_plan** out);
/* ... with per-group weights, as dvsr_edvr_plan_create_ex. */
int dvsr_estimator_plan_create_ex(const dvsr_estimator_config* cfg, int B, int H, int W, int grad_groups, int weight_sets,
                                  dvsr_estimator_plan** out);
void dvsr_estimator_plan_destroy(dvsr_estimator_plan* plan);
int dvsr_estimator_num_params(const dvsr_estimator_plan* plan);
/* tape length: forward ops (backward = 0) or backward ops (backward = 1); a measurement aid like
 * dvsr_edvr_num_launches / dvsr_edvr_num_backward_launches */
int dvsr_estimator_num_launches(const dvsr_estimator_plan* plan, int backward);
int dvsr_estimator_plan_work(const dvsr_estimator_plan* plan, double* out9);   /* as dvsr_edvr_plan_work */
size_t dvsr_estimator_workspace_bytes(const dvsr_estimator_plan* plan, int need_grad);
int dvsr_estimator_forward(const dvsr_estimator_plan* plan, const float* const* params, const float* x, float* out,
                           void* workspace, size_t workspace_bytes, dvsr_stream_t stream);
int dvsr_estimator_backward(const dvsr_estimator_plan* plan, const float* const* params, const float* x,
                            const float* grad_out, float* const* grad_params, void* workspace,
                            size_t workspace_bytes, dvsr_stream_t stream);

/* ---- Charbonnier loss (models/loss.py:19-30, the `pixel_criterion: cb` of every EDVR YAML) --------
 * loss[0] = mean(sqrt((x-y)^2 + eps)), deterministic two-stage reduction; backward writes
 * gx = grad_loss[0]/n * (x-y)/sqrt((x-y)^2+eps) (the gradient w.r.t. y is -gx). */
size_t dvsr_charbonnier_workspace_bytes(void);
int dvsr_charbonnier_forward(const float* x, const float* y, float* loss, long long n, float eps,
                             void* workspace, size_t workspace_bytes, dvsr_stream_t stream);
int dvsr_charbonnier_backward(const float* x, const float* y, const float* grad_loss, float* gx, long long n,
                              float eps, dvsr_stream_t stream);
/* Per-group losses of a batch: x, y = [groups][n]; loss[g], grad_loss[g] per group; every group is reduced exactly as a
 * scalar call on its n elements would be (same partial sums, same order -> bit-identical values).  The K frames adapted
 * as one batch (dvsr_edvr_plan_create_grouped) each have their own `cri_pix(netG(SLR), LR_center)`, models/loss.py:26-30.
 * Workspace: groups * dvsr_charbonnier_workspace_bytes(). */
int dvsr_charbonnier_forward_grouped(const float* x, const float* y, float* loss, long long n, int groups, float eps,
                                     void* workspace, size_t workspace_bytes, dvsr_stream_t stream);
int dvsr_charbonnier_backward_grouped(const float* x, const float* y, const float* grad_loss, float* gx, long long n,
                                      int groups, float eps, dvsr_stream_t stream);

/* ---- loss tail of the inner MAML step (test_dynavsr.py:264-274) ------------------------------------------
 * loss = cri_pix(netG(SLR), LR_center) + 10 * F.l1_loss(SLR, SLR_fixed): loss[0] = (base ? base[0] : 0) +
 * weight * mean|x - y| with `base` the pixel loss already on the device (one reduction + one 1-thread kernel instead
 * of the abs / mean / mul / add chain); backward writes gx = grad_loss[0] * weight / n * sign(x - y) (sign(0) = 0,
 * as torch's l1_loss); the gradient w.r.t. base is grad_loss itself.  Workspace: dvsr_charbonnier_workspace_bytes(). */
int dvsr_l1_tail_forward(const float* x, const float* y, const float* base, float weight, float* loss, long long n,
                         void* workspace, size_t workspace_bytes, dvsr_stream_t stream);
int dvsr_l1_tail_backward(const float* x, const float* y, const float* grad_loss, float weight, float* gx, long long n,
                          dvsr_stream_t stream);
/* Per-group form, as dvsr_charbonnier_*_grouped: x, y = [groups][n]; base, loss, grad_loss = [groups]. */
int dvsr_l1_tail_forward_grouped(const float* x, const float* y, const float* base, float weight, float* loss, long long n,
                                 int groups, void* workspace, size_t workspace_bytes, dvsr_stream_t stream);
int dvsr_l1_tail_backward_grouped(const float* x, const float* y, const float* grad_loss, float weight, float* gx,
                                  long long n, int groups, dvsr_stream_t stream);

/* ---- op-level entries to the pipelined / small-grid conv kernels ---------------------------------------------------
 * dvsr_conv2d_forward needs no workspace and runs the un-packed kernel.  These pack the weights into a caller
 * workspace on the stream and run the geometry the whole-network plan would pick for the shape (pipelined 4-row tiles,
 * or the K-split kernel on small grids): 1x1 and 3x3, stride 1, pad ks/2; same descriptor and epilogues.  The data
 * gradient is the same kernel over the transposed, tap-mirrored pack (single plain input). */
size_t dvsr_conv2d_packed_workspace_bytes(const dvsr_conv2d_desc* d);
int dvsr_conv2d_forward_packed(const dvsr_conv2d_desc* d, void* workspace, size_t workspace_bytes, dvsr_stream_t stream);
int dvsr_conv2d_dgrad_packed(const dvsr_conv2d_desc* d, const float* gy, float* gx0, void* workspace,
                             size_t workspace_bytes, dvsr_stream_t stream);
/* Which kernel dvsr_conv2d_forward_packed runs for d (tests and tools): geo[0..3] = input channels per chunk,
 * tile rows (x 32 pixels), 32-output-channel halves per workgroup, 1 when the halo is staged by LDS-DMA
 * (conv2d_dma_kernel: 3x3 / stride 1, 16-byte aligned inputs, W % 4 == 0, channel counts % 8 == 0), 2 for the row-split
 * DMA kernel of 7x7 / 9x9 convolutions (conv2d_dmarow_kernel) -- for those sizes the packed entries exist ONLY when
 * geo[3] == 2, otherwise use dvsr_conv2d_forward / _backward.  32-channel chunks = the K-split small-grid kernel.
 * geo[3] == 3: the Winograd F(2x2, 3x3) kernel (conv2d_wino_kernel; the DMA-halo kernel's conditions and at least 16 input
 * channels; geo[1] = 4: 4x64-pixel workgroup tiles, 8: 8x32), chosen per shape by a cost model; the environment variable
 * DVSR_CONV_WINO=0 keeps the direct kernels, =2 takes it wherever it is eligible.  Same results within fp32 round-off. */
int dvsr_conv2d_packed_geometry(const dvsr_conv2d_desc* d, int geo[4]);

/* Weight / bias gradient of a single-input 3x3 stride-1 convolution with both operands rounded to bf16 on
 * v_mfma_f32_32x32x16_bf16 (fp32 accumulation and flush): what the EDVR plan runs for its weight gradients when
 * network_G.bf16_mfma = 1 (BASELINE configs[4]).  gy = gradient w.r.t. the pre-activation output; gb may be NULL.
 * Workspace: dvsr_conv2d_backward_workspace_bytes(d). */
int dvsr_conv2d_wgrad_bf16(const dvsr_conv2d_desc* d, const float* gy, float* gw, float* gb, void* workspace,
                           size_t workspace_bytes, dvsr_stream_t stream);
/* The same weight gradient at fp32 accuracy on the bf16 pipe: both operands split exactly into three bf16 pieces, six
 * partial products per fp32 product (what the plans run for 3x3 stride-1 layers; arguments as dvsr_conv2d_wgrad_bf16). */
int dvsr_conv2d_wgrad_split3(const dvsr_conv2d_desc* d, const float* gy, float* gw, float* gb, void* workspace,
                             size_t workspace_bytes, dvsr_stream_t stream);

/* ---- TOFlow backbone ops (codes/models/archs/TOF_arch.py:25-140, arch_util.py:55-79) --------
 * The convolutions of SpyNet (7x7) and of the TOFlow head (9x9, 1x1) go through dvsr_conv2d_forward / _backward
 * (ks = 7, 9, 1); the ops below are the rest of the graph.  All fp32 NCHW, HBM-bound streaming kernels.
 *
 * flow_warp (arch_util.py:55-79): out[n,c,y,x] = bilinear sample of x[n,c] at (x + flow[n,0,y,x], y + flow[n,1,y,x])
 * through F.grid_sample's align_corners=False mapping of the (size-1)-normalised grid, zeros padding.  `flow` is
 * channel-first [N,2,H,W] (the reference permutes to NHW2 first).  out_bstride / gout_bstride (0 = dense) let the
 * output be a channel slice of a wider tensor (torch.cat([ref, warped, flow]) is built in place).  Backward:
 * grad_x (zeroed here, atomics) and / or grad_flow; either may be NULL. */
int dvsr_flow_warp_forward(const float* x, const float* flow, float* out, int N, int C, int H, int W,
                           long long out_bstride, dvsr_stream_t stream);
int dvsr_flow_warp_backward(const float* x, const float* flow, const float* grad_out, float* grad_x, float* grad_flow,
                            int N, int C, int H, int W, long long gout_bstride, dvsr_stream_t stream);
/* F.avg_pool2d(x, kernel_size=2, stride=2) (TOF_arch.py:67-74); planes = N*C; output (H/2) x (W/2). */
int dvsr_avgpool2_forward(const float* x, float* y, long long planes, int H, int W, dvsr_stream_t stream);
int dvsr_avgpool2_backward(const float* grad_y, float* grad_x, long long planes, int H, int W, int accumulate,
                           dvsr_stream_t stream);
/* F.interpolate(x, size=(Ho,Wo), mode='bilinear', align_corners=True) * mul (TOF_arch.py:84-85); y_plane_stride
 * (0 = Ho*Wo) addresses a channel slice.  Backward zeroes grad_x and scatters. */
int dvsr_resize_bilinear_ac_forward(const float* x, float* y, long long planes, int H, int W, int Ho, int Wo, float mul,
                                    long long y_plane_stride, dvsr_stream_t stream);
int dvsr_resize_bilinear_ac_backward(const float* grad_y, float* grad_x, long long planes, int H, int W, int Ho, int Wo,
                                     float mul, long long gy_plane_stride, dvsr_stream_t stream);
/* F.interpolate(x, scale_factor=scale, mode='bicubic', align_corners=True): the drivers bring the (S)LR clip to the
 * output size before TOFlow (test_dynavsr.py:188-193, 245-250; train_dynavsr.py:314-320).  ATen's cubic convolution
 * (A = -0.75), clamped taps.  Backward zeroes grad_x and scatters. */
int dvsr_upsample_bicubic_ac_forward(const float* x, float* y, long long planes, int H, int W, int scale,
                                     dvsr_stream_t stream);
int dvsr_upsample_bicubic_ac_backward(const float* grad_y, float* grad_x, long long planes, int H, int W, int scale,
                                      dvsr_stream_t stream);
/* out[n,c,:] (+)= x[n,c,:] * scale[c] + shift[c]; scale / shift may be NULL (1 / 0): normalize / denormalize
 * (TOF_arch.py:13-22) and channel-slice copies; batch strides 0 = dense. */
int dvsr_channel_affine(const float* x, const float* scale, const float* shift, float* out, int N, int C, long long HW,
                        long long x_bstride, long long out_bstride, int accumulate, dvsr_stream_t stream);
/* nn.BatchNorm2d (+ the ReLU behind it, TOF_arch.py:33-42).  training != 0: batch statistics, running estimates
 * updated in place (momentum, unbiased variance); else the running estimates.  save_mean / save_rstd [C] feed the
 * backward, which takes the gradient w.r.t. the (post-ReLU) output and needs y only when relu. */
size_t dvsr_batchnorm_workspace_bytes(int C);
int dvsr_batchnorm_forward(const float* x, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* y, float* save_mean, float* save_rstd, int N, int C, long long HW,
                           int training, float momentum, float eps, int relu, void* workspace, size_t workspace_bytes,
                           dvsr_stream_t stream);
int dvsr_batchnorm_backward(const float* x, const float* grad_y, const float* y, const float* gamma,
                            const float* save_mean, const float* save_rstd, float* grad_x, float* grad_gamma,
                            float* grad_beta, int N, int C, long long HW, int training, int relu, void* workspace,
                            size_t workspace_bytes, dvsr_stream_t stream);

/* ---- DUF backbone ops (codes/models/archs/DUF_arch.py:32-180) ------------------------------------
 * Frames are the batch axis ([B*T][C][H][W]).  Conv3d (1,3,3) / (1,1,1) are dvsr_conv2d_* over the frames; the (3,3,3)
 * Conv3d of the dense blocks (:45-58) is a 3x3 conv2d over 3C channels after this gather of frames t-1, t, t+1:
 * y[(b*To + t)][c*3 + kt] = x[(b*T + t + kt - pad_t)][c] or 0; pad_t = 1 keeps T (padding (1,1,1)), pad_t = 0 is the
 * T-reducing block (padding (0,1,1), To = T - 2).  The Conv3d weight [Cout][C][3][3][3] is that conv's
 * [Cout][3C][3][3] weight as it lies in memory.  BatchNorm3d = dvsr_batchnorm_* over N = B*T. */
int dvsr_temporal_gather3_forward(const float* x, float* y, int B, int T, int C, long long HW, int pad_t,
                                  dvsr_stream_t stream);
int dvsr_temporal_gather3_backward(const float* grad_y, float* grad_x, int B, int T, int C, long long HW, int pad_t,
                                   dvsr_stream_t stream);
/* Tail of DUF.forward (:160-176) in one kernel: Fx = softmax over the 25 taps of filter_logits [B][25*R][H][W]
 * (channel f*R + r, R = scale^2), DynamicUpsamplingFilter_3C (:86-110) of the centre frame x_center [B][3][H][W] (5x5
 * patch, zero padded), + residual [B][3R][H][W] (channel c*R + r, or 3r + c when adapt_official reorders it, :17-29),
 * F.pixel_shuffle(scale) -> out [B][3][scale*H][scale*W].  Backward: grad_logits and grad_residual are written in
 * full; grad_x_center may be NULL (the clip is data in the inner loop).  adapt_official is a bit set: bit 0 = the
 * residual's channel order above; bit 1 = `filter_logits` already holds the 25 filter TAPS, applied as given with no
 * softmax (DynamicUpsamplingFilter_3C used as a module on its own, :100-110; grad_logits is then d/d(taps)). */
int dvsr_dynamic_filter_forward(const float* x_center, const float* filter_logits, const float* residual, float* out,
                                int B, int H, int W, int scale, int adapt_official, dvsr_stream_t stream);
int dvsr_dynamic_filter_backward(const float* x_center, const float* filter_logits, const float* grad_out,
                                 float* grad_logits, float* grad_residual, float* grad_x_center, int B, int H, int W,
                                 int scale, int adapt_official, dvsr_stream_t stream);

/* ---- 3x3 convolution with very few outputs (EDVR's conv_last: 64 -> 3 at the HR size) -------------------------
 * y = act(conv3x3(x, w) + bias) [+ res], stride 1, zero padding 1, Cout <= 4; x [N][C][H][W], w [Cout][C][3][3], res / y
 * [N][Cout][H][W] (res may be NULL: EDVR_arch.py:311-312 adds the bilinear base frame here).  C = 64, Cout <= 3, W % 4 == 0
 * and 16-byte aligned tensors run as a 9 Cout-row GEMM on the fp32 matrix pipe + a shift-add; everything else on the
 * vector ALU. */
int dvsr_conv3x3_small_cout(const float* x, const float* w, const float* bias, const float* res, float* y, int N, int C, int H,
                            int W, int Cout, int act, dvsr_stream_t stream);

/* ---- two 1x1 convolutions over one input (TSA fusion) ---------------------------------------------------
 * EDVR_arch.py:183-202: fea = lrelu(fea_fusion(x)), att = lrelu(sAtt_1(x)) read the same [N][Cin][H][W] tensor (Cin = nframes *
 * nf); one pass over it produces both [N][64][H][W] outputs.  w0 / w1: [64][Cin] (the modules' [64][Cin][1][1] weights), b0 /
 * b1: [64] or NULL, act as in dvsr_conv2d_desc.  Needs Cin % 16 == 0, (H * W) % 4 == 0 and 16-byte aligned tensors; anything
 * else returns DVSR_ERR_UNSUPPORTED (the plans fall back to two dvsr_conv2d launches). */
int dvsr_conv1x1_dual(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, float* y0, float* y1,
                      int N, int Cin, int H, int W, int act, dvsr_stream_t stream);

/* ---- random patch crops of the inner step (train.maml.use_patch) -----------------------------------------
 * test_dynavsr.py:118-145 (and train_dynavsr.py:208-243) crop `num_patch` random patches out of the SLR clip and its
 * target with preprocessing.common_crop (data/meta_learner/preprocessing.py:57-85) and stack them into a batch.
 * dst[p][n][y][x] = src[n][scale*py[p] + y][scale*px[p] + x] for n < planes, 0 <= y, x < scale*edge; py / px are HOST
 * arrays of P <= 64 positions on the lowest-resolution grid (the host draws them, like the reference).  Backward
 * zeroes grad_src and adds the (overlapping) patches back with atomics. */
int dvsr_patch_gather_forward(const float* src, float* dst, const int* py, const int* px, int P, int planes, int H, int W,
                              int edge, int scale, dvsr_stream_t stream);
int dvsr_patch_gather_backward(const float* grad_dst, float* grad_src, const int* py, const int* px, int P, int planes,
                               int H, int W, int edge, int scale, dvsr_stream_t stream);

/* ---- inner-loop optimiser steps over lists of parameter tensors ------------------------------------
 * test_dynavsr.py:223-231 steps torch.optim.Adam(lr_alpha, betas) / torch.optim.SGD(lr_alpha) over the
 * ~158 tensors of netG + netE once per inner iteration; these entry points do one such step (same
 * single-tensor formulas, no amsgrad / momentum) with one launch per 48 tensors.  All arrays are HOST
 * arrays of length n_tensors; a NULL gradient or numel <= 0 skips that tensor (torch.optim skips
 * parameters whose .grad is None).  `step` is the 1-based step count of the Adam bias correction. */
int dvsr_adam_step(float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const long long* numel, int n_tensors, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, dvsr_stream_t stream);
int dvsr_sgd_step(float* const* params, const float* const* grads, const long long* numel, int n_tensors,
                  float lr, float weight_decay, dvsr_stream_t stream);
/* The per-frame deep copies of the networks (test_dynavsr.py:208 `modelcp.netG = deepcopy(model.netG)`) for a batch of
 * frames: dst[i] = [copies][numel[i]] floats, every copy = src[i].  HOST arrays of device pointers, one launch per 48
 * tensors (the framework's foreach copy with broadcast sources is one launch per tensor: 158 for netG + netE). */
int dvsr_replicate_tensors(const float* const* src, float* const* dst, const long long* numel, int n_tensors, int copies,
                           dvsr_stream_t stream);

/* ---- per-frame image metrics on the device ------------------------------------------
 * Replaces, per super-resolved frame of test_dynavsr.py:285-292 (and of the validation loops of
 * train_dynavsr.py), the device->host copy of the fp32 frame plus
 *   util.tensor2img(rlt, mode='rgb')      codes/utils/util.py:112-142   clamp to [lo,hi], rescale, x255,
 *                                                                       round half to even, uint8 HWC
 *   util.calculate_psnr(img, hr_image)    codes/utils/util.py:262-269   20 log10(255 / sqrt(mse)) over uint8
 *   util.calculate_ssim(img, hr_image)    codes/utils/util.py:271-313   11x11 Gaussian (sigma 1.5) window in
 *                                                                       float64, "valid" region, mean over
 *                                                                       pixels and channels
 * sr, gt: [C,H,W] fp32 device tensors in the network's value range (both are quantised the tensor2img way:
 * the reference's hr_image is tensor2img(GT)).  out (DEVICE, 2 doubles): out[0] = mean squared difference of
 * the two uint8 frames (exact integer sum / (C*H*W); the caller forms the PSNR, inf when 0), out[1] = mean
 * SSIM (NaN when H or W <= 10: the reference's mean of an empty map).  sr_hwc_u8 (nullable): the uint8
 * [H,W,C] image of sr for the PNG writer.  Deterministic (integer atomics + fixed-order sums). */
size_t dvsr_frame_metrics_workspace_bytes(int C, int H, int W);
int dvsr_frame_metrics(const float* sr, const float* gt, int C, int H, int W, float lo, float hi,
                       unsigned char* sr_hwc_u8, double* out, void* workspace, size_t workspace_bytes,
                       dvsr_stream_t stream);

/* ---- synthetic degradation on the device -------------------------------------------
 * Replaces the tensor part of Degradation.apply (codes/data/random_kernel_generator.py:84-130):
 *   img = ReflectionPad2d(K // 2)(img);  lr = conv2d(img, kernel.repeat(3,1,1,1), groups=3, stride=scale)
 * for a clip img [N,C,H,W] (N frames) -> out [N,C,Ho,Wo], Ho = (H + 2 (K//2) - K) / scale + 1.
 * kernels: [n_kernels,K,K] fp32 on the device, already centre-of-mass shifted (kernel_shift :51-76 stays on
 * the host: a 27 x 27 spline shift).  n_kernels = 1: one kernel for every frame (:91-103); otherwise frame i
 * uses kernel (i + kernel_offset) mod n_kernels (:104-122: offset 0 for N = n_kernels, -1 for N = n_kernels + 2).
 * quantise != 0 fuses vsrbase.py:185 `mul(255).clamp(0, 255).round().div(255)` into the store. */
int dvsr_degrade_apply(const float* img, const float* kernels, float* out, int N, int C, int H, int W, int K,
                       int scale, int n_kernels, int kernel_offset, int quantise, dvsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNAVSR_HIP_H_ */
