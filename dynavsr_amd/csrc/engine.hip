// EDVR execution plan: the whole backbone (EDVR_arch.py:254-313) as a static tape of kernel
// launches over one caller-provided workspace arena.  Replaces the reference's Python module
// graph + autograd for this network: one C call enqueues the ~190 forward launches (and one more
// the backward), so small clips (the 16x-smaller super-LR clip of the inner MAML step) are not
// bound by host-side framework overhead, and the stream can be captured into a hipGraph.
//
// Design notes
//  * The N frames of a clip are batched through feature extraction AND through PCD alignment
//    (the reference loops frames in Python, EDVR_arch.py:291-296); the reference frame's features
//    enter each two-input conv as a broadcast second pointer (x1_bdiv), so no clone/cat/stack
//    tensor is ever materialised.
//  * Every activation gets its own slot in the arena (bump allocation); the arena doubles as the
//    saved-activation store for backward.  288 GB of HBM make the ~3.3 GB of a 180x320 clip a
//    non-issue and remove all allocator traffic from the hot loop.
//  * Parameters are addressed by their position in the reference's state-dict order
//    (dynavsr_amd/spec.py mirrors the walk below).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

enum { SP_ARENA = 0, SP_INPUT = 1, SP_OUTPUT = 2, SP_NONE = 3 };
struct T {
  int space = SP_NONE;
  size_t off = 0, numel = 0;
  bool valid() const { return space != SP_NONE; }
};

enum OpType { OP_CONV, OP_DCN, OP_UP, OP_POOL, OP_GATE, OP_BLEND, OP_ADD, OP_PAD, OP_MEANSUB, OP_ADDMEAN };

struct Op {
  OpType type;
  const char* name = "";
  T x0, x1, res, y, y2;  // generic tensor slots (meaning depends on type)
  int pw = -1, pb = -1;  // parameter indices (weight, bias)
  // conv / dcn geometry
  int N = 0, c0 = 0, c1 = 0, H = 0, W = 0, Cout = 0, ks = 3, stride = 1, act = 0, ps = 0;
  int x1_bdiv = 1, dg = 8;
  int pad = -1;        // conv: < 0 = ks / 2 (EDVR); the estimator convolves explicitly padded tensors with 0
  int pmode = 0, T = 1;  // OP_PAD: PAD_* mode; frames per clip (OP_PAD / OP_MEANSUB / OP_ADDMEAN)
  // wmap: the parameter is a [Cout][c0/4][4][4] stride-2 kernel, run as 2x2 over the space-to-depth
  // input; its re-laid-out copy (and the gradient of that copy) live at w2_off of the two arenas
  int wmap = 0;
  size_t w2_off = 0;
  int no_dgrad = 0;  // the input does not need a gradient (it derives from the network input only)
  // OP_CONV: the output is stored straight into the explicitly padded tensor of the next conv (PS_PAD_REFLECT*, common.h)
  // = the y of the OP_PAD that follows; the dense y slot is never written (its GRADIENT slot is: backward is unchanged).
  // OP_PAD: fused_into >= 0: no forward launch, the producer (that op index) wrote the padded tensor.
  int pad_out = 0;
  struct T ypad;
  int fused_into = -1;
  // OP_CONV: dual >= 0: this 1x1 conv and op `dual` (same input, 64 outputs each) run as ONE launch (conv1x1_dual.hip) when
  // the tensors allow it; the partner carries fused_into = this op and launches nothing then.  The backward tape is unchanged.
  int dual = -1;
  long long x0_bs = 0, x1_bs = 0;
  // streaming ops
  int S = 1;
  float mul = 1.f;
  size_t planes = 0;
  int gB = 0, gN = 0, gC = 0;
  size_t gHW = 0;
  // packed-weight slots (conv2d_v2.hip): forward pack in the activation arena, the two
  // data-gradient packs in the backward-only region
  size_t wp_off = 0, dpk_off[2] = {0, 0};
  size_t wp_floats = 0, dpk_floats[2] = {0, 0};
  ConvGeo geo = {8, 8, 2}, dgeo[2] = {{8, 8, 2}, {8, 8, 2}};  // launch geometry chosen at plan time
  // NO-GRAD forwards (a workspace without the gradient region: test(), the baseline / adapted forwards of the per-frame
  // pipeline -- Video_base_model.py:197-201) may take another kernel for the same layer: the F(4x4, 3x3) Winograd kernel
  // (conv2d_wino5.hip).  Training tapes keep `geo`, so that everything the backward re-reads, the batched == per-frame
  // identities and the goldens of the inner step are what they were.  ng_off == wp_off: same geometry, same pack.
  ConvGeo geo_ng = {8, 8, 2};
  size_t wp_ng_off = 0, wp_ng_floats = 0;
};

// ---- backward tape -------------------------------------------------------------------------
// Pointer spaces of the backward pass.
enum { R_ACT = 0, R_GRAD = 1, R_X = 2, R_GX = 3, R_GOUT = 4, R_TMP = 5, R_NONE = 6 };
struct Ref {
  int space = R_NONE;
  size_t off = 0;
};
enum BType { B_MEMSET, B_COPYADD, B_ACT, B_WGRAD, B_DGRAD, B_REDUCE, B_DCN, B_UP, B_POOL, B_GATE, B_BLEND,
             B_PADFOLD, B_ADDMEAN, B_WUNMAP };
struct BOp {
  BType type;
  int fwd = -1;          // index of the forward op this belongs to
  Ref a, b, c, d, e, f;  // meaning depends on type
  size_t n = 0;          // element count for streaming ops
  int accum = 0, which = 0, cnt = 1, B = 0;
  int mask_op = -1;  // B_DGRAD / B_PADFOLD: forward op whose activation backward is fused into this launch
  size_t ws_off = 0, ws_bytes = 0;  // B_WGRAD: this launch's private slot region inside the wgrad scratch
  long long bs = 0;
  size_t per = 0;
};

static inline int conv_pad(const Op& o) { return o.pad < 0 ? o.ks / 2 : o.pad; }
static inline int conv_out(const Op& o, int in) { return (in + 2 * conv_pad(o) - o.ks) / o.stride + 1; }

}  // namespace dvsr

struct dvsr_edvr_plan {
  dvsr_edvr_config cfg;
  int B, H, W;
  int n_params = 0;
  size_t arena_floats = 0;
  std::vector<dvsr::Op> ops;
  std::vector<std::pair<std::string, dvsr::T>> named;
  std::vector<std::pair<size_t, size_t>> allocs;  // (offset, numel) of every arena slot, ascending
  std::vector<dvsr::BOp> bops;                    // backward tape (already in execution order)
  size_t dpack_floats = 0;                        // packed transposed weights for the dgrad launches
  bool use_v1 = false;                            // DVSR_CONV_V1=1: un-pipelined conv kernel (A/B aid)
  size_t tmp_floats = 0;                          // dense dgrad staging for broadcast/strided views
  size_t scratch_bytes = 0;                       // wgrad partials / DCN column buffer
  size_t wscratch_bytes = 0;                      // private wgrad scratch of the side stream
  // Backward concurrency: weight gradients do not feed the data-gradient chain, so they run on a
  // side stream (forked per layer with an event, joined once at the end).  On the small inner-step
  // clips a single conv launch fills only 25-100 % of the CUs, and the two streams overlap.
  mutable hipStream_t side = nullptr;
  mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int side_streams = 1;                           // DVSR_BWD_STREAMS=0 disables
  int fork_every = 3;                             // weight gradients of this many layers share one fork (see backward)
  const char* bwd_unsupported = nullptr;          // set by build_backward when the tape has an op it cannot differentiate
  // Per-group parameter gradients: the batch holds `wgroups` groups of B / wgroups consecutive clips and the backward
  // writes one gradient PER GROUP (grad_params[i] = [wgroups][numel_i]) -- the frames of a video adapted from the same
  // weights run as one batch (test_dynavsr.py:208-277 with adapt_iter = 1; train_dynavsr.py:355-426), each keeping its own
  // gradient.  Data gradients and activations are per sample anyway; only the weight-gradient launches change.
  int wgroups = 1;
  // Per-group WEIGHTS as well (wsets == wgroups) or one set for the batch (wsets == 1): with per-group weights every
  // params[i] points to [wsets][numel_i] floats and batch group g convolves with set g -- the private copies of K frames
  // that have diverged (second and later inner steps; the adapted forwards) still run as one batch.
  int wsets = 1;
  // estimator plans under DVSR_EST_SPLIT=1: the 3x3 stride-1 convolutions over explicitly padded inputs (pad = 0; the data
  // gradient's pad = 2) take the exact 3-way bf16 split kernel too (cfg.bf16_mfma = 2)
  bool split_any_pad = false;
};

namespace dvsr {

struct Builder {
  dvsr_edvr_plan& p;
  int pcur = 0;  // parameter cursor (state-dict order)
  explicit Builder(dvsr_edvr_plan& plan) : p(plan) {}

  T alloc(const char* name, size_t numel) {
    T t;
    t.space = SP_ARENA;
    t.off = p.arena_floats;
    t.numel = numel;
    p.allocs.emplace_back(t.off, numel);
    p.arena_floats += (numel + 63) & ~(size_t)63;  // 256-byte aligned slots
    if (name && *name) p.named.emplace_back(name, t);
    return t;
  }
  static T view(const T& base, size_t off, size_t numel) {
    T t = base;
    t.off += off;
    t.numel = numel;
    return t;
  }
  struct CP { int w, b; };
  CP take() { CP c{pcur, pcur + 1}; pcur += 2; return c; }

  // y = act(conv(cat(x0, x1)) + b) [+ res]   (optionally pixel-shuffled)
  T conv(const char* name, CP cp, T x0, int c0, T x1, int c1, int N, int H, int W, int Cout, int ks,
         int stride, int act, T res = T(), int ps = 0, int x1_bdiv = 1, long long x0_bs = 0,
         long long x1_bs = 0, T y_override = T(), int pad = -1, int wmap = 0) {
    Op o;
    o.type = OP_CONV; o.name = name; o.pw = cp.w; o.pb = cp.b; o.pad = pad; o.wmap = wmap;
    o.x0 = x0; o.x1 = x1; o.res = res;
    o.N = N; o.c0 = c0; o.c1 = c1; o.H = H; o.W = W; o.Cout = Cout; o.ks = ks; o.stride = stride;
    o.act = act; o.ps = ps; o.x1_bdiv = x1_bdiv; o.x0_bs = x0_bs; o.x1_bs = x1_bs;
    const int Ho = conv_out(o, H), Wo = conv_out(o, W);
    o.y = y_override.valid() ? y_override : alloc(name, (size_t)N * Cout * Ho * Wo);
    // (the gradient arena mirrors this slot: one re-laid-out gradient per group)
    if (wmap) o.w2_off = alloc("", (size_t)std::max(p.wgroups, p.wsets) * Cout * (c0 + c1) * ks * ks).off;
    {
      // (estimator plans: also the 2x2 space-to-depth form of the 4x4 stride-2 convolutions, DVSR_EST_SPLIT2=0 keeps those
      // on the fp32 MFMA)
      const char* s2 = getenv("DVSR_EST_SPLIT2");
      const int split2 = s2 ? atoi(s2) : 1;   // (read per plan build, as DVSR_EST_SPLIT is; 2: forward launches only)
      const bool bf = p.cfg.bf16_mfma && (ks == 3 || (ks == 2 && p.split_any_pad && p.cfg.bf16_mfma == 2 && split2 && c0 % 16 == 0)) &&
                      stride == 1 && (pad < 0 || p.split_any_pad) && (c1 == 0 || c0 % 16 == 0);
      // bf16_mfma = 2: 8-row tiles (two 32-pixel rows per wave) once they still give ~a workgroup per CU
      static const int split_th8_from = getenv("DVSR_SPLIT_TH8_FROM") ? atoi(getenv("DVSR_SPLIT_TH8_FROM")) : 200;
      auto as_bf = [&](ConvGeo g, int ho, int wo, int cout, bool dgrad = false) {
        if (!bf || (ks == 2 && split2 == 2 && dgrad)) return g;
        // the 2x2 form only where it pays (launches of at least a workgroup per CU): small launches stay on the fp32 kernel
        if (ks == 2 && (long long)ceil_div(wo, 32) * ceil_div(ho, 4) * N * ceil_div(cout, 64) < 256) return g;
        g.cc = 16; g.th = 4; g.bf = p.cfg.bf16_mfma == 2 ? 2 : 1;
        if (g.bf == 2 && (long long)ceil_div(wo, 32) * ceil_div(ho, 8) * N * ceil_div(cout, 32 * g.mt) >= split_th8_from)
          g.th = 8;
        return g;
      };
      // the K-split small-grid kernel takes plain inputs with an explicit pad of 1 and 32-channel chunks
      // (bit 0), the DMA-halo kernel plain inputs in whole 8-channel chunks (bit 1)
      const bool plain = pad < 0 && !wmap && !p.cfg.bf16_mfma;
      // ... and the Winograd kernel where the DMA-halo kernel could run (bit 2; its epilogue stores plain or
      // PixelShuffle(2) tiles)
      const int ks_ok = (plain && (c1 == 0 || c0 % 32 == 0) ? 1 : 0) | (plain && c0 % 8 == 0 && c1 % 8 == 0 ? 2 | 4 : 0);
      const int fwd_ok = (ps == 0 || (ps == 2 && !res.valid())) ? ks_ok : (ks_ok & ~4);
      o.geo = as_bf(conv2_choose(ks, stride, N, Ho, Wo, Cout, c0 + c1, fwd_ok), Ho, Wo, Cout);
      o.wp_floats = (size_t)ceil_div(Cout, 64) * ceil_div(c0 + c1, o.geo.cc) * conv2_pch_cc(ks, o.geo.cc, o.geo.bf, o.geo.dma);
      o.wp_off = alloc("", o.wp_floats * p.wsets).off;   // one pack per weight set, consecutive
      // bit 3: the NO-GRAD forward may take the F(4x4, 3x3) kernel (never the data gradients: accumulate / mask epilogues)
      o.geo_ng = o.geo; o.wp_ng_off = o.wp_off; o.wp_ng_floats = o.wp_floats;
      if ((fwd_ok & 4) && !o.geo.bf) {
        const ConvGeo g5 = conv2_choose(ks, stride, N, Ho, Wo, Cout, c0 + c1, fwd_ok | 8);
        if (g5.dma == 5) {
          o.geo_ng = g5;
          o.wp_ng_floats = (size_t)ceil_div(Cout, 64) * ceil_div(c0 + c1, g5.cc) * conv2_pch_cc(ks, g5.cc, 0, 5);
          o.wp_ng_off = alloc("", o.wp_ng_floats * p.wsets).off;
        }
      }
      for (int which = 0; which < 2; ++which) {
        const int ci = which ? c1 : c0;
        if (!ci) continue;
        // dgrad = stride-1 conv over the input grid with Cout' = ci, Ctot' = Cout
        // (data gradient: the gradient tensor is the plain input unless it is pixel-shuffled or zero-dilated)
        o.dgeo[which] = as_bf(conv2_choose(ks, 1, N, H, W, ci, Cout, (!ps && stride == 1) ? ks_ok : 0), H, W, ci, true);
        o.dpk_floats[which] = (size_t)ceil_div(ci, 64) * ceil_div(Cout, o.dgeo[which].cc) *
                              conv2_pch_cc(ks, o.dgeo[which].cc, o.dgeo[which].bf, o.dgeo[which].dma);
        o.dpk_off[which] = p.dpack_floats;
        p.dpack_floats += o.dpk_floats[which] * p.wsets;
      }
    }
    p.ops.push_back(o);
    return o.y;
  }
  T dcn(const char* name, int pw, int pb, T x, T om, int N, int C, int H, int W, int dg, int act) {
    Op o;
    o.type = OP_DCN; o.name = name; o.pw = pw; o.pb = pb; o.x0 = x; o.x1 = om;
    o.N = N; o.c0 = C; o.H = H; o.W = W; o.Cout = C; o.dg = dg; o.act = act;
    o.y = alloc(name, (size_t)N * C * H * W);
    if (C % (dg * 8) == 0) {  // LDS-sampler kernel: weights packed like a conv with 8-channel chunks
      o.wp_floats = (size_t)ceil_div(C, 64) * (C / 8) * mdcn_pack_floats();
      o.wp_off = alloc("", o.wp_floats * p.wsets).off;
    }
    p.ops.push_back(o);
    return o.y;
  }
  T up(const char* name, T x, size_t planes, int H, int W, int S, float mul) {
    Op o;
    o.type = OP_UP; o.name = name; o.x0 = x; o.planes = planes; o.H = H; o.W = W; o.S = S; o.mul = mul;
    o.y = alloc(name, planes * H * W * S * S);
    p.ops.push_back(o);
    return o.y;
  }
  // explicit padding / space-to-depth / temporal gather (pad.hip); x: [N][C][H][W]
  T padop(const char* name, T x, int mode, int N, int C, int H, int W, int Tn) {
    Op o;
    o.type = OP_PAD; o.name = name; o.x0 = x; o.pmode = mode; o.N = N; o.c0 = C; o.H = H; o.W = W; o.T = Tn;
    o.y = alloc(name, pad_out_numel(mode, (size_t)N, C, H, W));
    p.ops.push_back(o);
    return o.y;
  }
  void pool(const char* name, T x, size_t planes, int H, int W, T& ymax, T& yavg) {
    Op o;
    o.type = OP_POOL; o.name = name; o.x0 = x; o.planes = planes; o.H = H; o.W = W;
    const size_t n = planes * (size_t)((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    o.y = ymax = alloc((std::string(name) + "_max").c_str(), n);
    o.y2 = yavg = alloc((std::string(name) + "_avg").c_str(), n);
    p.ops.push_back(o);
  }
};

// Walks the network in forward order; parameters are consumed in state-dict order, which is
// NOT forward order inside PCD/TSA, hence the explicit index bookkeeping there.
static int build_plan(dvsr_edvr_plan& p) {
  const dvsr_edvr_config& c = p.cfg;
  const int B = p.B, Nf = c.nframes, C = c.nf, H = p.H, W = p.W, BN = B * Nf, dg = c.groups;
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
  const size_t HW = (size_t)H * W, HW2 = (size_t)H2 * W2, HW4 = (size_t)H4 * W4;
  Builder b(p);
  const int L = ACT_LRELU, R = ACT_RELU, NO = ACT_NONE;
  T none;
  T xin; xin.space = SP_INPUT; xin.off = 0; xin.numel = (size_t)BN * 3 * HW;

  // ---- feature extraction (EDVR_arch.py:272-279)
  T f1 = b.conv("conv_first", b.take(), xin, 3, none, 0, BN, H, W, C, 3, 1, L);
  for (int i = 0; i < c.front_RBs; ++i) {
    auto p1 = b.take(); auto p2 = b.take();
    T t = b.conv("fe_rb_a", p1, f1, C, none, 0, BN, H, W, C, 3, 1, R);
    f1 = b.conv("fe_rb_b", p2, t, C, none, 0, BN, H, W, C, 3, 1, NO, f1);
  }
  p.named.emplace_back("L1_fea", f1);
  T f2 = b.conv("fea_L2_conv1", b.take(), f1, C, none, 0, BN, H, W, C, 3, 2, L);
  f2 = b.conv("L2_fea", b.take(), f2, C, none, 0, BN, H2, W2, C, 3, 1, L);
  T f3 = b.conv("fea_L3_conv1", b.take(), f2, C, none, 0, BN, H2, W2, C, 3, 2, L);
  f3 = b.conv("L3_fea", b.take(), f3, C, none, 0, BN, H4, W4, C, 3, 1, L);

  // ---- PCD alignment, all N frames batched (EDVR_arch.py:95-128, 281-297)
  // state-dict order inside pcd_align: L3_offset_conv1, L3_offset_conv2, L3_dcnpack{w,b,com.w,com.b},
  // L2_offset_conv1..3, L2_dcnpack{..}, L2_fea_conv, L1_offset_conv1..3, L1_dcnpack{..}, L1_fea_conv,
  // cas_offset_conv1, cas_offset_conv2, cas_dcnpack{..}
  struct DP { int w, b; Builder::CP com; };
  auto take_dcn = [&]() { DP d; d.w = b.pcur; d.b = b.pcur + 1; b.pcur += 2; d.com = b.take(); return d; };
  auto L3o1 = b.take(); auto L3o2 = b.take(); DP L3d = take_dcn();
  auto L2o1 = b.take(); auto L2o2 = b.take(); auto L2o3 = b.take(); DP L2d = take_dcn(); auto L2f = b.take();
  auto L1o1 = b.take(); auto L1o2 = b.take(); auto L1o3 = b.take(); DP L1d = take_dcn(); auto L1f = b.take();
  auto cso1 = b.take(); auto cso2 = b.take(); DP csd = take_dcn();
  const int ctr = c.center;
  // reference-frame views: batch item b of the "x1" operand is frame (b*Nf + ctr)
  T ref1 = Builder::view(f1, (size_t)ctr * C * HW, 0), ref2 = Builder::view(f2, (size_t)ctr * C * HW2, 0),
    ref3 = Builder::view(f3, (size_t)ctr * C * HW4, 0);
  const long long rs1 = (long long)Nf * C * HW, rs2 = (long long)Nf * C * HW2, rs3 = (long long)Nf * C * HW4;
  // L3
  T o3 = b.conv("L3_offset_conv1", L3o1, f3, C, ref3, C, BN, H4, W4, C, 3, 1, L, none, 0, Nf, 0, rs3);
  o3 = b.conv("L3_offset", L3o2, o3, C, none, 0, BN, H4, W4, C, 3, 1, L);
  T om3 = b.conv("L3_om", L3d.com, o3, C, none, 0, BN, H4, W4, dg * 27, 3, 1, NO);
  T fe3 = b.dcn("L3_aligned", L3d.w, L3d.b, f3, om3, BN, C, H4, W4, dg, L);
  // L2
  T o2 = b.conv("L2_offset_conv1", L2o1, f2, C, ref2, C, BN, H2, W2, C, 3, 1, L, none, 0, Nf, 0, rs2);
  T u3 = b.up("L3_offset_up", o3, (size_t)BN * C, H4, W4, 2, 2.f);
  o2 = b.conv("L2_offset_conv2", L2o2, o2, C, u3, C, BN, H2, W2, C, 3, 1, L);
  o2 = b.conv("L2_offset", L2o3, o2, C, none, 0, BN, H2, W2, C, 3, 1, L);
  T om2 = b.conv("L2_om", L2d.com, o2, C, none, 0, BN, H2, W2, dg * 27, 3, 1, NO);
  T d2 = b.dcn("L2_dcn", L2d.w, L2d.b, f2, om2, BN, C, H2, W2, dg, NO);
  T uf3 = b.up("L3_aligned_up", fe3, (size_t)BN * C, H4, W4, 2, 1.f);
  T fe2 = b.conv("L2_aligned", L2f, d2, C, uf3, C, BN, H2, W2, C, 3, 1, L);
  // L1
  T o1 = b.conv("L1_offset_conv1", L1o1, f1, C, ref1, C, BN, H, W, C, 3, 1, L, none, 0, Nf, 0, rs1);
  T u2 = b.up("L2_offset_up", o2, (size_t)BN * C, H2, W2, 2, 2.f);
  o1 = b.conv("L1_offset_conv2", L1o2, o1, C, u2, C, BN, H, W, C, 3, 1, L);
  o1 = b.conv("L1_offset", L1o3, o1, C, none, 0, BN, H, W, C, 3, 1, L);
  T om1 = b.conv("L1_om", L1d.com, o1, C, none, 0, BN, H, W, dg * 27, 3, 1, NO);
  T d1 = b.dcn("L1_dcn", L1d.w, L1d.b, f1, om1, BN, C, H, W, dg, NO);
  T uf2 = b.up("L2_aligned_up", fe2, (size_t)BN * C, H2, W2, 2, 1.f);
  T fe1 = b.conv("L1_aligned", L1f, d1, C, uf2, C, BN, H, W, C, 3, 1, NO);
  // cascade
  T oc = b.conv("cas_offset_conv1", cso1, fe1, C, ref1, C, BN, H, W, C, 3, 1, L, none, 0, Nf, 0, rs1);
  oc = b.conv("cas_offset", cso2, oc, C, none, 0, BN, H, W, C, 3, 1, L);
  T omc = b.conv("cas_om", csd.com, oc, C, none, 0, BN, H, W, dg * 27, 3, 1, NO);
  T aligned = b.dcn("aligned", csd.w, csd.b, fe1, omc, BN, C, H, W, dg, L);  // [B][Nf][C][H][W]

  // ---- TSA fusion (EDVR_arch.py:163-203); state-dict order: tAtt_1, tAtt_2, fea_fusion, sAtt_1..5,
  // sAtt_L1..L3, sAtt_add_1, sAtt_add_2
  auto tA1 = b.take(); auto tA2 = b.take(); auto ffu = b.take(); auto s1 = b.take(); auto s2 = b.take();
  auto s3 = b.take(); auto s4 = b.take(); auto s5 = b.take(); auto sL1 = b.take(); auto sL2 = b.take();
  auto sL3 = b.take(); auto sa1 = b.take(); auto sa2 = b.take();
  T emb_ref = b.conv("tsa_emb_ref", tA2, Builder::view(aligned, (size_t)ctr * C * HW, 0), C, none, 0, B, H,
                     W, C, 3, 1, NO, none, 0, 1, (long long)Nf * C * HW);
  T emb = b.conv("tsa_emb", tA1, aligned, C, none, 0, BN, H, W, C, 3, 1, NO);
  T cor = b.alloc("tsa_cor", (size_t)BN * HW);
  T gated = b.alloc("tsa_gated", (size_t)BN * C * HW);
  {
    Op o; o.type = OP_GATE; o.name = "tsa_gate"; o.x0 = emb; o.x1 = emb_ref; o.res = aligned;
    o.y = cor; o.y2 = gated; o.gB = B; o.gN = Nf; o.gC = C; o.gHW = HW;
    p.ops.push_back(o);
  }
  T fea = b.conv("tsa_fea", ffu, gated, Nf * C, none, 0, B, H, W, C, 1, 1, L);
  T att = b.conv("tsa_att1", s1, gated, Nf * C, none, 0, B, H, W, C, 1, 1, L);
  {  // fea_fusion and sAtt_1 read the same 5 x 64-channel tensor (EDVR_arch.py:183-202): one pass over it
    static const bool dual_on = [] { const char* v = getenv("DVSR_TSA_DUAL"); return !(v && v[0] == '0'); }();
    const int ia = (int)p.ops.size() - 2, ib = ia + 1;
    if (dual_on && C == 64 && !p.cfg.bf16_mfma) { p.ops[ia].dual = ib; p.ops[ib].fused_into = ia; }
  }
  T pmx, pav;
  b.pool("tsa_pool1", att, (size_t)B * C, H, W, pmx, pav);
  att = b.conv("tsa_att2", s2, pmx, C, pav, C, B, H2, W2, C, 1, 1, L);
  T attL = b.conv("tsa_attL1", sL1, att, C, none, 0, B, H2, W2, C, 1, 1, L);
  T qmx, qav;
  b.pool("tsa_pool2", attL, (size_t)B * C, H2, W2, qmx, qav);
  attL = b.conv("tsa_attL2", sL2, qmx, C, qav, C, B, H4, W4, C, 3, 1, L);
  attL = b.conv("tsa_attL3", sL3, attL, C, none, 0, B, H4, W4, C, 3, 1, L);
  T attLu = b.up("tsa_attL_up", attL, (size_t)B * C, H4, W4, 2, 1.f);
  T a3 = b.conv("tsa_att3", s3, att, C, none, 0, B, H2, W2, C, 3, 1, L);
  T att3 = b.alloc("tsa_att3_sum", (size_t)B * C * HW2);
  {
    Op o; o.type = OP_ADD; o.name = "tsa_att3_add"; o.x0 = a3; o.x1 = attLu; o.y = att3;
    p.ops.push_back(o);
  }
  T a4 = b.conv("tsa_att4", s4, att3, C, none, 0, B, H2, W2, C, 1, 1, L);
  T a4u = b.up("tsa_att4_up", a4, (size_t)B * C, H2, W2, 2, 1.f);
  T a5 = b.conv("tsa_att", s5, a4u, C, none, 0, B, H, W, C, 3, 1, NO);
  T ad = b.conv("tsa_add1", sa1, a5, C, none, 0, B, H, W, C, 1, 1, L);
  ad = b.conv("tsa_add2", sa2, ad, C, none, 0, B, H, W, C, 1, 1, NO);
  T out = b.alloc("tsa_out", (size_t)B * C * HW);
  {
    Op o; o.type = OP_BLEND; o.name = "tsa_blend"; o.x0 = fea; o.x1 = a5; o.res = ad; o.y = out;
    p.ops.push_back(o);
  }

  // ---- reconstruction (EDVR_arch.py:302-312)
  for (int i = 0; i < c.back_RBs; ++i) {
    auto p1 = b.take(); auto p2 = b.take();
    T t = b.conv("rc_rb_a", p1, out, C, none, 0, B, H, W, C, 3, 1, R);
    out = b.conv("rc_rb_b", p2, t, C, none, 0, B, H, W, C, 3, 1, NO, out);
  }
  p.named.emplace_back("recon", out);
  int h = H, w = W;
  if (c.scale == 4) {
    out = b.conv("upconv1", b.take(), out, C, none, 0, B, h, w, C * 4, 3, 1, L, none, 2);
    h *= 2; w *= 2;
    out = b.conv("upconv2", b.take(), out, C, none, 0, B, h, w, 256, 3, 1, L, none, 2);
  } else {
    out = b.conv("upconv2", b.take(), out, C, none, 0, B, h, w, 256, 3, 1, L, none, 2);
  }
  h *= 2; w *= 2;
  out = b.conv("HRconv", b.take(), out, 64, none, 0, B, h, w, 64, 3, 1, L);
  // base = bilinear x scale of the centre LR frame; one plane group per batch item
  T base = b.alloc("base", (size_t)B * 3 * h * w);
  for (int bi = 0; bi < B; ++bi) {
    Op o; o.type = OP_UP; o.name = "base_up";
    o.x0 = Builder::view(xin, ((size_t)bi * Nf + ctr) * 3 * HW, 3 * HW);
    o.planes = 3; o.H = H; o.W = W; o.S = c.scale; o.mul = 1.f;
    o.y = Builder::view(base, (size_t)bi * 3 * h * w, (size_t)3 * h * w);
    p.ops.push_back(o);
  }
  T yout; yout.space = SP_OUTPUT; yout.off = 0; yout.numel = (size_t)B * 3 * h * w;
  b.conv("conv_last", b.take(), out, 64, none, 0, B, h, w, 3, 3, 1, NO, base, 0, 1, 0, 0, yout);
  p.n_params = b.pcur;
  return DVSR_OK;
}


// Builds the backward tape by walking the forward tape in reverse.  Gradient buffers mirror the
// activation arena (same offsets in a second arena).  The first contribution to a gradient
// buffer writes it when it covers the whole slot with plain stores; partial (views) or atomic
// (DCN input gradient) first contributions zero the slot first; later ones accumulate.
struct BackBuilder {
  dvsr_edvr_plan& p;
  std::vector<char> written;  // per allocation: 0 untouched, 1 written, 2 = a whole-slot copy is PENDING (see copyadd)
  struct Pending { Ref src; size_t n; int fwd; };
  std::vector<Pending> pending;  // per allocation, valid while written == 2
  bool gx_zeroed = false;
  explicit BackBuilder(dvsr_edvr_plan& plan) : p(plan), written(plan.allocs.size(), 0), pending(plan.allocs.size()) {}

  int alloc_index(size_t off) const {
    int lo = 0, hi = (int)p.allocs.size() - 1, ans = -1;
    while (lo <= hi) {
      const int mid = (lo + hi) / 2;
      if (p.allocs[mid].first <= off) { ans = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return ans;
  }
  static Ref act(const T& t) {
    Ref r;
    r.space = t.space == SP_ARENA ? R_ACT : (t.space == SP_INPUT ? R_X : R_NONE);
    r.off = t.off;
    return r;
  }
  static Ref grad(const T& t) {
    Ref r;
    r.space = t.space == SP_ARENA ? R_GRAD : (t.space == SP_INPUT ? R_GX : (t.space == SP_OUTPUT ? R_GOUT : R_NONE));
    r.off = t.off;
    return r;
  }
  // A pending copy becomes a real launch (someone is about to read or accumulate into the slot).
  void materialize(int ai) {
    if (ai < 0 || written[ai] != 2) return;
    BOp o;
    o.type = B_COPYADD; o.fwd = pending[ai].fwd; o.a.space = R_GRAD; o.a.off = p.allocs[ai].first; o.b = pending[ai].src;
    o.n = pending[ai].n; o.accum = 0;
    p.bops.push_back(o);
    written[ai] = 1;
  }
  void materialize(const T& t) { if (t.space == SP_ARENA) materialize(alloc_index(t.off)); }
  // A data-gradient launch that covers the whole slot takes a pending copy as its residual input instead
  // (gx = dgrad + gy: the skip connection of a residual block costs no launch of its own).
  bool take_pending_as_residual(const T& t, Ref* src) {
    if (t.space != SP_ARENA) return false;
    const int ai = alloc_index(t.off);
    if (written[ai] != 2 || t.off != p.allocs[ai].first || t.numel != p.allocs[ai].second || pending[ai].n != t.numel)
      return false;
    *src = pending[ai].src;
    written[ai] = 1;
    return true;
  }
  // Declares a contribution to grad(t); returns the accumulate flag for it.
  int contribute(const T& t, bool full, int fwd) {
    if (t.space == SP_INPUT) return 1;  // gx is zeroed once up front, everything accumulates
    const int ai = alloc_index(t.off);
    const bool whole = full && t.off == p.allocs[ai].first && t.numel == p.allocs[ai].second;
    materialize(ai);
    if (written[ai]) return 1;
    written[ai] = 1;
    if (whole) return 0;
    BOp m;
    m.type = B_MEMSET; m.fwd = fwd; m.a.space = R_GRAD; m.a.off = p.allocs[ai].first; m.n = p.allocs[ai].second;
    p.bops.push_back(m);
    return 1;
  }
  void copyadd(const T& dst, const Ref& src, size_t n, int fwd) {
    if (dst.space == SP_ARENA) {   // first, whole-slot contribution: keep the copy pending
      const int ai = alloc_index(dst.off);
      static const bool defer = [] { const char* v = getenv("DVSR_FUSE_RES_BWD"); return !(v && v[0] == '0'); }();
      if (defer && !written[ai] && dst.off == p.allocs[ai].first && dst.numel == p.allocs[ai].second && n == dst.numel) {
        written[ai] = 2;
        pending[ai] = Pending{src, n, fwd};
        return;
      }
    }
    BOp o;
    o.type = B_COPYADD; o.fwd = fwd; o.a = grad(dst); o.b = src; o.n = n;
    o.accum = contribute(dst, true, fwd);
    p.bops.push_back(o);
  }
};

// Index of the conv whose activated output is exactly tensor `t`, if the op at `consumer` is the ONLY reader of
// that output: then the consumer's gradient launch is the sole, final contribution to grad(t) and can apply
// the producer's activation backward itself (one launch less per layer).  -1 otherwise.
static int sole_producer_with_act(const dvsr_edvr_plan& p, const T& t, int consumer) {
  if (t.space != SP_ARENA) return -1;
  int prod = -1;
  for (int j = 0; j < consumer; ++j) {
    const Op& q = p.ops[j];
    if (q.type == OP_CONV && q.act != ACT_NONE && q.y.space == SP_ARENA && q.y.off == t.off && q.y.numel == t.numel &&
        !q.ps)
      prod = j;
  }
  if (prod < 0) return -1;
  const Op& q = p.ops[prod];
  int readers = 0;
  for (size_t j = 0; j < p.ops.size(); ++j) {
    const Op& r = p.ops[j];
    for (const T* in : {&r.x0, &r.x1, &r.res})
      if (in->space == SP_ARENA && in->off < q.y.off + q.y.numel && in->off + in->numel > q.y.off) ++readers;
  }
  return readers == 1 ? prod : -1;
}

static void build_backward(dvsr_edvr_plan& p) {
  BackBuilder bb(p);
  size_t tmp = 0, scratch = 0, wscratch = 0;
  std::vector<char> act_fused(p.ops.size(), 0);  // the producer's B_ACT is done by its consumer's launch
  // DVSR_FUSE_ACT_BWD=0 keeps every activation backward a separate launch (A/B aid); the un-pipelined conv
  // kernel (DVSR_CONV_V1) has no mask input
  const bool fuse_act = !p.use_v1 && [] { const char* v = getenv("DVSR_FUSE_ACT_BWD"); return !(v && v[0] == '0'); }();
  for (int i = (int)p.ops.size() - 1; i >= 0; --i) {
    const Op& o = p.ops[i];
    const Ref gy = BackBuilder::grad(o.y);
    bb.materialize(o.y);    // this op reads the gradients of its outputs: pending copies into them become launches
    bb.materialize(o.y2);
    if (o.type == OP_GATE) bb.materialize(o.res);
    switch (o.type) {
      case OP_CONV: {
        const int Ho = conv_out(o, o.H), Wo = conv_out(o, o.W);
        // y = act(conv + b) + res: the residual gradient is the RAW gy, and act' would need the sign of the pre-residual
        // value, which the tape does not keep (B_ACT masks gy in place by the sign of y; a deferred residual copy would
        // then read the masked gy).  No tape combines the two (fe_rb_b, rc_rb_b, conv_last are ACT_NONE): refuse it.
        if (o.res.valid() && o.act != ACT_NONE) p.bwd_unsupported = "a convolution with both a residual input and an activation";
        if (o.res.valid()) bb.copyadd(o.res, gy, o.y.numel, i);
        if (o.act != ACT_NONE && !act_fused[i]) {
          BOp a; a.type = B_ACT; a.fwd = i; a.a = gy; a.b = BackBuilder::act(o.y); a.n = o.y.numel;
          p.bops.push_back(a);
        }
        for (int which = 0; which < 2; ++which) {
          if (which == 1 && !o.c1) break;
          BOp w; w.type = B_WGRAD; w.fwd = i; w.which = which; w.a = BackBuilder::act(which ? o.x1 : o.x0); w.b = gy;
          p.bops.push_back(w);
          // every weight gradient owns its slot region: the slot sums of ALL layers are reduced by one
          // batched launch at the end of the backward (wgrad_reduce_batch)
          BOp& wr = p.bops.back();
          wr.ws_bytes = (conv2d_wgrad_workspace_bytes(o.N, which ? o.c1 : o.c0, o.H, o.W, o.Cout, o.ks, o.stride,
                                                      conv_pad(o), p.wgroups) + 255) & ~(size_t)255;
          wr.ws_off = wscratch;
          wscratch += wr.ws_bytes;
          if (o.wmap) {  // dW of the re-laid-out copy -> gradient of the 4x4 parameter (same stream as the wgrad)
            BOp u; u.type = B_WUNMAP; u.fwd = i;
            p.bops.push_back(u);
          }
        }
        for (int which = 0; which < 2; ++which) {
          if ((which == 1 && !o.c1) || o.no_dgrad) break;
          const T& xin = which ? o.x1 : o.x0;
          const int ci = which ? o.c1 : o.c0;
          const bool strided = which ? (o.x1_bdiv > 1 || o.x1_bs != 0) : (o.x0_bs != 0);
          BOp d; d.type = B_DGRAD; d.fwd = i; d.which = which; d.b = gy;
          if (!strided) {
            T full = xin; full.numel = (size_t)o.N * ci * o.H * o.W;
            if (!p.use_v1 && bb.take_pending_as_residual(full, &d.e)) d.accum = 0;   // gx = dgrad + (pending copy's source)
            else d.accum = bb.contribute(full, true, i);
            d.a = BackBuilder::grad(xin);
            if (!d.accum && !o.c1 && fuse_act && d.e.space == R_NONE) {
              const int prod = sole_producer_with_act(p, full, i);
              if (prod >= 0) { d.mask_op = prod; act_fused[prod] = 1; }
            }
            p.bops.push_back(d);
          } else {
            // dense dgrad into the staging buffer, then fold frames into the strided view
            d.accum = 0; d.a.space = R_TMP; d.a.off = 0;
            p.bops.push_back(d);
            const int bdiv = which ? o.x1_bdiv : 1;
            const size_t per = (size_t)ci * o.H * o.W;
            tmp = std::max(tmp, (size_t)o.N * per);
            BOp r; r.type = B_REDUCE; r.fwd = i; r.a = BackBuilder::grad(xin); r.b.space = R_TMP; r.b.off = 0;
            r.B = o.N / bdiv; r.cnt = bdiv; r.per = per;
            r.bs = which ? (o.x1_bs ? o.x1_bs : (long long)per) : o.x0_bs;
            T part = xin; part.numel = 1;  // never "whole": forces zero-init when first
            r.accum = bb.contribute(part, false, i);
            // the memset (if any) must precede the dgrad+reduce pair: contribute() already pushed it
            p.bops.push_back(r);
          }
        }
        (void)Ho; (void)Wo;
        break;
      }
      case OP_DCN: {
        if (o.act != ACT_NONE) {
          BOp a; a.type = B_ACT; a.fwd = i; a.a = gy; a.b = BackBuilder::act(o.y); a.n = o.y.numel;
          p.bops.push_back(a);
        }
        BOp d; d.type = B_DCN; d.fwd = i; d.b = gy;
        T xv = o.x0; xv.numel = 1;
        bb.contribute(xv, false, i);       // atomics: zero first unless already written
        bb.contribute(o.x1, true, i);      // om gradient is written in full
        d.a = BackBuilder::grad(o.x0); d.c = BackBuilder::grad(o.x1);
        p.bops.push_back(d);
        scratch = std::max(scratch, mdcn_backward_workspace_bytes(o.N, o.c0, o.H, o.W, o.Cout, 1, 1, 1, p.wgroups));
        break;
      }
      case OP_UP: {
        BOp u; u.type = B_UP; u.fwd = i; u.b = gy; u.a = BackBuilder::grad(o.x0);
        T xin = o.x0; xin.numel = o.planes * o.H * o.W;
        u.accum = bb.contribute(xin, true, i);
        p.bops.push_back(u);
        break;
      }
      case OP_POOL: {
        BOp u; u.type = B_POOL; u.fwd = i; u.a = BackBuilder::grad(o.x0); u.b = BackBuilder::grad(o.y);
        u.c = BackBuilder::grad(o.y2);
        u.accum = bb.contribute(o.x0, true, i);
        p.bops.push_back(u);
        break;
      }
      case OP_GATE: {
        BOp g; g.type = B_GATE; g.fwd = i; g.b = BackBuilder::grad(o.y2);
        g.a = BackBuilder::grad(o.x0); g.c = BackBuilder::grad(o.x1); g.d = BackBuilder::grad(o.res);
        bb.contribute(o.x0, true, i); bb.contribute(o.x1, true, i); bb.contribute(o.res, true, i);
        p.bops.push_back(g);
        break;
      }
      case OP_BLEND: {
        bb.copyadd(o.res, gy, o.y.numel, i);  // d/d(att_add) = g
        BOp b; b.type = B_BLEND; b.fwd = i; b.b = gy; b.a = BackBuilder::grad(o.x0); b.c = BackBuilder::grad(o.x1);
        bb.contribute(o.x0, true, i);
        b.accum = bb.contribute(o.x1, true, i);
        p.bops.push_back(b);
        break;
      }
      case OP_ADD:
        bb.copyadd(o.x0, gy, o.y.numel, i);
        bb.copyadd(o.x1, gy, o.y.numel, i);
        break;
      case OP_PAD: {
        if (o.x0.space == SP_INPUT || o.no_dgrad) break;  // no gradient w.r.t. the network input
        BOp f; f.type = B_PADFOLD; f.fwd = i; f.b = gy; f.a = BackBuilder::grad(o.x0);
        T xin = o.x0; xin.numel = (size_t)o.N * o.c0 * o.H * o.W;
        f.accum = bb.contribute(xin, true, i);
        if (!f.accum && fuse_act) {
          const int prod = sole_producer_with_act(p, xin, i);
          if (prod >= 0) { f.mask_op = prod; act_fused[prod] = 1; }
        }
        p.bops.push_back(f);
        break;
      }
      case OP_MEANSUB:
        break;  // x is data: no gradient
      case OP_ADDMEAN: {
        BOp f; f.type = B_ADDMEAN; f.fwd = i; f.b = gy; f.a = BackBuilder::grad(o.x0);
        bb.contribute(o.x0, true, i);
        p.bops.push_back(f);
        break;
      }
    }
  }
  for (size_t ai = 0; ai < p.allocs.size(); ++ai) bb.materialize((int)ai);
  for (size_t i = 0; i < p.ops.size(); ++i)   // a conv without a dense output needs its activation backward in the pad fold
    if (p.ops[i].type == OP_CONV && p.ops[i].pad_out && p.ops[i].act != ACT_NONE && !act_fused[i])
      p.bwd_unsupported = "a padded-store convolution whose activation backward is not fused into the pad fold";
  p.tmp_floats = (tmp + 63) & ~(size_t)63;
  p.scratch_bytes = (scratch + 255) & ~(size_t)255;
  p.wscratch_bytes = (wscratch + 255) & ~(size_t)255;
}

struct BBases {
  float* arena = nullptr; float* garena = nullptr; const float* x = nullptr; float* gx = nullptr;
  const float* gout = nullptr; float* tmp = nullptr; float* dpack = nullptr;
  bool use_v1 = false;
  float* at(const Ref& r) const {
    switch (r.space) {
      case R_ACT: return arena + r.off;
      case R_GRAD: return garena + r.off;
      case R_X: return const_cast<float*>(x) + r.off;
      case R_GX: return gx ? gx + r.off : nullptr;
      case R_GOUT: return const_cast<float*>(gout) + r.off;
      case R_TMP: return tmp + r.off;
      default: return nullptr;
    }
  }
};

// DVSR_WGRAD_SPLIT3=0: the 3x3 stride-1 weight gradients stay on the fp32 MFMA kernel (A/B aid); default: the exact 3-way
// bf16 split kernel (conv2d_wgrad_bf16.hip)
static bool wgrad_split3_on() {
  static const bool on = [] { const char* v = getenv("DVSR_WGRAD_SPLIT3"); return !(v && v[0] == '0'); }();
  return on;
}

// Argument marshalling of the two gradient launches of a conv, shared by the separate and the fused paths.
static int prep_wgrad(const dvsr_edvr_plan& p, const BOp& b, float* const* GP, const BBases& bs, void* scratch,
                      size_t scratch_bytes, hipStream_t st, WgradReduceEntry* defer, WgradLaunch* out) {
  const Op* o = &p.ops[b.fwd];
  const int ci = b.which ? o->c1 : o->c0;
  float* dW = o->wmap ? bs.garena + o->w2_off : GP[o->pw];
  return conv2d_wgrad_prepare(bs.at(b.a), b.which ? o->x1_bs : o->x0_bs, b.which ? o->x1_bdiv : 1, bs.at(b.b), o->ps, dW,
                              b.which ? nullptr : GP[o->pb], o->N, ci, o->H, o->W, o->Cout, o->c0 + o->c1,
                              b.which ? o->c0 : 0, o->ks, o->stride, scratch, scratch_bytes, st, 1, conv_pad(*o), defer, out,
                              (p.cfg.bf16_mfma == 1 && !o->wmap) ? 1 : (wgrad_split3_on() ? 2 : 0), p.wgroups,
                              (long long)o->Cout * (o->c0 + o->c1) * o->ks * o->ks, o->Cout);
}

// ConvExtra of a launch whose batch items take per-sample weight sets (plan.wsets > 1)
static inline void set_wsets(const dvsr_edvr_plan& p, int N, size_t pack_floats, int cout, ConvExtra* ex) {
  if (p.wsets <= 1) return;
  ex->wdiv = N / p.wsets; ex->w_gs = (long long)pack_floats; ex->b_gs = cout;
}

static void dgrad_desc(const dvsr_edvr_plan& p, const BOp& b, const float* const* P, const BBases& bs,
                       dvsr_conv2d_desc* gd, ConvExtra* exd) {
  const Op* o = &p.ops[b.fwd];
  const int Ho = conv_out(*o, o->H), Wo = conv_out(*o, o->W);
  dvsr_conv2d_desc g = {};
  g.x0 = bs.at(b.b); g.w = o->wmap ? bs.arena + o->w2_off : P[o->pw]; g.y = bs.at(b.a); g.N = o->N; g.c0 = o->Cout;
  g.Cout = b.which ? o->c1 : o->c0;
  g.ks = o->ks; g.stride = 1; g.pad = o->ks - 1 - conv_pad(*o); g.act = ACT_NONE; g.x1_bdiv = 1;
  g.res = bs.at(b.e);   // skip-connection gradient folded into this launch (BackBuilder::take_pending_as_residual)
  ConvExtra ex;
  ex.wt = 1; ex.w_ctot = o->c0 + o->c1; ex.w_coff = b.which ? o->c0 : 0; ex.accum = b.accum; ex.in_ps = o->ps ? 1 : 0;
  if (b.mask_op >= 0 && !bs.use_v1) { ex.gmask = bs.arena + p.ops[b.mask_op].y.off; ex.gmask_act = p.ops[b.mask_op].act; }
  if (o->stride == 2) { ex.in_dil = 2; ex.Hs = Ho; ex.Ws = Wo; g.H = o->H; g.W = o->W; }
  else { g.H = Ho; g.W = Wo; }
  set_wsets(p, o->N, o->dpk_floats[b.which], 0, &ex);
  *gd = g; *exd = ex;
}

static int run_backward_op(const dvsr_edvr_plan& p, const BOp& b, const float* const* P, float* const* GP,
                           const BBases& bs, void* scratch, size_t scratch_bytes, hipStream_t st,
                           int scratch_is_zero = 0, WgradReduceEntry* defer = nullptr) {
  const Op* o = b.fwd >= 0 ? &p.ops[b.fwd] : nullptr;
  switch (b.type) {
    case B_MEMSET: {
      hipError_t e = hipMemsetAsync(bs.at(b.a), 0, b.n * sizeof(float), st);
      DVSR_REQUIRE(e == hipSuccess, DVSR_ERR_HIP, "backward memset: %s", hipGetErrorString(e));
      return DVSR_OK;
    }
    case B_COPYADD: {
      float* dst = bs.at(b.a);
      if (!dst) return DVSR_OK;  // gradient w.r.t. the external input not requested
      if (b.accum) return add_inplace(dst, bs.at(b.b), b.n, st);
      hipError_t e = hipMemcpyAsync(dst, bs.at(b.b), b.n * sizeof(float), hipMemcpyDeviceToDevice, st);
      DVSR_REQUIRE(e == hipSuccess, DVSR_ERR_HIP, "backward copy: %s", hipGetErrorString(e));
      return DVSR_OK;
    }
    case B_ACT:
      return act_bwd_inplace(bs.at(b.a), bs.at(b.b), b.n, o->act, st);
    case B_WGRAD: {
      const int ci = b.which ? o->c1 : o->c0;
      float* dW = o->wmap ? bs.garena + o->w2_off : GP[o->pw];
      return conv2d_wgrad_run(bs.at(b.a), b.which ? o->x1_bs : o->x0_bs, b.which ? o->x1_bdiv : 1, bs.at(b.b),
                              o->ps, dW, b.which ? nullptr : GP[o->pb], o->N, ci, o->H, o->W, o->Cout,
                              o->c0 + o->c1, b.which ? o->c0 : 0, o->ks, o->stride, scratch, scratch_bytes, st,
                              scratch_is_zero, conv_pad(*o), defer, p.wgroups,
                              (long long)o->Cout * (o->c0 + o->c1) * o->ks * o->ks, o->Cout);
    }
    case B_WUNMAP:   // (the map is per output channel: the stacked per-group gradients are a [wgroups * Cout] tensor)
      return w4_to_s2d(bs.garena + o->w2_off, GP[o->pw], p.wgroups * o->Cout, o->c0 / 4, 1, st);
    case B_PADFOLD: {
      float* gx = bs.at(b.a);
      if (!gx) return DVSR_OK;
      // (a producer that stored straight into this op's padded tensor left no dense output: its mask is read there)
      const bool mpad = b.mask_op >= 0 && p.ops[b.mask_op].pad_out;
      return pad_bwd(bs.at(b.b), gx, o->pmode, o->N, o->c0, o->H, o->W, o->T, b.accum, st,
                     b.mask_op >= 0 ? bs.arena + (mpad ? o->y.off : p.ops[b.mask_op].y.off) : nullptr,
                     b.mask_op >= 0 ? p.ops[b.mask_op].act : 0, mpad ? 1 : 0);
    }
    case B_ADDMEAN:
      return addmean_bwd(bs.at(b.b), bs.at(b.a), o->N / o->T, o->c0, o->T, (size_t)o->H * o->W, st);
    case B_DGRAD: {
      if (!bs.at(b.a)) return DVSR_OK;
      dvsr_conv2d_desc g;
      ConvExtra ex;
      dgrad_desc(p, b, P, bs, &g, &ex);
      if (bs.use_v1) return conv2d_run(g, ex, st);
      return conv2d_packed_run(g, bs.dpack + o->dpk_off[b.which], ex, o->dgeo[b.which], st);
    }
    case B_REDUCE: {
      float* dst = bs.at(b.a);
      if (!dst) return DVSR_OK;
      return reduce_frames(dst, b.bs, bs.at(b.b), b.B, b.cnt, b.per, b.accum, st);
    }
    case B_DCN: {
      const size_t P_ = (size_t)o->H * o->W;
      const long long bstride = (long long)o->dg * 27 * P_;
      const float* om = bs.arena + o->x1.off;
      float* gom = bs.at(b.c);
      return mdcn_backward_run(bs.arena + o->x0.off, om, bstride, om + (size_t)o->dg * 18 * P_, bstride, 1, P[o->pw],
                               bs.at(b.b), bs.at(b.a), gom, bstride, gom + (size_t)o->dg * 18 * P_, bstride,
                               GP[o->pw], GP[o->pb], o->N, o->c0, o->H, o->W, o->Cout, 1, 1, 1, o->dg, scratch,
                               scratch_bytes, st, p.wgroups, (long long)o->Cout * o->c0 * 9, o->Cout,
                               p.wsets > 1 ? (long long)o->Cout * o->c0 * 9 : 0);
    }
    case B_UP: {
      float* gx = bs.at(b.a);
      if (!gx) return DVSR_OK;
      return upsample_bilinear_bwd(bs.at(b.b), gx, o->planes, o->H, o->W, o->S, o->mul, b.accum, st);
    }
    case B_POOL:
      return pool3s2_bwd(bs.arena + o->x0.off, bs.at(b.b), bs.at(b.c), bs.at(b.a), o->planes, o->H, o->W, b.accum, st);
    case B_GATE:
      return tsa_gate_bwd(bs.arena + o->x0.off, bs.arena + o->x1.off, bs.arena + o->res.off, bs.arena + o->y.off,
                          bs.at(b.b), bs.at(b.a), bs.at(b.c), bs.at(b.d), o->gB, o->gN, o->gC, o->gHW, st,
                          bs.garena + o->y.off);  // cor's (otherwise unused) gradient slot holds g_dot
    case B_BLEND:
      return tsa_blend_bwd(bs.arena + o->x0.off, bs.arena + o->x1.off, bs.at(b.b), bs.at(b.a), bs.at(b.c),
                           o->y.numel, b.accum, st);
  }
  return DVSR_ERR_INVALID;
}

struct Bases {
  float* arena; const float* x; float* out; bool use_v1;
  bool nograd = false;   // the workspace has no gradient region: the forward may run the no-grad geometries (Op::geo_ng)
  float* at(const T& t) const {
    if (t.space == SP_ARENA) return arena + t.off;
    if (t.space == SP_INPUT) return const_cast<float*>(x) + t.off;
    if (t.space == SP_OUTPUT) return out + t.off;
    return nullptr;
  }
};

// The weight-gradient side streams: a small pool per device for the whole process (never destroyed).  Plans on different
// launch streams share it: their weight gradients then queue behind each other, ordered by the plans' own fork / join events.
// Candidate 0 is THE side stream whenever it runs beside the launch stream; the others exist only for launch streams it
// shares a hardware queue with (side_stream_for).
constexpr int SIDE_POOL = 4;
static hipStream_t pool_side_stream(int i) {
  static std::mutex mu;
  static hipStream_t streams[64][SIDE_POOL] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || i < 0 || i >= SIDE_POOL) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  if (!streams[dev][i] && hipStreamCreateWithFlags(&streams[dev][i], hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    streams[dev][i] = nullptr;
  }
  return streams[dev][i];
}
static hipStream_t shared_side_stream() { return pool_side_stream(0); }

// ---- does work on the side stream run BESIDE work on a launch stream?
// ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams that land on one queue run
// behind each other: a plan whose weight gradients sit on such a side stream pays the fork / join events and gets no
// overlap (the inner step measured 10.0 instead of 8.2 ms with an RCCL communicator's streams in the process, r02-r04).
// Which queue a stream gets depends on every stream the process created before -- nothing a plan can know -- so it is
// MEASURED, once per (device, launch stream): a kernel that spins for 150 us on the launch stream, a marker kernel on the
// side stream behind it; the marker's start time tells whether it waited for the spinner.  dvsr_edvr_backward falls back to
// single-stream weight gradients for a launch stream that fails the probe.  DVSR_BWD_PROBE=0 skips it (assume overlap).
__global__ void probe_spin_kernel(long long* out, long long ticks) {
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  long long t = t0;
  while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(8); t = __builtin_amdgcn_s_memrealtime(); }
  out[0] = t0;
  out[1] = t;
}
__global__ void probe_mark_kernel(long long* out) { out[2] = __builtin_amdgcn_s_memrealtime(); }

// 1: overlaps, 0: serialised, -1: could not tell (capturing, allocation failure, probe disabled by the caller)
static int probe_side_overlap(hipStream_t st, hipStream_t side) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return -1; }
  long long* d = nullptr;
  if (hipMalloc(&d, 4 * sizeof(long long)) != hipSuccess) { (void)hipGetLastError(); return -1; }
  int res = -1;
  long long h[4] = {0, 0, 0, 0};
  // (the streams are drained first: what is still queued on either would be measured instead)
  if (hipStreamSynchronize(st) == hipSuccess && hipStreamSynchronize(side) == hipSuccess &&
      hipMemsetAsync(d, 0, sizeof(h), st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
    // (one untimed marker first: the first launch on a fresh stream costs its queue set-up, > 100 us at times -- timed, that is a
    // false "serialised")
    hipLaunchKernelGGL(probe_mark_kernel, dim3(1), dim3(64), 0, side, d);
    (void)hipStreamSynchronize(side);
    (void)hipMemsetAsync(d, 0, sizeof(h), st);
    (void)hipStreamSynchronize(st);
    hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, st, d, 15000LL);   // 150 us of the 100 MHz counter
    hipLaunchKernelGGL(probe_mark_kernel, dim3(1), dim3(64), 0, side, d);
    if (hipStreamSynchronize(side) == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[1] > h[0] && h[2] > 0)
      res = h[2] < h[1] - 5000 ? 1 : 0;   // the marker started at least 50 us before the spinner ended
  }
  (void)hipGetLastError();
  (void)hipFree(d);
  return res;
}

// The side stream to use beside launch stream `st`: the first of the pool that the probe finds running concurrently with
// it (a new stream lands on another hardware queue than its predecessor, so one of four consecutive candidates is off the
// launch stream's queue unless everything is on one), cached per (device, launch stream).  nullptr: none overlaps -- the
// caller keeps its weight gradients on `st`.  *known = false: the probe could not run (stream capture): candidate 0, unprobed.
static hipStream_t side_stream_for(hipStream_t st, bool* known = nullptr) {
  static const bool probe_on = [] { const char* v = getenv("DVSR_BWD_PROBE"); return !(v && v[0] == '0'); }();
  if (known) *known = probe_on;
  if (!probe_on) return shared_side_stream();
  // (a NEGATIVE answer is not kept for ever: the stream -> hardware-queue mapping moves as the process creates streams, and a
  // destroyed stream's handle can come back as another stream -- it is asked again every 64th use)
  struct Entry { int dev; hipStream_t st, side; int uses; };
  static std::mutex mu;
  static std::vector<Entry> cache;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  for (size_t i = 0; i < cache.size(); ++i) {
    Entry& e = cache[i];
    if (e.dev != dev || e.st != st) continue;
    if (e.side || ++e.uses < 64) return e.side;
    cache.erase(cache.begin() + i);   // re-probe below
    break;
  }
  for (int i = 0; i < SIDE_POOL; ++i) {
    hipStream_t cand = pool_side_stream(i);
    if (!cand) break;
    const int r = probe_side_overlap(st, cand);
    if (r < 0) {   // (capturing: no measurement possible, nothing cached)
      if (known) *known = false;
      return shared_side_stream();
    }
    if (r == 1) {
      cache.push_back({dev, st, cand, 0});
      return cand;
    }
  }
  cache.push_back({dev, st, nullptr, 0});
  return nullptr;
}

// 1: a side stream of the pool runs beside `stream` on this device (the plans fork their weight gradients onto it), 0: none
// does (they stay on `stream`), -1: unknown (see probe_side_overlap)
extern "C" int dvsr_side_stream_overlaps(dvsr_stream_t stream) {
  bool known = true;
  hipStream_t side = side_stream_for((hipStream_t)stream, &known);
  if (!known) return -1;
  return side ? 1 : 0;
}

// Packs the weights of every conv of the tape (forward: wt=0; backward: the two transposed views).
static int pack_all(const dvsr_edvr_plan& p, const float* const* P, float* arena_base, float* fwd_base,
                    float* bwd_base, hipStream_t st, bool nograd = false) {
  PackTable t;
  t.n = 0;
  auto flush = [&]() { int rc = pack_weights_run(t, st); t.n = 0; return rc; };
  const int S = p.wsets;   // weight sets: params are [S][numel], every pack slot holds S consecutive packs
  for (const Op& o : p.ops) {
    if (o.type == OP_DCN && fwd_base && o.wp_floats) {
      for (int ws = 0; ws < S; ++ws) {
        PackEntry& e = t.e[t.n++];
        e = PackEntry{};   // (slots are reused after a flush: no field may keep the previous occupant's value -- perm!)
        e.w = P[o.pw] + (size_t)ws * o.Cout * o.c0 * 9; e.P = fwd_base + o.wp_off + (size_t)ws * o.wp_floats;
        e.Cout = o.Cout; e.Ctot = o.c0; e.KK = 9; e.CC = 8; e.wt = 0;
        e.w_ctot = 0; e.w_coff = 0; e.ncb = ceil_div(o.Cout, 64); e.nchunks = o.c0 / 8; e.bf = 0; e.perm = mdcn_pack_perm(o.W);
        e.pch = e.perm == 6 ? mdcn_pack_floats() : conv2_pch(3, 1);   // (the chunk pitch of the layout; the slot fits either)
        if (t.n == 48) { int rc = flush(); if (rc) return rc; }
      }
    }
    if (o.type != OP_CONV) continue;
    const int ctot = o.c0 + o.c1, KK = o.ks * o.ks;
    const size_t wnum = (size_t)o.Cout * ctot * KK;
    const float* wsrc = P[o.pw];
    if (o.wmap) {  // the re-laid-out copy is (re)built by the forward; the backward packs read the same slot
      wsrc = arena_base + o.w2_off;
      if (fwd_base) {   // (the map is per output channel: S stacked sets are one [S * Cout] tensor)
        int rc = w4_to_s2d(P[o.pw], arena_base + o.w2_off, S * o.Cout, ctot / 4, 0, st);
        if (rc) return rc;
      }
    }
    for (int ws = 0; ws < S; ++ws) {
      if (fwd_base) {
        PackEntry& e = t.e[t.n++];
        e = PackEntry{};
        const ConvGeo& fg = nograd ? o.geo_ng : o.geo;
        e.w = wsrc + ws * wnum; e.Cout = o.Cout; e.Ctot = ctot; e.KK = KK;
        e.P = nograd ? fwd_base + o.wp_ng_off + (size_t)ws * o.wp_ng_floats : fwd_base + o.wp_off + (size_t)ws * o.wp_floats;
        e.CC = fg.cc; e.wt = 0; e.w_ctot = 0; e.w_coff = 0; e.ncb = ceil_div(o.Cout, 64);
        e.nchunks = ceil_div(ctot, e.CC); e.bf = fg.bf; e.perm = fg.dma; e.pch = conv2_pch_cc(o.ks, e.CC, e.bf, fg.dma);
        if (t.n == 48) { int rc = flush(); if (rc) return rc; }
      }
      if (bwd_base) {
        for (int which = 0; which < 2; ++which) {
          const int ci = which ? o.c1 : o.c0;
          if (!ci) continue;
          PackEntry& e = t.e[t.n++];
          e = PackEntry{};
          e.w = wsrc + ws * wnum; e.P = bwd_base + o.dpk_off[which] + (size_t)ws * o.dpk_floats[which]; e.Cout = ci; e.Ctot = o.Cout;
          e.KK = KK;
          e.CC = o.dgeo[which].cc; e.wt = 1; e.w_ctot = ctot; e.w_coff = which ? o.c0 : 0;
          e.ncb = ceil_div(ci, 64); e.nchunks = ceil_div(o.Cout, e.CC); e.bf = o.dgeo[which].bf;
          e.pch = conv2_pch_cc(o.ks, e.CC, e.bf, o.dgeo[which].dma); e.perm = o.dgeo[which].dma;
          if (t.n == 48) { int rc = flush(); if (rc) return rc; }
        }
      }
    }
  }
  return flush();
}

static int run_forward_op(const dvsr_edvr_plan& p, const Op& o, const float* const* P, const Bases& bs, hipStream_t st) {
  switch (o.type) {
    case OP_CONV: {
      dvsr_conv2d_desc d;
      d.x0 = bs.at(o.x0); d.x1 = bs.at(o.x1); d.w = P[o.pw]; d.bias = P[o.pb]; d.res = bs.at(o.res);
      d.y = bs.at(o.y);
      d.N = o.N; d.c0 = o.c0; d.c1 = o.c1; d.H = o.H; d.W = o.W; d.Cout = o.Cout; d.ks = o.ks;
      d.stride = o.stride; d.pad = conv_pad(o); d.act = o.act; d.pixel_shuffle = o.ps;
      if (o.pad_out) { d.y = bs.at(o.ypad); d.pixel_shuffle = o.pad_out; }
      if (!bs.use_v1 && (o.dual >= 0 || o.fused_into >= 0)) {   // the pair of 1x1 convs over one input (see Op::dual)
        const Op& oa = o.dual >= 0 ? o : p.ops[o.fused_into];
        const Op& ob = p.ops[oa.dual];
        const long long HW = (long long)oa.H * oa.W;
        // (what conv1x1_dual_run assumes of BOTH ops, checked where it is launched -- not only where the pair is marked: two plain
        // 1x1 convs of 64 outputs over ONE single-pointer input, same activation, nothing fused into their stores)
        auto plain1x1 = [](const Op& q) {
          return q.type == OP_CONV && q.ks == 1 && q.stride == 1 && q.c1 == 0 && q.Cout == 64 && !q.res.valid() && !q.ps && !q.wmap && !q.pad_out;
        };
        const bool pair_ok = plain1x1(oa) && plain1x1(ob) && oa.act == ob.act && oa.c0 == ob.c0 && oa.N == ob.N && oa.H == ob.H &&
                             oa.W == ob.W && oa.x0.space == ob.x0.space && oa.x0.off == ob.x0.off;
        if (pair_ok && conv1x1_dual_ok(bs.at(oa.x0), P[oa.pw], P[ob.pw], bs.at(oa.y), bs.at(ob.y), oa.c0, HW)) {
          if (o.fused_into >= 0) return DVSR_OK;   // the partner's launch wrote this op's output
          return conv1x1_dual_run(bs.at(oa.x0), P[oa.pw], P[oa.pb], P[ob.pw], P[ob.pb], bs.at(oa.y), bs.at(ob.y), oa.N, oa.c0,
                                  (int)HW, oa.act, st, p.wsets > 1 ? oa.N / p.wsets : 1,
                                  p.wsets > 1 ? (long long)oa.Cout * oa.c0 : 0, p.wsets > 1 ? oa.Cout : 0);
        }
      }
      if (o.wmap) d.w = bs.arena + o.w2_off;
      d.x1_bdiv = o.x1_bdiv; d.x0_bstride = o.x0_bs; d.x1_bstride = o.x1_bs;
      if (bs.use_v1) return conv2d_run(d, ConvExtra(), st);
      if (o.Cout <= 4 && o.ks == 3 && o.stride == 1 && !o.c1 && !o.ps && !o.x0_bs && o.pad < 0)  // conv_last
        return conv3x3_small_cout_run(d.x0, d.w, d.bias, d.res, d.y, o.N, o.c0, o.H, o.W, o.Cout, o.act, st,
                                      p.wsets > 1 ? o.N / p.wsets : 1, p.wsets > 1 ? (long long)o.Cout * o.c0 * 9 : 0,
                                      p.wsets > 1 ? o.Cout : 0);
      ConvExtra ex;
      if (bs.nograd) {
        set_wsets(p, o.N, o.wp_ng_floats, o.Cout, &ex);
        return conv2d_packed_run(d, bs.arena + o.wp_ng_off, ex, o.geo_ng, st);
      }
      set_wsets(p, o.N, o.wp_floats, o.Cout, &ex);
      return conv2d_packed_run(d, bs.arena + o.wp_off, ex, o.geo, st);
    }
    case OP_DCN: {
      const float* om = bs.at(o.x1);
      const long long bstride = (long long)o.dg * 27 * o.H * o.W;
      if (!bs.use_v1 && o.wp_floats)
        return mdcn_forward_packed_run(bs.at(o.x0), om, bstride, om + (size_t)o.dg * 18 * o.H * o.W, bstride, 1,
                                       bs.arena + o.wp_off, P[o.pb], bs.at(o.y), o.N, o.c0, o.H, o.W, o.Cout,
                                       o.dg, o.act, st, p.wsets > 1 ? o.N / p.wsets : 1,
                                       p.wsets > 1 ? (long long)o.wp_floats : 0, p.wsets > 1 ? o.Cout : 0, mdcn_pack_perm(o.W));
      return mdcn_forward_run(bs.at(o.x0), om, bstride, om + (size_t)o.dg * 18 * o.H * o.W, bstride, 1,
                              P[o.pw], P[o.pb], bs.at(o.y), o.N, o.c0, o.H, o.W, o.Cout, 3, 3, 1, 1,
                              1, 1, o.dg, o.act, st);
    }
    case OP_UP:
      return upsample_bilinear_fwd(bs.at(o.x0), bs.at(o.y), o.planes, o.H, o.W, o.S, o.mul, st);
    case OP_POOL:
      return pool3s2_fwd(bs.at(o.x0), bs.at(o.y), bs.at(o.y2), o.planes, o.H, o.W, st);
    case OP_GATE:
      return tsa_gate_fwd(bs.at(o.x0), bs.at(o.x1), bs.at(o.res), bs.at(o.y), bs.at(o.y2), o.gB, o.gN,
                          o.gC, o.gHW, st);
    case OP_BLEND:
      return tsa_blend_fwd(bs.at(o.x0), bs.at(o.x1), bs.at(o.res), bs.at(o.y), o.y.numel, st);
    case OP_PAD:
      if (o.fused_into >= 0) return DVSR_OK;   // the producing conv stored the padded tensor itself
      return pad_fwd(bs.at(o.x0), bs.at(o.y), o.pmode, o.N, o.c0, o.H, o.W, o.T, st);
    case OP_MEANSUB:
      return meansub_fwd(bs.at(o.x0), bs.at(o.y), bs.at(o.y2), bs.at(o.res), o.N / o.T, o.c0, o.T, o.H, o.W, st);
    case OP_ADDMEAN:
      return addmean_fwd(bs.at(o.x0), bs.at(o.x1), bs.at(o.y), o.N / o.T, o.c0, o.T, (size_t)o.H * o.W, st);
    case OP_ADD:
      return add_out(bs.at(o.y), bs.at(o.x0), bs.at(o.x1), o.y.numel, st);
  }
  return DVSR_ERR_INVALID;
}

}  // namespace dvsr

using namespace dvsr;

extern "C" int dvsr_edvr_plan_create(const dvsr_edvr_config* cfg, int B, int H, int W,
                                     dvsr_edvr_plan** out) {
  return dvsr_edvr_plan_create_grouped(cfg, B, H, W, 1, out);
}

extern "C" int dvsr_edvr_plan_create_grouped(const dvsr_edvr_config* cfg, int B, int H, int W, int grad_groups,
                                             dvsr_edvr_plan** out) {
  return dvsr_edvr_plan_create_ex(cfg, B, H, W, grad_groups, 1, out);
}

extern "C" int dvsr_edvr_plan_create_ex(const dvsr_edvr_config* cfg, int B, int H, int W, int grad_groups,
                                        int weight_sets, dvsr_edvr_plan** out) {
  DVSR_REQUIRE(cfg && out, DVSR_ERR_INVALID, "edvr_plan_create: null argument");
  DVSR_REQUIRE(weight_sets == 1 || weight_sets == grad_groups, DVSR_ERR_INVALID,
               "edvr_plan_create: weight_sets=%d must be 1 or grad_groups=%d", weight_sets, grad_groups);
  DVSR_REQUIRE(B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, DVSR_ERR_INVALID,
               "edvr_plan_create: B=%d H=%d W=%d (H, W must be positive multiples of 4)", B, H, W);
  DVSR_REQUIRE(grad_groups >= 1 && B % grad_groups == 0, DVSR_ERR_INVALID,
               "edvr_plan_create: grad_groups=%d must divide the batch B=%d", grad_groups, B);
  DVSR_REQUIRE(cfg->nf > 0 && cfg->nf % cfg->groups == 0 && cfg->nframes > 0 && cfg->front_RBs >= 0 &&
                   cfg->back_RBs >= 0, DVSR_ERR_INVALID, "edvr_plan_create: bad network config");
  DVSR_REQUIRE(cfg->scale == 4 || cfg->scale == 2, DVSR_ERR_UNSUPPORTED,
               "edvr_plan_create: scale=%d (2 or 4)", cfg->scale);
  DVSR_REQUIRE(cfg->center >= 0 && cfg->center < cfg->nframes, DVSR_ERR_INVALID,
               "edvr_plan_create: center=%d out of range", cfg->center);
  const int cpg = cfg->nf / cfg->groups;
  DVSR_REQUIRE(cpg == 4 || cpg == 8 || cpg == 16, DVSR_ERR_UNSUPPORTED,
               "edvr_plan_create: nf/groups=%d (supported: 4, 8, 16)", cpg);
  dvsr_edvr_plan* p = new dvsr_edvr_plan();
  p->cfg = *cfg; p->B = B; p->H = H; p->W = W; p->wgroups = grad_groups; p->wsets = weight_sets;
  { const char* v = getenv("DVSR_CONV_V1"); p->use_v1 = v && v[0] == '1'; }
  if (weight_sets > 1 && (p->use_v1 || cpg % 8 != 0)) {
    delete p;
    DVSR_REQUIRE(false, DVSR_ERR_UNSUPPORTED, "edvr_plan_create: per-sample weight sets need the packed kernels (no "
                 "DVSR_CONV_V1) and nf/groups a multiple of 8");
  }
  { const char* v = getenv("DVSR_BWD_STREAMS"); p->side_streams = (v && v[0] == '0') ? 0 : 1; }
  int rc = build_plan(*p);
  if (rc != DVSR_OK) { delete p; return rc; }
  build_backward(*p);
  *out = p;
  return DVSR_OK;
}

extern "C" void dvsr_edvr_plan_destroy(dvsr_edvr_plan* p) {
  if (!p) return;
  if (p->side) (void)hipStreamSynchronize(p->side);   // (the side streams are process-wide, per device: not destroyed)
  if (p->ev_fork) {
    (void)hipEventDestroy(p->ev_fork);
    (void)hipEventDestroy(p->ev_join);
  }
  delete p;
}

extern "C" int dvsr_edvr_num_params(const dvsr_edvr_plan* p) { return p ? p->n_params : -1; }

extern "C" int dvsr_edvr_num_launches(const dvsr_edvr_plan* p) { return p ? (int)p->ops.size() : -1; }

// need_grad = 0: activation arena only.  need_grad = 1: + gradient arena + staging + scratch,
// laid out [activations | gradients | dgrad staging | packed dgrad weights | wgrad partials / DCN columns].
extern "C" size_t dvsr_edvr_workspace_bytes(const dvsr_edvr_plan* p, int need_grad) {
  if (!p) return 0;
  size_t b = p->arena_floats * sizeof(float);
  if (need_grad)
    b += (p->arena_floats + p->tmp_floats + p->dpack_floats) * sizeof(float) + p->scratch_bytes + p->wscratch_bytes;
  return b;
}

extern "C" int dvsr_edvr_backward(const dvsr_edvr_plan* p, const float* const* params, const float* x,
                                  const float* grad_out, float* const* grad_params, float* grad_x, void* ws,
                                  size_t ws_bytes, dvsr_stream_t stream) {
  DVSR_REQUIRE(p && params && x && grad_out && grad_params && ws, DVSR_ERR_INVALID, "edvr_backward: null argument");
  DVSR_REQUIRE(!p->bwd_unsupported, DVSR_ERR_UNSUPPORTED, "edvr_backward: the tape holds %s", p->bwd_unsupported);
  DVSR_REQUIRE(ws_bytes >= dvsr_edvr_workspace_bytes(p, 1), DVSR_ERR_WORKSPACE,
               "edvr_backward: workspace %zu < %zu bytes (allocate with need_grad=1 BEFORE the forward)", ws_bytes,
               dvsr_edvr_workspace_bytes(p, 1));
  hipStream_t st = (hipStream_t)stream;
  BBases bs;
  bs.arena = (float*)ws; bs.garena = bs.arena + p->arena_floats; bs.x = x; bs.gx = grad_x; bs.gout = grad_out;
  bs.tmp = bs.garena + p->arena_floats;
  bs.dpack = bs.tmp + p->tmp_floats;
  bs.use_v1 = p->use_v1;
  void* scratch = bs.dpack + p->dpack_floats;
  if (!p->use_v1) {
    int rc = pack_all(*p, params, bs.arena, nullptr, bs.dpack, st);
    if (rc != DVSR_OK) return rc;
  }
  if (grad_x) {
    const size_t n = (size_t)p->B * p->cfg.nframes * 3 * p->H * p->W;
    DVSR_REQUIRE(hipMemsetAsync(grad_x, 0, n * sizeof(float), st) == hipSuccess, DVSR_ERR_HIP,
                 "edvr_backward: memset of grad_x failed");
  }
  // fork/join state of the side stream (events created on first use, owned by the plan)
  bool use_side = p->side_streams != 0;
  if (use_side && !p->ev_fork) {
    // Side streams are per DEVICE, shared by all plans (pool_side_stream): ROCm maps streams onto a few hardware queues
    // (GPU_MAX_HW_QUEUES, 4 by default; more than ~6 in use and the command processor time-slices them: EDVR-L
    // forward+backward 20 -> 31 ms), and streams that share a queue run behind each other.  With a stream per plan the
    // mapping, and with it the overlap, depended on how many plans and other streams (an RCCL communicator holds some)
    // the process had created before: the inner step measured 8.2 or 10.0 ms, EDVR-L fp32 forward+backward 20.1 or
    // 23.5 ms.  dynavsr_amd/_lib.py asks for 6 queues: launch stream, RCCL's, this one, adapt_video's.
    if (hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      p->ev_fork = nullptr;
      use_side = false;
    }
  }
  // ... and WHICH of them runs beside this launch stream is measured (side_stream_for: once per launch stream); none: the
  // fork / join events would only cost, the weight gradients stay on the launch stream
  if (use_side) {
    p->side = side_stream_for(st);
    if (!p->side) use_side = false;
  }
  void* wscratch = (char*)scratch + p->scratch_bytes;
  // the weight-gradient slot regions: zeroed once here, every reduce re-zeroes what it read
  DVSR_REQUIRE(hipMemsetAsync(wscratch, 0, p->wscratch_bytes, st) == hipSuccess, DVSR_ERR_HIP,
               "edvr_backward: memset of the wgrad scratch failed");
  bool forked = false;
  int last_fork_fwd = -1;
  std::vector<WgradReduceEntry> reduces;
  std::vector<const BOp*> unmaps;
  reduces.reserve(p->bops.size());
  // An event recorded on the main stream delays the main stream's NEXT kernel by ~6 us (profiles/r02b timeline),
  // which on the small inner-step clips is a third of a layer: the weight gradients of `fork_every` consecutive
  // layers therefore share one fork (a gradient buffer is never rewritten once its layer's turn has come, so
  // launching a weight gradient a few layers late is safe).  DVSR_BWD_FORK_EVERY=<layers> (1 = a fork per layer).
  static const int fork_env = [] { const char* v = getenv("DVSR_BWD_FORK_EVERY"); return v ? atoi(v) : 0; }();
  const int fork_every = fork_env > 0 ? fork_env : p->fork_every;
  std::vector<WgradLaunch> waiting;
  int waiting_layers = 0;
  auto flush_waiting = [&]() -> int {
    if (waiting.empty()) return DVSR_OK;
    hipStream_t ws = st;
    if (use_side) {
      DVSR_REQUIRE(hipEventRecord(p->ev_fork, st) == hipSuccess && hipStreamWaitEvent(p->side, p->ev_fork, 0) == hipSuccess,
                   DVSR_ERR_HIP, "edvr_backward: fork to the wgrad stream failed");
      ws = p->side;
      forked = true;
    }
    for (const WgradLaunch& l : waiting) {
      int rc = conv2d_wgrad_launch(l, ws);
      if (rc != DVSR_OK) return rc;
    }
    waiting.clear();
    waiting_layers = 0;
    return DVSR_OK;
  };
  for (const BOp& b : p->bops) {
    int rc = DVSR_OK;
    if (b.type == B_WGRAD) {
      reduces.emplace_back();
      WgradLaunch l;
      rc = prep_wgrad(*p, b, grad_params, bs, (char*)wscratch + b.ws_off, b.ws_bytes, st, &reduces.back(), &l);
      if (rc != DVSR_OK) return rc;
      if (b.fwd != last_fork_fwd) {   // gy of this layer is final on `st` at this point of the tape
        if (waiting_layers >= fork_every) rc = flush_waiting();
        last_fork_fwd = b.fwd;
        ++waiting_layers;
      }
      waiting.push_back(l);
      // only SMALL weight gradients wait for company (the small-grid kernel was chosen for them): a large one -- the
      // estimator's at 176x320 run 100-300 us each, longer than its data-gradient chain -- must start at once, or
      // the side stream is still busy long after the main stream has finished (profiles/r02_g_estimator_timeline.txt)
      if (rc == DVSR_OK && (!use_side || !l.kys)) rc = flush_waiting();
    } else if (b.type == B_WUNMAP) {
      unmaps.push_back(&b);  // needs the reduced gradient: after the batched reduce below
    } else {
      rc = run_backward_op(*p, b, params, grad_params, bs, scratch, p->scratch_bytes, st);
    }
    if (rc != DVSR_OK) return rc;
  }
  {
    int rc = flush_waiting();
    if (rc != DVSR_OK) return rc;
    hipStream_t ws = use_side && forked ? p->side : st;
    rc = wgrad_reduce_batch(reduces.data(), (int)reduces.size(), ws);
    if (rc != DVSR_OK) return rc;
    for (const BOp* b : unmaps) {
      rc = run_backward_op(*p, *b, params, grad_params, bs, nullptr, 0, ws);
      if (rc != DVSR_OK) return rc;
    }
  }
  if (forked)
    DVSR_REQUIRE(hipEventRecord(p->ev_join, p->side) == hipSuccess &&
                     hipStreamWaitEvent(st, p->ev_join, 0) == hipSuccess,
                 DVSR_ERR_HIP, "edvr_backward: join of the wgrad stream failed");
  return DVSR_OK;
}

extern "C" int dvsr_edvr_num_backward_launches(const dvsr_edvr_plan* p) { return p ? (int)p->bops.size() : -1; }

static int edvr_forward_impl(const dvsr_edvr_plan* p, const float* const* params, const float* x, float* out, void* ws,
                             size_t ws_bytes, dvsr_stream_t stream, bool packs_valid);

extern "C" int dvsr_edvr_forward(const dvsr_edvr_plan* p, const float* const* params, const float* x,
                                 float* out, void* ws, size_t ws_bytes, dvsr_stream_t stream) {
  return edvr_forward_impl(p, params, x, out, ws, ws_bytes, stream, false);
}

// The forward WITHOUT its weight-packing launches: the workspace still holds the packs an earlier dvsr_edvr_forward of this
// plan left there for the same parameter values (a video's clips through one frozen network: Video_base_model.test() in a
// loop -- the packs are six launches, ~2 % of a 180x320 forward, and a pure function of the weights).
extern "C" int dvsr_edvr_forward_packed(const dvsr_edvr_plan* p, const float* const* params, const float* x,
                                        float* out, void* ws, size_t ws_bytes, dvsr_stream_t stream) {
  return edvr_forward_impl(p, params, x, out, ws, ws_bytes, stream, true);
}

static int edvr_forward_impl(const dvsr_edvr_plan* p, const float* const* params, const float* x, float* out, void* ws,
                             size_t ws_bytes, dvsr_stream_t stream, bool packs_valid) {
  DVSR_REQUIRE(p && params && x && out && ws, DVSR_ERR_INVALID, "edvr_forward: null argument");
  DVSR_REQUIRE(ws_bytes >= p->arena_floats * sizeof(float), DVSR_ERR_WORKSPACE,
               "edvr_forward: workspace %zu < %zu bytes", ws_bytes, p->arena_floats * sizeof(float));
  Bases bs{(float*)ws, x, out, p->use_v1};
  // a workspace too small for dvsr_edvr_backward is a no-grad forward (the header's contract: "allocate with need_grad = 1
  // BEFORE the forward"): only then may a layer run on a kernel the backward's tape was not built around
  bs.nograd = ws_bytes < dvsr_edvr_workspace_bytes(p, 1);
  if (!p->use_v1 && !packs_valid) {
    int rc = pack_all(*p, params, bs.arena, bs.arena, nullptr, (hipStream_t)stream, bs.nograd);
    if (rc != DVSR_OK) return rc;
  }
  for (const Op& o : p->ops) {
    int rc = run_forward_op(*p, o, params, bs, (hipStream_t)stream);
    if (rc != DVSR_OK) return rc;
  }
  return DVSR_OK;
}

// Algorithmic work of one launch: FLOPs = 2*MAC of the contraction (+ the bilinear blends for the
// DCN sampler); bytes = every distinct input read once + every output written once (fp32).
static void op_work(const Op& o, const char** kind, double* flops, double* bytes) {
  *flops = 0; *bytes = 0; *kind = "other";
  switch (o.type) {
    case OP_CONV: {
      const int Ho = conv_out(o, o.H), Wo = conv_out(o, o.W);
      const double ctot = o.c0 + o.c1, px = (double)o.N * Ho * Wo;
      *flops = 2.0 * px * o.Cout * ctot * o.ks * o.ks;
      *bytes = 4.0 * ((double)o.N * o.c0 * o.H * o.W + (double)(o.N / o.x1_bdiv) * o.c1 * o.H * o.W +
                      px * o.Cout * (o.res.valid() ? 2 : 1) + (double)o.Cout * ctot * o.ks * o.ks);
      *kind = o.ks == 1 ? "conv1x1" : (o.ks == 2 ? "conv2x2" : (o.stride == 2 ? "conv3x3s2" : "conv3x3s1"));
      break;
    }
    case OP_DCN: {
      const double px = (double)o.N * o.H * o.W;
      *flops = 2.0 * px * o.Cout * o.c0 * 9 + px * o.c0 * 9 * 8.0;
      *bytes = 4.0 * (px * o.c0 + px * o.dg * 27 + px * o.Cout + (double)o.Cout * o.c0 * 9);
      *kind = "mdcn";
      break;
    }
    case OP_UP: *bytes = 4.0 * o.planes * o.H * o.W * (1.0 + o.S * o.S); *kind = "upsample"; break;
    case OP_POOL: *bytes = 4.0 * o.planes * o.H * o.W * 1.5; *kind = "pool"; break;
    case OP_GATE: *bytes = 4.0 * ((double)o.gB * o.gN * o.gC * o.gHW * 3 + (double)o.gB * o.gC * o.gHW +
                                 (double)o.gB * o.gN * o.gHW); *kind = "tsa_gate"; break;
    case OP_BLEND: *bytes = 4.0 * o.y.numel * 4; *kind = "tsa_blend"; break;
    case OP_ADD: *bytes = 4.0 * o.y.numel * 3; *kind = "add"; break;
    case OP_PAD: *bytes = 4.0 * ((double)o.N * o.c0 * o.H * o.W + (double)o.y.numel); *kind = "pad"; break;
    case OP_MEANSUB: *bytes = 4.0 * o.y.numel * 3; *kind = "meansub"; break;
    case OP_ADDMEAN: *bytes = 4.0 * o.y.numel * 2; *kind = "addmean"; break;
  }
}

extern "C" int dvsr_edvr_op_info(const dvsr_edvr_plan* p, int index, char* kind, int kind_cap,
                                 char* name, int name_cap, double* flops, double* bytes) {
  DVSR_REQUIRE(p && kind && name && flops && bytes && index >= 0 && index < (int)p->ops.size(),
               DVSR_ERR_INVALID, "edvr_op_info: bad argument");
  const char* k;
  op_work(p->ops[index], &k, flops, bytes);
  snprintf(kind, kind_cap, "%s", k);
  if (p->ops[index].type == OP_CONV)
  {
    // (the geometry of the NO-GRAD forward -- what dvsr_edvr_forward_timed's workspace runs; a training tape's forward runs
    // Op::geo, which differs only where the tag ends in "w5": those layers are "w3" there)
    const ConvGeo& g = p->ops[index].geo_ng;
    snprintf(name, name_cap, "%s[%d/%d/%d%s]", p->ops[index].name, g.cc, g.th, g.mt,
             g.dma == 5 ? "w5" : (g.dma == 4 ? "w3" : (g.dma == 3 ? "w" : (g.dma ? "d" : ""))));
  }
  else
    snprintf(name, name_cap, "%s", p->ops[index].name);
  return DVSR_OK;
}

// Contraction work of a whole plan, forward and backward tapes (out: NINE doubles): out[0] / out[2] = algorithmic FLOPs (2 x MACs of the direct
// sums: convolutions and the DCN contraction; weight + data gradients for the backward), out[1] / out[3] = the same work as
// the kernels shape it, in fp32 products -- launches on the Winograd F(2x2, 3x3) kernels (geo.dma 3, 4) do 16/36 of theirs, on
// the F(4x4, 3x3) kernel (geo.dma 5) 36/144;
// out[4] = algorithmic bytes of the forward tape.  out[5] / out[7] = FLOPs ISSUED to the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32), out[6] / out[8] = FLOPs ISSUED to the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) by the forward /
// backward tape: a launch on the exact 3-way operand split issues SIX bf16 products per fp32 product (geo.bf == 2, the
// Winograd bf16x3 kernels geo.dma == 4 / 5, the split3 weight gradient), a plain bf16 launch (geo.bf == 1) one.  bench.py prices
// every roofline fraction with the issued figures against the peak of the pipe they were issued to.
static int plan_work(const dvsr_edvr_plan* p, double* out9, bool nograd) {
  DVSR_REQUIRE(p && out9, DVSR_ERR_INVALID, "edvr_plan_work: null argument");
  double fa = 0, fe = 0, ba = 0, be = 0, fby = 0, f32p[2] = {0, 0}, bfp[2] = {0, 0};
  auto conv_part = [](const Op& o, int ci) {
    return 2.0 * (double)o.N * conv_out(o, o.H) * conv_out(o, o.W) * o.Cout * ci * o.ks * o.ks;
  };
  // one conv-shaped launch: f algorithmic FLOPs on geometry g -> (fp32 products done, pipe they go to)
  auto issue = [&](const ConvGeo& g, double f, double* ex, int pass) {
    // (F(2x2, 3x3): 16 multiplies per 2x2 outputs instead of 36; F(4x4, 3x3), geo.dma == 5: 36 per 4x4 outputs instead of 144)
    const double shaped = g.dma == 5 ? f * 0.25 : (g.dma >= 3 ? f * (16.0 / 36.0) : f);
    *ex += shaped;
    if (g.dma == 4 || g.dma == 5 || g.bf == 2) bfp[pass] += 6.0 * shaped;
    else if (g.bf == 1) bfp[pass] += shaped;
    else f32p[pass] += shaped;
  };
  for (const Op& o : p->ops) {
    {
      const char* k; double fl, by;
      op_work(o, &k, &fl, &by);
      fby += by;
    }
    if (o.type == OP_CONV) {
      const double f = conv_part(o, o.c0 + o.c1);
      fa += f; issue(nograd ? o.geo_ng : o.geo, f, &fe, 0);
    } else if (o.type == OP_DCN) {
      const double f = 2.0 * (double)o.N * o.H * o.W * o.Cout * o.c0 * 9;
      fa += f; fe += f; f32p[0] += f;
    }
  }
  for (const BOp& b : p->bops) {
    if (b.fwd < 0) continue;
    const Op& o = p->ops[b.fwd];
    if (b.type == B_WGRAD) {
      const double f = conv_part(o, b.which ? o.c1 : o.c0);
      ba += f; be += f;
      // (conv2d_wgrad_prepare's choice, restated: prep_wgrad's mode, then the kernel's own eligibility)
      const int mode = (p->cfg.bf16_mfma == 1 && !o.wmap) ? 1 : (wgrad_split3_on() ? 2 : 0);
      const int bf = (mode && o.stride == 1 && (o.ks == 3 || (o.ks == 2 && mode == 2))) ? mode : 0;
      if (bf == 2) bfp[1] += 6.0 * f; else if (bf == 1) bfp[1] += f; else f32p[1] += f;
    } else if (b.type == B_DGRAD) {
      const double f = conv_part(o, b.which ? o.c1 : o.c0);
      ba += f; issue(o.dgeo[b.which], f, &be, 1);
    } else if (b.type == B_DCN) {
      const double f = 2.0 * 2.0 * (double)o.N * o.H * o.W * o.Cout * o.c0 * 9;   // dcol + dW
      ba += f; be += f; f32p[1] += f;
    }
  }
  out9[0] = fa; out9[1] = fe; out9[2] = ba; out9[3] = be;
  out9[4] = fby;   // algorithmic bytes of the forward tape: every launch's distinct inputs once + outputs once (op_work)
  out9[5] = f32p[0]; out9[6] = bfp[0]; out9[7] = f32p[1]; out9[8] = bfp[1];
  return DVSR_OK;
}
extern "C" int dvsr_edvr_plan_work(const dvsr_edvr_plan* p, double* out9) { return plan_work(p, out9, false); }
// ... with the forward tape priced as a NO-GRAD forward runs it (Op::geo_ng: the F(4x4, 3x3) kernel where the plan takes it)
extern "C" int dvsr_edvr_plan_work_nograd(const dvsr_edvr_plan* p, double* out9) { return plan_work(p, out9, true); }

// Same launches as dvsr_edvr_forward with a hipEvent recorded on `stream` around every launch;
// synchronises the stream and returns per-launch milliseconds (measurement aid for bench.py).
extern "C" int dvsr_edvr_forward_timed(const dvsr_edvr_plan* p, const float* const* params,
                                       const float* x, float* out, void* ws, size_t ws_bytes,
                                       dvsr_stream_t stream, float* op_ms) {
  DVSR_REQUIRE(p && params && x && out && ws && op_ms, DVSR_ERR_INVALID, "edvr_forward_timed: null argument");
  DVSR_REQUIRE(ws_bytes >= p->arena_floats * sizeof(float), DVSR_ERR_WORKSPACE,
               "edvr_forward_timed: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t n = p->ops.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) DVSR_REQUIRE(hipEventCreate(&e) == hipSuccess, DVSR_ERR_HIP, "hipEventCreate failed");
  Bases bs{(float*)ws, x, out, p->use_v1};
  bs.nograd = ws_bytes < dvsr_edvr_workspace_bytes(p, 1);
  int rc = p->use_v1 ? DVSR_OK : pack_all(*p, params, bs.arena, bs.arena, nullptr, st, bs.nograd);
  hipEventRecord(ev[0], st);
  for (size_t i = 0; i < n && rc == DVSR_OK; ++i) {
    rc = run_forward_op(*p, p->ops[i], params, bs, st);
    hipEventRecord(ev[i + 1], st);
  }
  if (rc == DVSR_OK && hipStreamSynchronize(st) != hipSuccess) {
    set_error("edvr_forward_timed: stream synchronize failed");
    rc = DVSR_ERR_HIP;
  }
  if (rc == DVSR_OK)
    for (size_t i = 0; i < n; ++i) hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]);
  for (auto& e : ev) hipEventDestroy(e);
  return rc;
}

// Where launch `index` of the forward tape leaves its result: which = 0 the output, 1 the second output of the two-output
// launches (pool: the average plane; TSA gate: the gated features).  *in_arena = 1: offset_floats counts from the start of
// the workspace; 0: the launch writes the network's output tensor (conv_last).  Test aid: every saved activation of a
// need_grad forward can be read back, launch by launch (tests/test_gpu_edvr.py: the kink-free gradient check).
extern "C" int dvsr_edvr_op_output(const dvsr_edvr_plan* p, int index, int which, int* in_arena, long long* offset_floats,
                                   long long* numel) {
  DVSR_REQUIRE(p && in_arena && offset_floats && numel && index >= 0 && index < (int)p->ops.size() && (which == 0 || which == 1),
               DVSR_ERR_INVALID, "edvr_op_output: bad argument");
  const T& t = which ? p->ops[index].y2 : p->ops[index].y;
  DVSR_REQUIRE(t.valid(), DVSR_ERR_INVALID, "edvr_op_output: launch %d has no output %d", index, which);
  *in_arena = t.space == SP_ARENA;
  *offset_floats = (long long)t.off;
  *numel = (long long)t.numel;
  return DVSR_OK;
}

extern "C" int dvsr_edvr_tensor_info(const dvsr_edvr_plan* p, const char* name, long long* offset_floats,
                                     long long* numel) {
  DVSR_REQUIRE(p && name && offset_floats && numel, DVSR_ERR_INVALID, "edvr_tensor_info: null argument");
  for (const auto& kv : p->named)
    if (kv.first == name && kv.second.space == SP_ARENA) {
      *offset_floats = (long long)kv.second.off;
      *numel = (long long)kv.second.numel;
      return DVSR_OK;
    }
  set_error("edvr_tensor_info: no tensor named '%s'", name);
  return DVSR_ERR_INVALID;
}


// =================================================================================================
// Down-scaling estimators MFDN / SFDN (LRimg_estimator.py:38-117) on the same tape machinery.
// Every convolution runs on the dense stride-1 MFMA kernels over explicitly padded tensors
// (pad.hip): ReflectionPad2d(1) -> PAD_REFLECT, the 4x4 stride-2 convs -> PAD_REFLECT_S2D + a 2x2
// conv over 4C channels, Conv3d(k3) over ReplicationPad3d(1) -> PAD_REPL_T3 + a 3x3 conv over 3C
// channels (the Conv3d weight [Cout][C][3][3][3] IS the [Cout][3C][3][3] tensor that needs).
// Input x: [B][in_nc][T][H][W] (T = 1 for SFDN, i.e. [B][in_nc][H][W]); output [B][in_nc][T][H/s][W/s].
// Parameters in state-dict order conv0.weight, conv0.bias, ... conv6.bias.
// =================================================================================================
struct dvsr_estimator_plan {
  dvsr_edvr_plan core;
  dvsr_estimator_config ecfg;
};

namespace dvsr {

static int build_estimator(dvsr_estimator_plan& ep) {
  dvsr_edvr_plan& p = ep.core;
  const dvsr_estimator_config& c = ep.ecfg;
  const int video = c.kind == DVSR_ESTIMATOR_MFDN;
  const int B = p.B, Tn = video ? c.nframes : 1, BT = B * Tn, nf = c.nf, ic = c.in_nc;
  int H = p.H, W = p.W;
  Builder b(p);
  const int L = ACT_LRELU, NO = ACT_NONE;
  T none;
  T xin; xin.space = SP_INPUT; xin.off = 0; xin.numel = (size_t)BT * ic * H * W;
  // x - mean, frames become the batch axis
  T xm = b.alloc("xm", xin.numel);
  T mean = b.alloc("mean", (size_t)BT * ic);
  {
    Op o; o.type = OP_MEANSUB; o.name = "meansub"; o.x0 = xin; o.y = xm; o.y2 = mean;
    o.res = b.alloc("", (size_t)BT * ic * meansub_slices());  // row-mean partial sums
    o.N = BT; o.c0 = ic; o.H = H; o.W = W; o.T = Tn;
    p.ops.push_back(o);
  }
  // DVSR_EST_FUSE_PAD=0 keeps every padding a separate launch (A/B aid).  Fused: the conv that produces x stores it
  // straight into the reflect-padded (space-to-depth) tensor this conv reads -- the pad op stays on the tape for the
  // backward (its fold + the producer's activation backward) but launches nothing in the forward.
  static const bool fuse_pad = [] { const char* v = getenv("DVSR_EST_FUSE_PAD"); return !(v && v[0] == '0'); }();
  auto fuse = [&](int mode) {
    // the op before the pad op just pushed must be the conv that produced its input, with this pad as its only reader
    const int pi = (int)p.ops.size() - 1;
    if (!fuse_pad || pi < 1 || (mode != PAD_REFLECT && mode != PAD_REFLECT_S2D)) return;
    Op& pad = p.ops[pi];
    Op& prod = p.ops[pi - 1];
    if (prod.type != OP_CONV || prod.y.space != SP_ARENA || prod.y.off != pad.x0.off || prod.ps || prod.res.valid()) return;
    prod.pad_out = mode == PAD_REFLECT ? PS_PAD_REFLECT : PS_PAD_REFLECT_S2D;
    prod.ypad = pad.y;
    pad.fused_into = pi - 1;
  };
  auto conv3 = [&](const char* pn, const char* cn, T x, int cin, int cout, int mode, bool first) {
    T xp = b.padop(pn, x, mode, BT, cin, H, W, Tn);
    fuse(mode);
    if (first) p.ops.back().no_dgrad = 1;
    const int cin_eff = mode == PAD_REPL_T3 ? 3 * cin : cin;
    T y = b.conv(cn, b.take(), xp, cin_eff, none, 0, BT, H + 2, W + 2, cout, 3, 1, L, none, 0, 1, 0, 0, T(), 0);
    if (first) p.ops.back().no_dgrad = 1;
    return y;
  };
  auto conv4s2 = [&](const char* pn, const char* cn, T x, int cin, int cout) {
    T xp = b.padop(pn, x, PAD_REFLECT_S2D, BT, cin, H, W, Tn);
    fuse(PAD_REFLECT_S2D);
    T y = b.conv(cn, b.take(), xp, 4 * cin, none, 0, BT, (H + 2) / 2, (W + 2) / 2, cout, 2, 1, L, none, 0, 1, 0, 0,
                 T(), 0, 1);
    H /= 2; W /= 2;
    return y;
  };
  T y;
  if (video) {
    y = conv3("pad0", "conv0", xm, ic, nf, PAD_REPL_T3, true);        // Conv3d(in_nc, nf, 3) :76
    y = conv3("pad1", "conv1", y, nf, nf, PAD_REFLECT, false);         // :82
    y = conv4s2("pad2", "conv2", y, nf, 2 * nf);                       // :83
    if (c.scale == 4) y = conv4s2("pad3", "conv3", y, 2 * nf, nf);     // :85
    else y = conv3("pad3", "conv3", y, 2 * nf, nf, PAD_REFLECT, false);  // :84
    y = conv3("pad4", "conv4", y, nf, nf, PAD_REFLECT, false);         // :86
    y = conv3("pad5", "conv5", y, nf, nf, PAD_REPL_T3, false);         // Conv3d(nf, nf, 3) :88
  } else {  // SFDN :44-53
    y = conv3("pad0", "conv0", xm, ic, nf, PAD_REFLECT, true);
    y = conv3("pad1", "conv1", y, nf, nf, PAD_REFLECT, false);
    y = conv3("pad2", "conv2", y, nf, nf, PAD_REFLECT, false);
    y = conv4s2("pad3", "conv3", y, nf, 2 * nf);
    y = conv3("pad4", "conv4", y, 2 * nf, 2 * nf, PAD_REFLECT, false);
    y = conv3("pad5", "conv5", y, 2 * nf, nf, PAD_REFLECT, false);
  }
  y = b.conv("conv6", b.take(), y, nf, none, 0, BT, H, W, ic, 1, 1, NO, none, 0, 1, 0, 0, T(), 0);
  {
    Op o; o.type = OP_ADDMEAN; o.name = "addmean"; o.x0 = y; o.x1 = mean;
    o.y.space = SP_OUTPUT; o.y.off = 0; o.y.numel = (size_t)BT * ic * H * W;
    o.N = BT; o.c0 = ic; o.H = H; o.W = W; o.T = Tn;
    p.ops.push_back(o);
  }
  p.n_params = b.pcur;
  return DVSR_OK;
}

}  // namespace dvsr

extern "C" int dvsr_estimator_plan_create(const dvsr_estimator_config* cfg, int B, int H, int W,
                                          dvsr_estimator_plan** out) {
  return dvsr_estimator_plan_create_grouped(cfg, B, H, W, 1, out);
}

extern "C" int dvsr_estimator_plan_create_grouped(const dvsr_estimator_config* cfg, int B, int H, int W, int grad_groups,
                                                  dvsr_estimator_plan** out) {
  return dvsr_estimator_plan_create_ex(cfg, B, H, W, grad_groups, 1, out);
}

extern "C" int dvsr_estimator_plan_create_ex(const dvsr_estimator_config* cfg, int B, int H, int W, int grad_groups,
                                             int weight_sets, dvsr_estimator_plan** out) {
  DVSR_REQUIRE(cfg && out, DVSR_ERR_INVALID, "estimator_plan_create: null argument");
  DVSR_REQUIRE(weight_sets == 1 || weight_sets == grad_groups, DVSR_ERR_INVALID,
               "estimator_plan_create: weight_sets=%d must be 1 or grad_groups=%d", weight_sets, grad_groups);
  DVSR_REQUIRE(grad_groups >= 1 && B > 0 && B % grad_groups == 0, DVSR_ERR_INVALID,
               "estimator_plan_create: grad_groups=%d must divide the batch B=%d", grad_groups, B);
  DVSR_REQUIRE(cfg->kind == DVSR_ESTIMATOR_MFDN || cfg->kind == DVSR_ESTIMATOR_SFDN, DVSR_ERR_INVALID,
               "estimator_plan_create: kind=%d", cfg->kind);
  DVSR_REQUIRE(cfg->nf > 0 && cfg->in_nc > 0, DVSR_ERR_INVALID, "estimator_plan_create: nf=%d in_nc=%d", cfg->nf,
               cfg->in_nc);
  if (cfg->kind == DVSR_ESTIMATOR_MFDN) {
    DVSR_REQUIRE(cfg->scale == 2 || cfg->scale == 4, DVSR_ERR_UNSUPPORTED,
                 "estimator_plan_create: MFDN scale=%d (2 or 4, LRimg_estimator.py:72)", cfg->scale);
    DVSR_REQUIRE(cfg->nframes > 0, DVSR_ERR_INVALID, "estimator_plan_create: nframes=%d", cfg->nframes);
  } else {
    DVSR_REQUIRE(cfg->scale == 2, DVSR_ERR_UNSUPPORTED, "estimator_plan_create: SFDN is x2 only (got %d)", cfg->scale);
  }
  const int s = cfg->scale;
  DVSR_REQUIRE(B > 0 && H >= 2 * s && W >= 2 * s && H % s == 0 && W % s == 0, DVSR_ERR_INVALID,
               "estimator_plan_create: B=%d H=%d W=%d (H, W must be multiples of the scale %d)", B, H, W, s);
  dvsr_estimator_plan* ep = new dvsr_estimator_plan();
  ep->ecfg = *cfg;
  dvsr_edvr_plan& p = ep->core;
  p.cfg = dvsr_edvr_config{cfg->nf, cfg->nframes, 1, 0, 0, cfg->scale, 0, 0};
  {
    // The estimators' 3x3 stride-1 convolutions run over explicitly (reflection-)padded tensors, whose pitch W + 2 keeps them
    // off the DMA-halo and Winograd kernels; the register-staged kernel with the exact 3-way bf16 operand split
    // (cfg.bf16_mfma = 2: fp32-accurate, six bf16 products per fp32 product) is 1.3x the fp32 MFMA on them: MFDN x4 forward
    // at 5x3x176x320 0.72 -> 0.62 ms, the batched inner step 4.55 -> 4.36 ms per frame.  DVSR_EST_SPLIT=0: fp32 MFMA.
    const char* v = getenv("DVSR_EST_SPLIT");
    if (!v || atoi(v)) { p.cfg.bf16_mfma = 2; p.split_any_pad = true; }
  }
  p.B = B; p.H = H; p.W = W; p.wgroups = grad_groups; p.wsets = weight_sets;
  { const char* v = getenv("DVSR_BWD_STREAMS"); p.side_streams = (v && v[0] == '0') ? 0 : 1; }
  p.fork_every = 1;   // seven layers whose weight gradients outlast the data-gradient chain: every fork at once
  int rc = build_estimator(*ep);
  if (rc != DVSR_OK) { delete ep; return rc; }
  build_backward(p);
  *out = ep;
  return DVSR_OK;
}

extern "C" void dvsr_estimator_plan_destroy(dvsr_estimator_plan* ep) {
  if (!ep) return;
  dvsr_edvr_plan& p = ep->core;
  if (p.side) (void)hipStreamSynchronize(p.side);
  if (p.ev_fork) {
    (void)hipEventDestroy(p.ev_fork);
    (void)hipEventDestroy(p.ev_join);
  }
  delete ep;
}

extern "C" int dvsr_estimator_num_params(const dvsr_estimator_plan* ep) { return ep ? ep->core.n_params : -1; }

extern "C" int dvsr_estimator_num_launches(const dvsr_estimator_plan* ep, int backward) {
  if (!ep) return -1;
  return backward ? (int)ep->core.bops.size() : (int)ep->core.ops.size();
}

extern "C" int dvsr_estimator_plan_work(const dvsr_estimator_plan* ep, double* out9) {
  DVSR_REQUIRE(ep, DVSR_ERR_INVALID, "estimator_plan_work: null plan");
  return dvsr_edvr_plan_work(&ep->core, out9);
}

extern "C" size_t dvsr_estimator_workspace_bytes(const dvsr_estimator_plan* ep, int need_grad) {
  return ep ? dvsr_edvr_workspace_bytes(&ep->core, need_grad) : 0;
}

extern "C" int dvsr_estimator_forward(const dvsr_estimator_plan* ep, const float* const* params, const float* x,
                                      float* out, void* ws, size_t ws_bytes, dvsr_stream_t stream) {
  DVSR_REQUIRE(ep, DVSR_ERR_INVALID, "estimator_forward: null plan");
  return dvsr_edvr_forward(&ep->core, params, x, out, ws, ws_bytes, stream);
}

extern "C" int dvsr_estimator_backward(const dvsr_estimator_plan* ep, const float* const* params, const float* x,
                                       const float* grad_out, float* const* grad_params, void* ws, size_t ws_bytes,
                                       dvsr_stream_t stream) {
  DVSR_REQUIRE(ep, DVSR_ERR_INVALID, "estimator_backward: null plan");
  return dvsr_edvr_backward(&ep->core, params, x, grad_out, grad_params, nullptr, ws, ws_bytes, stream);
}
