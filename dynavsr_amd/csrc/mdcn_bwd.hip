// Backward of the modulated deformable convolution.
//
// Reference: modulated_deform_conv_cuda_backward (deform_conv_cuda.cpp:566-679):
//   dcol = W^T . gout (addmm, :617-620) -> col2im_coord kernel (kernel.cu:694-766: grad offset/mask)
//   -> col2im kernel (kernel.cu:634-692: grad input, atomicAdd) -> im2col again (:569-632)
//   -> grad_weight += gout . col^T, grad_bias += gout . 1 (:653-665).
// Two paths:
//  * EDVR's configuration (3x3, stride/pad/dilation 1, C/dg a multiple of 8): ONE fused kernel,
//    mdcn_bwd_fused_kernel -- dcol = W^T . gout on the MFMA stays in registers (never in HBM), the per-(pixel,
//    tap) sampler produces the offset / mask gradients in registers, the input gradient is accumulated in an
//    LDS window with 64-bit fixed-point ds_add_u64 and leaves as one fp32 atomic per touched element; the weight
//    gradient is contracted IN the kernel as well (r03): the modulated samples the sampler has in registers go through
//    LDS into a second MFMA phase, dW_tile[o][tap, c] = sum over the tile's pixels of gout * sample, and leave as one
//    [Cout][72] partial per workgroup (plain stores) that a small kernel sums -- NO column buffer anywhere (the
//    reference, and this file until r02, wrote [N, C*9, H*W] floats and ran a 1x1 weight-gradient GEMM over them:
//    663 MB written + read per L1 call at 5x64x180x320).  Groups of 16 channels are walked as two 8-channel chunks.
//    Round 6: where Cout = 64, W % 4 == 0 and the tensors are 16-byte aligned both contractions run on the bf16 pipe under
//    the exact 3-way operand split (template parameter SPLIT; DVSR_DCN_BWD=fp32 keeps the fp32 MFMAs); the A operands of
//    the first phase are laid out once per call (mdcn_bwd_wt_kernel / mdcn_bwd_wt3_kernel), the window is staged with
//    16-byte loads, the lane's offsets / masks are streamed through the sampling phase one tap ahead, and the C / 8 chunk
//    workgroups of a tile share one XCD's L2.  1.58 -> 1.02-1.06 ms at 5x64x180x320.
//  * every other configuration (C/dg = 4, other strides / dilations; DVSR_DCN_BWD=unfused forces it): the
//    three-kernel form of the reference -- the two contractions on the MFMA conv kernels (dcol = 1x1 "dgrad"
//    over the flattened [Cout][C*9] weight, dW/db = 1x1 wgrad over the column buffer) around the fused
//    col2im / col2im_coord sampler kernels below; the [C*9, P] buffers do round-trip HBM there.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dvsr {

struct DcnB {
  const float* x; const float* off; const float* msk;
  long long off_bs, msk_bs;
  int mask_logit;
  int N, C, H, W, Ho, Wo, stride, pad, dil, dg, cpg;
};

// col[n][(g*cpg + c)*9 + tap][p] = mask * bilinear(x[n, g*cpg + c], p + tap + offset)
__global__ void mdcn_im2col_kernel(DcnB a, float* __restrict__ col) {
  const size_t P = (size_t)a.Ho * a.Wo, HW = (size_t)a.H * a.W;
  const size_t total = (size_t)a.N * a.dg * 9 * P;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % P;
    size_t t = i / P;
    const int tap = (int)(t % 9); t /= 9;
    const int g = (int)(t % a.dg);
    const int n = (int)(t / a.dg);
    const int py = (int)(p / a.Wo), px = (int)(p % a.Wo);
    const float* offn = a.off + (size_t)n * a.off_bs;
    const float oh = offn[(size_t)(g * 18 + 2 * tap) * P + p];
    const float ow = offn[(size_t)(g * 18 + 2 * tap + 1) * P + p];
    float m = a.msk[(size_t)n * a.msk_bs + (size_t)(g * 9 + tap) * P + p];
    if (a.mask_logit) m = sigmoidf_(m);
    const float h_im = (float)(py * a.stride - a.pad + (tap / 3) * a.dil) + oh;
    const float w_im = (float)(px * a.stride - a.pad + (tap % 3) * a.dil) + ow;
    DcnTap tp;
    const bool in = make_tap(h_im, w_im, a.H, a.W, tp);
    const float* xg = a.x + ((size_t)n * a.C + g * a.cpg) * HW;
    float* dst = col + ((size_t)n * a.C * 9 + (size_t)(g * a.cpg) * 9 + tap) * P + p;
    for (int c = 0; c < a.cpg; ++c) {
      float v = 0.f;
      if (in) {
        const float* pl = xg + (size_t)c * HW;
        const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
        const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
        v = (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * m;
      }
      dst[(size_t)c * 9 * P] = v;
    }
  }
}

// From dcol[n][(g*cpg + c)*9 + tap][p]: grad offset (dy, dx), grad mask (or mask logit) and the
// input-gradient scatter.  goff/gmsk are written (=), gx is accumulated with hardware fp32 atomics
// (order differs run to run exactly like the reference's atomicAdd, kernel.cu:687).
__global__ void mdcn_col2im_coord_kernel(DcnB a, const float* __restrict__ dcol, float* __restrict__ goff,
                                         long long goff_bs, float* __restrict__ gmsk, long long gmsk_bs,
                                         float* __restrict__ gx) {
  const size_t P = (size_t)a.Ho * a.Wo, HW = (size_t)a.H * a.W;
  const size_t total = (size_t)a.N * a.dg * 9 * P;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % P;
    size_t t = i / P;
    const int tap = (int)(t % 9); t /= 9;
    const int g = (int)(t % a.dg);
    const int n = (int)(t / a.dg);
    const int py = (int)(p / a.Wo), px = (int)(p % a.Wo);
    const float* offn = a.off + (size_t)n * a.off_bs;
    const float oh = offn[(size_t)(g * 18 + 2 * tap) * P + p];
    const float ow = offn[(size_t)(g * 18 + 2 * tap + 1) * P + p];
    const float mraw = a.msk[(size_t)n * a.msk_bs + (size_t)(g * 9 + tap) * P + p];
    const float m = a.mask_logit ? sigmoidf_(mraw) : mraw;
    const float h_im = (float)(py * a.stride - a.pad + (tap / 3) * a.dil) + oh;
    const float w_im = (float)(px * a.stride - a.pad + (tap % 3) * a.dil) + ow;
    DcnTap tp;
    const bool in = make_tap(h_im, w_im, a.H, a.W, tp);
    float gh = 0.f, gw = 0.f, gm = 0.f;
    if (in) {
      const float* xg = a.x + ((size_t)n * a.C + g * a.cpg) * HW;
      float* gxg = gx ? gx + ((size_t)n * a.C + g * a.cpg) * HW : nullptr;
      const float* dc_ = dcol + ((size_t)n * a.C * 9 + (size_t)(g * a.cpg) * 9 + tap) * P + p;
      const float hh = 1.f - tp.lh, hw = 1.f - tp.lw;
      for (int c = 0; c < a.cpg; ++c) {
        const float dc = dc_[(size_t)c * 9 * P];
        const float* pl = xg + (size_t)c * HW;
        const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
        const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
        gm += dc * (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4);           // kernel.cu:752
        gh += (-hw * v1 - tp.lw * v2 + hw * v3 + tp.lw * v4) * dc * m;              // :541-550
        gw += (-hh * v1 + hh * v2 - tp.lh * v3 + tp.lh * v4) * dc * m;              // :552-561
        if (gxg) {
          const float top = dc * m;                                                // :672
          float* gp = gxg + (size_t)c * HW;
          if (tp.v1) unsafeAtomicAdd(gp + tp.o1, tp.w1 * top);
          if (tp.v2) unsafeAtomicAdd(gp + tp.o2, tp.w2 * top);
          if (tp.v3) unsafeAtomicAdd(gp + tp.o3, tp.w3 * top);
          if (tp.v4) unsafeAtomicAdd(gp + tp.o4, tp.w4 * top);
        }
      }
    }
    goff[(size_t)n * goff_bs + (size_t)(g * 18 + 2 * tap) * P + p] = gh;
    goff[(size_t)n * goff_bs + (size_t)(g * 18 + 2 * tap + 1) * P + p] = gw;
    if (a.mask_logit) gm *= m * (1.f - m);
    gmsk[(size_t)n * gmsk_bs + (size_t)(g * 9 + tap) * P + p] = gm;
  }
}

// -------------------------------------------------------------------------------------------------
// Fused backward for the EDVR configuration (8 channels per deformable group, 3x3, stride 1, pad 1,
// dilation 1).  One workgroup = one 8x32-pixel tile of one (frame, group):
//   1. dcol tile on MFMA, kept in registers: D[m = tap*8 + c][pixel] = sum_o W[o][g*8+c][tap] * gout[o][pixel]
//      (3 M-tiles of 32 rows, 72 used; 2 pixel rows per wave; K = Cout).  With this row order a lane
//      holds, for each of its pixels, 4 channels (c = 4*hi + 0..3) of 4 taps per M-tile -- the lane pair
//      (l, l+32) holds the 8 channels of a (pixel, tap).  dcol never touches HBM.
//   2. per (pixel, tap): bilinear geometry once, 4 corner x 4 channel reads from the LDS-staged input
//      window, mask / offset gradient = in-register sums + one exchange with the partner lane, the
//      modulated sample is written to the column buffer for the weight gradient, and the input gradient
//      is accumulated with LDS atomics into a window of the same shape, laid out [channel][y][x] so that
//      the lanes of a wave (adjacent pixels) hit adjacent banks.  The window is 64-bit fixed point and
//      the atomics are ds_add_u64: ds_add_f32 was measured ~10x slower than the integer LDS atomics
//      (357 vs 103 us for the whole kernel at 5x44x80);
//   3. the window is flushed with ONE global atomic per touched element: ~6 k per workgroup instead of
//      the 74 k (256 px x 9 taps x 4 corners x 8 channels) of the unfused kernel.
// Samples whose 2x2 footprint leaves the window (|offset| > HALO) take exact global gathers / atomics.
// -------------------------------------------------------------------------------------------------
struct DcnF {
  const float* x; const float* off; const float* msk; const float* w; const float* gout;
  float* gx; float* goff; float* gmsk;
  float* dwp;  // [N][C/8][tiles][Cout][72] weight-gradient partials (m = tap * 8 + c), or null: no weight gradient
  float* dbp;  // [N][tiles][Cout] bias-gradient partials
  long long off_bs, msk_bs, goff_bs, gmsk_bs;
  int mask_logit, N, C, H, W, Cout, dg, tiles_x, tiles_y;
  int sub;  // 8-channel chunks per deformable group (1: EDVR-M, 2: EDVR-L); a workgroup takes one of the C/8 chunks
  int wdiv = 1; long long w_gs = 0;  // per-sample weight sets: frame n convolves with w + (n / wdiv) * w_gs
  const float* wtp = nullptr;        // [weight set][C / 8][3][Cout / 2][64]: the A operands of phase 1 in LDS order (mdcn_bwd_wt_kernel)
  int vec = 0;                       // W % 4 == 0 and x / gout 16-byte aligned: the window and the gout tiles are staged with 16-byte loads
#ifdef DVSR_CONV_TRACE
  long long* trace;  // debug build only (tools/dcn_bwd_trace.py): 16 cycle stamps per workgroup
#endif
};
#ifdef DVSR_CONV_TRACE
#define DCNB_STAMP(i)                                                                                              \
  do {                                                                                                             \
    if (a.trace && threadIdx.x == 0)                                                                               \
      a.trace[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
static long long* g_dcnb_trace = nullptr;
static int g_dcnb_countdown = -1;
extern "C" int dvsr_debug_dcn_bwd_trace(void* buf, int launch_index) {
  g_dcnb_trace = (long long*)buf;
  g_dcnb_countdown = launch_index;
  return 0;
}
#else
#define DCNB_STAMP(i) \
  do {                \
  } while (0)
#endif

typedef __bf16 dbbf8 __attribute__((ext_vector_type(8)));
typedef __bf16 dbbf2 __attribute__((ext_vector_type(2)));
typedef float dbf2 __attribute__((ext_vector_type(2)));
typedef unsigned dbu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned dcnb_cvt_pk(float x, float y) {   // {bf16(x) in bits 15:0, bf16(y) in bits 31:16}, RNE
  return __builtin_bit_cast(unsigned, __builtin_convertvector(dbf2{x, y}, dbbf2));
}
// two fp32 values -> their exact three-way bf16 split, packed pairwise: v = hi + mid + lo to 2^-24 relative (common.h)
__device__ __forceinline__ void dcnb_split_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
  h = dcnb_cvt_pk(v0, v1);
  const float r0 = v0 - __builtin_bit_cast(float, h << 16), r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = dcnb_cvt_pk(r0, r1);
  const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = dcnb_cvt_pk(q0, q1);
}
__device__ __forceinline__ f32x16 dcnb_mma(dbu4 a, dbu4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dbbf8, a), __builtin_bit_cast(dbbf8, b), c, 0, 0, 0);
}

// SPLIT (round 6; Cout = 64, 16-byte aligned tensors with W % 4 == 0): the two contractions -- dcol = W^T gout and the tile's
// weight gradient -- run on v_mfma_f32_32x32x16_bf16 under the exact 3-way bf16 split of both operands (six bf16 products per
// fp32 product, fp32 accumulate: fp32 results), as the forward's contraction does since round 5.  The fp32 MFMAs they replace
// occupied the vector datapath for 24.6 k cycles per wave with nothing running beside them; the bf16 MFMAs take 9.2 k on a
// pipe the sampler of the co-resident workgroup issues beside.
template <int HALO, bool SPLIT>
__global__ __launch_bounds__(256, 2) void mdcn_bwd_fused_kernel(DcnF a) {
  constexpr int TH = 8, TW = 32, XH = TH + 2 + 2 * HALO, XW = TW + 2 + 2 * HALO, XPX = XH * XW;
  // The INPUT window's rows start at the 16-byte boundary left of the window (column ox0 - 8: XA0 columns before its first
  // one) and are XWA floats long, so that a row is twelve aligned 16-byte groups, each wholly inside or outside the image
  // when W % 4 == 0; the GRADIENT window keeps the tight pitch XW (two workgroups per CU: 27.6 + 48.4 KB each).
  constexpr int XWA = 48, XA0 = 8 - 1 - HALO, XPXA = XH * XWA + 8;   // (+ 8: the lane halves' channels c, c + 4 sit 32 banks apart)
  static_assert(XA0 >= 0 && XA0 + XW <= XWA, "the aligned window rows must cover the sampling window");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_x = smem;              // [8][XH][XWA] input window
  float* const s_wt = smem + 8 * XPXA;  // [3][Cout/2][64]: A operands, lane-major (MFMA phase only) ...
  unsigned long long* const s_gq = reinterpret_cast<unsigned long long*>(smem + 8 * XPXA);  // ... then [8][XPX] gradient window
  __shared__ float s_max[8];
  const int KST = a.Cout >> 1;

  // kc = 8-channel chunk of the input; its deformable group g supplies offsets / masks.  With more
  // than one chunk per group the offset / mask gradients of the chunks are summed with atomics into buffers the
  // host zeroed (two commutative adds per element: still deterministic).
  // Workgroup -> (frame, tile, chunk): the C / 8 chunk workgroups of one (frame, tile) all read the same gout tile (twice each:
  // phase 1 and the weight gradient), so they sit on ONE XCD in consecutive dispatch slots -- workgroups are dealt to the eight
  // XCDs round-robin (blockIdx.x & 7) and an XCD has its own L2 (round 6; as a (tile, chunk, frame) grid the eight of a tile
  // were 230 dispatches apart and on eight different XCDs).
  const int nkc = a.C >> 3, ntile = a.tiles_x * a.tiles_y;
  const int slot = blockIdx.x >> 3;
  const int pair = (int)(blockIdx.x & 7) + 8 * (slot / nkc);   // (frame, tile) index
  if (pair >= a.N * ntile) return;
  const int kc = slot - (slot / nkc) * nkc, n = pair / ntile, tile = pair - n * ntile;
  const int g = kc / a.sub;
  const int tx_ = tile % a.tiles_x, ty_ = tile / a.tiles_x;
  const int oy0 = ty_ * TH, ox0 = tx_ * TW;
  const int wy0 = oy0 - 1 - HALO, wx0 = ox0 - 1 - HALO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const size_t HW = (size_t)a.H * a.W;
  const float* xg = a.x + ((size_t)n * a.C + kc * 8) * HW;
  DCNB_STAMP(0);

  // ---- stage the input window (zero outside the image) and W^T.
  // Round 6: both sets of loads are issued before the first LDS write (they used to be dependent round trips, and the W^T
  // image 24 stride-9 gathers per lane: the stamps had 44 k cycles here); the lane's 54 offset / mask values are fetched
  // after phase 1.
  const int px = ox0 + lo;
  int py[2];
  bool pv[2];
  size_t pofs[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    py[nt] = oy0 + 2 * wave + nt;
    pv[nt] = py[nt] < a.H && px < a.W;
    pofs[nt] = pv[nt] ? (size_t)py[nt] * a.W + px : 0;
  }
  const float* offn = a.off + (size_t)n * a.off_bs;
  const float* mskn = a.msk + (size_t)n * a.msk_bs;
  // W^T: the LDS image was laid out once per call by mdcn_bwd_wt_kernel (the 24 stride-9 gathers per lane this staging used to
  // do were the longest phase of the workgroup) -- 16-byte loads, six per lane and batch (one batch for Cout = 64)
  // (SPLIT: the image holds the three bf16 pieces of the A fragments, [piece][M-tile][16-cout chunk][lane][8] -- 36,864 bytes per
  // (weight set, chunk kc), mdcn_bwd_wt3_kernel; the per-(set, kc) stride in the workspace is that of the larger image)
  constexpr int WIMG = SPLIT ? 9216 : 0;   // floats per image (SPLIT); fp32 form: 3 * KST * 64
  const int wimg = SPLIT ? WIMG : 3 * KST * 64;
  const f32x4* wtp = reinterpret_cast<const f32x4*>(a.wtp + ((size_t)(a.w_gs ? n / a.wdiv : 0) * (a.C >> 3) + kc) * (size_t)wimg);
  const int nv = wimg / 4;       // 16-byte groups: 1536 for Cout = 64 (2304 split), 3072 for 128
  constexpr int WE = SPLIT ? 9 : 6;
  f32x4 rw[WE];
  auto load_w = [&](int base) {
#pragma unroll
    for (int e = 0; e < WE; ++e) {
      const int idx = base + tid + 256 * e;
      rw[e] = idx < nv ? wtp[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_w = [&](int base) {
#pragma unroll
    for (int e = 0; e < WE; ++e) {
      const int idx = base + tid + 256 * e;
      if (idx < nv) *reinterpret_cast<f32x4*>(s_wt + 4 * idx) = rw[e];
    }
  };
  // The lane's offsets / masks are STREAMED through the sampling phase, one tap ahead (round 6): fetched in the prologue and held
  // across the MFMA phase -- 54 registers -- a third of them was spilled as it landed, one full vmcnt(0) wait each (the stamps
  // had 28 k cycles before the first barrier).  ovals[nt] = {offset y, offset x, mask (logit)} of a tap.
  const unsigned po32[2] = {(unsigned)pofs[0], (unsigned)pofs[1]};
  auto load_tap = [&](float (&v)[2][3], int tap) {
    const float* ph = offn + (size_t)(g * 18 + 2 * tap) * HW;
    const float* pm = mskn + (size_t)(g * 9 + tap) * HW;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      v[nt][0] = ph[po32[nt]];
      v[nt][1] = (ph + HW)[po32[nt]];
      v[nt][2] = pm[po32[nt]];
    }
  };
  if (a.vec) {
    // 8 channels x XH rows x 12 groups of 16 bytes: 7 loads per lane
    constexpr int NV = 8 * XH * (XWA / 4), VE = (NV + 255) / 256;
    f32x4 rv[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int idx = tid + 256 * e;
      const int c = idx / (XH * (XWA / 4)), r = idx - c * (XH * (XWA / 4));
      const int ry = r / (XWA / 4), v = r - ry * (XWA / 4);
      const int gy_ = wy0 + ry, gx_ = ox0 - 8 + 4 * v;
      const bool ok = idx < NV && (unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W;
      rv[e] = ok ? *reinterpret_cast<const f32x4*>(xg + (size_t)c * HW + (size_t)gy_ * a.W + gx_) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    load_w(0);
    DCNB_STAMP(10);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int idx = tid + 256 * e;
      const int c = idx / (XH * (XWA / 4)), r = idx - c * (XH * (XWA / 4));
      if (idx < NV) *reinterpret_cast<f32x4*>(s_x + c * XPXA + 4 * r) = rv[e];
    }
    DCNB_STAMP(11);
  } else {
    constexpr int XE = (8 * XPX + 255) / 256;
    float rx_[XE];
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int idx = tid + 256 * e;
      const int c = idx / XPX, r = idx - c * XPX;
      const int ry = r / XW, rx = r - ry * XW;
      const int gy_ = wy0 + ry, gx_ = wx0 + rx;
      const bool ok = idx < 8 * XPX && (unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W;
      rx_[e] = ok ? xg[(size_t)c * HW + (size_t)gy_ * a.W + gx_] : 0.f;
    }
    load_w(0);
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int idx = tid + 256 * e;
      const int c = idx / XPX, r = idx - c * XPX;
      const int ry = r / XW, rx = r - ry * XW;
      if (idx < 8 * XPX) s_x[c * XPXA + ry * XWA + XA0 + rx] = rx_[e];
    }
  }
  store_w(0);
  for (int base = 256 * WE; base < nv; base += 256 * WE) {   // (Cout = 128)
    load_w(base);
    store_w(base);
  }
  DCNB_STAMP(12);
  DCNB_STAMP(1);
  __syncthreads();
  DCNB_STAMP(2);

  // ---- 1. dcol tile: D[mt][nt], pixel row 2*wave + nt, column lo.  gout operands are fetched 8 k-steps
  // (48 MFMAs) ahead.
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j2][r] = 0.f;
  const float* gon = a.gout + (size_t)n * a.Cout * HW;
  float b0[8][2], b1[8][2];
  const unsigned pb32[2] = {(unsigned)(hi * HW + pofs[0]), (unsigned)(hi * HW + pofs[1])};   // (an image's gout < 2^32 bytes)
  auto load_b = [&](float (&b)[8][2], int kbase) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float* pl = gon + (size_t)(2 * (kbase + q)) * HW;   // wave-uniform
      b[q][0] = pl[pb32[0]];
      b[q][1] = pl[pb32[1]];
    }
  };
  if constexpr (SPLIT) {
    // K = 64 couts in four chunks of 16: lane (lo, hi) supplies couts 16 j + 8 hi + 0..7 of its two pixels -- eight loads per
    // pixel row and chunk, split into the three pieces pairwise --, the A fragments of a chunk are nine ds_read_b128.
    // Products per (M-tile, pixel row, chunk): Ah Bh, Ah Bm, Am Bh, Am Bm, Ah Bl, Al Bh.
    const dbu4* const s_w16 = reinterpret_cast<const dbu4*>(s_wt);   // [piece 3][mt 3][chunk 4][lane 64]
    float braw[2][2][8];
    auto load_c = [&](float (&b)[2][8], int j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const size_t ob = (size_t)(16 * j + 8 * hi + i) * HW;
        b[0][i] = gon[ob + pofs[0]];
        b[1][i] = gon[ob + pofs[1]];
      }
    };
    load_c(braw[0], 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j + 1 < 4) load_c(braw[(j + 1) & 1], j + 1);
      dbu4 Bh[2], Bm[2], Bl[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v0 = pv[nt] ? braw[j & 1][nt][2 * q] : 0.f, v1 = pv[nt] ? braw[j & 1][nt][2 * q + 1] : 0.f;
          unsigned h, m, l;
          dcnb_split_pair(v0, v1, h, m, l);
          Bh[nt][q] = h; Bm[nt][q] = m; Bl[nt][q] = l;
        }
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const dbu4 Ah = s_w16[((0 * 3 + mt) * 4 + j) * 64 + lane];
        const dbu4 Am = s_w16[((1 * 3 + mt) * 4 + j) * 64 + lane];
        const dbu4 Al = s_w16[((2 * 3 + mt) * 4 + j) * 64 + lane];
        // (the two pixel rows alternate: consecutive MFMAs never wait on each other's accumulator)
        acc[mt][0] = dcnb_mma(Ah, Bh[0], acc[mt][0]); acc[mt][1] = dcnb_mma(Ah, Bh[1], acc[mt][1]);
        acc[mt][0] = dcnb_mma(Ah, Bm[0], acc[mt][0]); acc[mt][1] = dcnb_mma(Ah, Bm[1], acc[mt][1]);
        acc[mt][0] = dcnb_mma(Am, Bh[0], acc[mt][0]); acc[mt][1] = dcnb_mma(Am, Bh[1], acc[mt][1]);
        acc[mt][0] = dcnb_mma(Am, Bm[0], acc[mt][0]); acc[mt][1] = dcnb_mma(Am, Bm[1], acc[mt][1]);
        acc[mt][0] = dcnb_mma(Ah, Bl[0], acc[mt][0]); acc[mt][1] = dcnb_mma(Ah, Bl[1], acc[mt][1]);
        acc[mt][0] = dcnb_mma(Al, Bh[0], acc[mt][0]); acc[mt][1] = dcnb_mma(Al, Bh[1], acc[mt][1]);
      }
    }
  } else {
  auto mfma8 = [&](const float (&b)[8][2], int kbase) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float v0 = pv[0] ? b[q][0] : 0.f, v1 = pv[1] ? b[q][1] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const float av = s_wt[(mt * KST + kbase + q) * 64 + lane];
        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, v0, acc[mt][0], 0, 0, 0);
        acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, v1, acc[mt][1], 0, 0, 0);
      }
    }
  };
  load_b(b0, 0);
  for (int kb = 0; kb < KST; kb += 16) {  // KST is a multiple of 16 (Cout % 32 == 0)
    load_b(b1, kb + 8);
    mfma8(b0, kb);
    if (kb + 16 < KST) load_b(b0, kb + 16);
    mfma8(b1, kb + 8);
  }
  }

  DCNB_STAMP(3);
  // ---- gradient-window scale: the input gradient is accumulated in 64-bit fixed point with the
  // workgroup's largest |dcol * mask| mapped to [2^39, 2^40): at most 9216 contributions meet in a cell, so
  // the sums stay below 2^54, and every contribution keeps >= 24 significant bits down to 2^-16 of the
  // maximum (better than an fp32 running sum).  Integer adds are associative: the window is deterministic.
  // (round 6: the scale is set from max |dcol| x max |mask| -- an upper bound of the largest product, so the sums still cannot
  // overflow; a sigmoid mask is at most 1 and costs nothing here, plain masks are read once for their maximum.  Against the
  // exact maximum this gives away log2(max |mask| / the mask at the largest |dcol|) of the 16 spare bits.)
  float amax = 0.f, mmax = 1.f;
  if (!a.mask_logit) {
    mmax = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) mmax = fmaxf(mmax, fabsf((mskn + (size_t)(g * 9 + tap) * HW)[po32[nt]]));
  }
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) amax = fmaxf(amax, fabsf(acc[tap >> 2][nt][(tap & 3) * 4 + cq]));
  for (int o = 32; o > 0; o >>= 1) {
    amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    mmax = fmaxf(mmax, __shfl_xor(mmax, o, 64));
  }
  if (lane == 0) { s_max[wave] = amax; s_max[4 + wave] = mmax; }
  __syncthreads();  // all waves are done with s_wt
  amax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])) * fmaxf(fmaxf(s_max[4], s_max[5]), fmaxf(s_max[6], s_max[7]));
  int aexp = 0;
  (void)frexpf(amax, &aexp);  // amax = f * 2^aexp, f in [0.5, 1)
  const float qscale = amax > 0.f ? ldexpf(1.f, 40 - aexp) : 1.f;
  const float qinv = amax > 0.f ? ldexpf(1.f, aexp - 40) : 1.f;
  for (int idx = tid; idx < 8 * XPX; idx += 256) s_gq[idx] = 0ull;
  __syncthreads();
  DCNB_STAMP(4);

  // ---- 2. sampling: this lane's 4 channels (4*hi .. 4*hi+3) of every (pixel, tap).
  // Straight-line fast path (round 2; the cycle-stamp trace, tools/dcn_bwd_trace.py, had this phase at half of the
  // kernel: ~850 instructions per (pixel row, tap), most of them exec-mask bookkeeping of nested divergent branches,
  // float -> int64 conversions and two divisions per sigmoid).  Window coordinates are clamped (always a legal LDS
  // address) and a sample that is off the tile or outside the window contributes exact zeros; the window is zero outside
  // the image, which reproduces the per-corner validity and -- but for the lower bound, tested -- the (-1,H)x(-1,W) gate
  // for every in-window sample, as in the forward kernel.  Samples outside the window are flagged and redone exactly in the rare loop below.
  float* goffn = a.goff + (size_t)n * a.goff_bs;
  float* gmskn = a.gmsk + (size_t)n * a.gmsk_bs;
  float* gxg = a.gx ? a.gx + ((size_t)n * a.C + kc * 8) * HW : nullptr;
  // float -> int64 (round to nearest) through the double mantissa: bits(double(v) + 1.5 * 2^52) - bits(1.5 * 2^52) is the
  // integer for |v| < 2^51: 4 VALU instructions (__float2ll_rn is a ~14-instruction sequence)
  auto q64 = [](float v) {
    return (unsigned long long)(__double_as_longlong((double)v + 6755399441055744.0) - 0x4338000000000000LL);
  };
  // The modulated samples of this lane's (pixel, tap, 4 channels) -- the B operands of phase 4 -- REPLACE the dcol values
  // in the accumulator registers as those are consumed (72 more live registers would spill: the kernel sits at 256).
#define COLR(tap_, nt_, cq_) acc[(tap_) >> 2][nt_][((tap_) & 3) * 4 + (cq_)]
  auto store_grads = [&](int tap, int nt, float gm, float gh, float gw, float m) {
    if (hi == 0) {
      float* ph = goffn + (size_t)(g * 18 + 2 * tap) * HW + pofs[nt];
      float* pm = gmskn + (size_t)(g * 9 + tap) * HW + pofs[nt];
      const float gmv = a.mask_logit ? gm * m * (1.f - m) : gm;
      if (a.sub == 1) {
        ph[0] = gh; ph[HW] = gw; pm[0] = gmv;
      } else {
        unsafeAtomicAdd(ph, gh); unsafeAtomicAdd(ph + HW, gw); unsafeAtomicAdd(pm, gmv);
      }
    }
  };
  unsigned fixbits = 0;  // bit 2 tap + nt: this lane's sample left the staged window
  float ov[2][2][3];     // [tap & 1]: the tap being sampled and the one in flight
  load_tap(ov[0], 0);
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int tap = mt * 4 + t4;
      if (tap >= 9) continue;
      const int ki = tap / 3, kj = tap - 3 * ki;
      if (tap + 1 < 9) load_tap(ov[(tap + 1) & 1], tap + 1);
      __builtin_amdgcn_sched_barrier(0);   // (the next tap's six loads go out here, not eight taps early)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float mraw = ov[tap & 1][nt][2];
        const float m = a.mask_logit ? __builtin_amdgcn_rcpf(1.f + __expf(-mraw)) : mraw;
        const float h_im = (float)(py[nt] - 1 + ki) + ov[tap & 1][nt][0];
        const float w_im = (float)(px - 1 + kj) + ov[tap & 1][nt][1];
        const float hf = floorf(h_im), wf = floorf(w_im);
        const float lh = h_im - hf, lw = w_im - wf;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const int ry = (int)hf - wy0, rx = (int)wf - wx0;
        const int inwin = ((unsigned)ry <= (unsigned)(XH - 2)) & ((unsigned)rx <= (unsigned)(XW - 2));
        // (h_im, w_im > -1: the zero padding reproduces the gate for the sample VALUE, but at exactly -1 the coordinate
        // gradient of the padded image is not zero while the reference's gate returns 0; the upper bounds need no test)
        const int ok = (int)pv[nt] & inwin & (h_im > -1.f) & (w_im > -1.f);
        fixbits |= (unsigned)((int)pv[nt] & (inwin ^ 1)) << (2 * tap + nt);
        const int cry = min(max(ry, 0), XH - 2), crx = min(max(rx, 0), XW - 2);
        const int cell = cry * XW + crx, cellx = cry * XWA + XA0 + crx;
        const float mk = ok ? m : 0.f;               // zero for samples this path does not own
        const float tsc = mk * qscale;
        float gm = 0.f, gh = 0.f, gw = 0.f;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
          const int c = 4 * hi + cq;
          const float d = acc[mt][nt][t4 * 4 + cq];
          const float* p1 = s_x + c * XPXA + cellx;
          const float v1 = p1[0], v2 = p1[1], v3 = p1[XWA], v4 = p1[XWA + 1];
          const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
          gm += d * val;                                              // kernel.cu:752
          gh += (-hw * v1 - lw * v2 + hw * v3 + lw * v4) * d;         // :541-550 (x mask below)
          gw += (-hh * v1 + hh * v2 - lh * v3 + lh * v4) * d;         // :552-561
          if (gxg) {
            // 64-bit fixed point: ds_add_u64 runs at full LDS rate; native ds_add_f32 (inline asm, no return) was
            // measured again in round 2: this phase 47 k -> 263 k cycles, the whole call 196 -> 425 us
            unsigned long long* q1 = s_gq + c * XPX + cell;
            const float ts = d * tsc;                                 // :672, scaled
#if defined(DCNB_NOATOM)   // (probe builds, results wrong: what the conversions + atomics / the atomics alone cost)
            asm volatile("" :: "v"(w1 * ts), "v"(w2 * ts), "v"(w3 * ts), "v"(w4 * ts), "v"(q1));
#elif defined(DCNB_NOCVT)
            atomicAdd(q1, (unsigned long long)__float_as_uint(w1 * ts));
            atomicAdd(q1 + 1, (unsigned long long)__float_as_uint(w2 * ts));
            atomicAdd(q1 + XW, (unsigned long long)__float_as_uint(w3 * ts));
            atomicAdd(q1 + XW + 1, (unsigned long long)__float_as_uint(w4 * ts));
#else
            atomicAdd(q1, q64(w1 * ts));
            atomicAdd(q1 + 1, q64(w2 * ts));
            atomicAdd(q1 + XW, q64(w3 * ts));
            atomicAdd(q1 + XW + 1, q64(w4 * ts));
#endif
          }
          // (a sample that left the window keeps d: loop 2b needs it, and stores the exact sample instead)
          COLR(tap, nt, cq) = ((int)pv[nt] & (inwin ^ 1)) ? d : val * mk;
        }
        gm = ok ? gm : 0.f; gh *= mk; gw *= mk;
        // the partner lane (other 4 channels of the same pixel) completes the sums
        gm += __shfl_xor(gm, 32, 64);
        gh += __shfl_xor(gh, 32, 64);
        gw += __shfl_xor(gw, 32, 64);
        if (pv[nt]) store_grads(tap, nt, gm, gh, gw, m);
      }
    }
  }
  // ---- 2b. samples that left the window (|offset| > HALO pixels): exact clamped global gathers / atomics; their
  // gradients overwrite the zeros the fast path stored.  Both lanes of a pixel pair take the same branch.
  if (__builtin_amdgcn_ballot_w64(fixbits != 0) != 0) {
#pragma unroll 1
    for (int b = 0; b < 18; ++b) {
      if (__builtin_amdgcn_ballot_w64((fixbits >> b) & 1) == 0) continue;
      const int tap = b >> 1, nt = b & 1;
      const int mt = tap >> 2, t4 = tap & 3;
      const int ki = tap / 3, kj = tap - 3 * ki;
      if ((fixbits >> b) & 1) {
        // runtime-indexed register arrays would go to scratch: select the operands with compile-time indices
        float dsel[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 2; ++r)
            if (t == tap && r == nt) {
#pragma unroll
              for (int cq = 0; cq < 4; ++cq) dsel[cq] = acc[t >> 2][r][(t & 3) * 4 + cq];
            }
        (void)mt; (void)t4;
        const int pyv = nt ? py[1] : py[0];
        const size_t pof = nt ? pofs[1] : pofs[0];
        // (the streamed offsets of this tap are gone: read again, this path is rare)
        const float oh = offn[(size_t)(g * 18 + 2 * tap) * HW + pof], ow = offn[(size_t)(g * 18 + 2 * tap + 1) * HW + pof];
        const float mraw2 = mskn[(size_t)(g * 9 + tap) * HW + pof];
        const float m = a.mask_logit ? __builtin_amdgcn_rcpf(1.f + __expf(-mraw2)) : mraw2;
        const float h_im = (float)(pyv - 1 + ki) + oh;
        const float w_im = (float)(px - 1 + kj) + ow;
        float gm = 0.f, gh = 0.f, gw = 0.f;
        float colv[4] = {0.f, 0.f, 0.f, 0.f};
        DcnTap tp;
        if (make_tap(h_im, w_im, a.H, a.W, tp)) {
          const float hf = floorf(h_im), wf = floorf(w_im);
          const float lh = h_im - hf, lw = w_im - wf;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
          for (int cq = 0; cq < 4; ++cq) {
            const int c = 4 * hi + cq;
            const float d = dsel[cq];
            const float* pl = xg + (size_t)c * HW;
            const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
            const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
            const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
            colv[cq] = val * m;
            gm += d * val;
            gh += (-hw * v1 - lw * v2 + hw * v3 + lw * v4) * d * m;
            gw += (-hh * v1 + hh * v2 - lh * v3 + lh * v4) * d * m;
            const float top = d * m;
            if (gxg) {
              float* gp = gxg + (size_t)c * HW;
              if (tp.v1) unsafeAtomicAdd(gp + tp.o1, w1 * top);
              if (tp.v2) unsafeAtomicAdd(gp + tp.o2, w2 * top);
              if (tp.v3) unsafeAtomicAdd(gp + tp.o3, w3 * top);
              if (tp.v4) unsafeAtomicAdd(gp + tp.o4, w4 * top);
            }
          }
        }
        gm += __shfl_xor(gm, 32, 64);
        gh += __shfl_xor(gh, 32, 64);
        gw += __shfl_xor(gw, 32, 64);
        // (nt is wave-uniform here: b is)
        if (hi == 0) {
          float* ph = goffn + (size_t)(g * 18 + 2 * tap) * HW + pof;
          float* pm = gmskn + (size_t)(g * 9 + tap) * HW + pof;
          const float gmv = a.mask_logit ? gm * m * (1.f - m) : gm;
          if (a.sub == 1) {
            ph[0] = gh; ph[HW] = gw; pm[0] = gmv;
          } else {
            unsafeAtomicAdd(ph, gh); unsafeAtomicAdd(ph + HW, gw); unsafeAtomicAdd(pm, gmv);
          }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 2; ++r)
            if (t == tap && r == nt) {
#pragma unroll
              for (int cq = 0; cq < 4; ++cq) COLR(t, r, cq) = colv[cq];
            }
      }
    }
  }
  DCNB_STAMP(5);
  __syncthreads();
  DCNB_STAMP(6);
  // ---- 3. flush the gradient window
  if (gxg)
  for (int idx = tid; idx < 8 * XPX; idx += 256) {
    const long long q = (long long)s_gq[idx];
    if (q == 0) continue;
    const float v = (float)q * qinv;
    const int c = idx / XPX, r = idx - c * XPX;
    const int ry = r / XW, rx = r - ry * XW;
    const int gy_ = wy0 + ry, gx_ = wx0 + rx;
#ifdef DCNB_NOFLUSH   // (probe build, results wrong: the window's global atomics as a load on the memory system)
    asm volatile("" :: "v"(v));
    if (false)
#endif
    if ((unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W)
      unsafeAtomicAdd(gxg + (size_t)c * HW + (size_t)gy_ * a.W + gx_, v);
  }
  DCNB_STAMP(7);
  if (!a.dwp) return;
  // ---- 4. weight gradient of the tile: D[o][m] = sum_p gout[o][p] * sample[m = tap * 8 + c][p] on the MFMA with
  // K = the tile's 256 pixels, in two halves of 4 pixel rows (LDS: both operands pixel-minor with an odd row pitch, so a
  // lane's operand -- row o or m = lane & 31, pixel 2 kk + hi -- is a conflict-free ds_read_b32).  A operand rows = o,
  // B operand columns = m (72 of 96 used).  Every wave takes a quarter of a half's k-steps for all 2 x 3 tiles (the dcol
  // accumulators are dead by now); the four partial tiles are summed through LDS and leave as ONE [64][72] block of
  // plain coalesced stores per workgroup and 64-cout block (mdcn_dw_reduce_kernel sums the blocks: at ~5 k floats per
  // workgroup, atomics would be 26 M per L1 call of a batch of 8 frames).  The bias gradient falls out of the A operands.
  constexpr int HP = 129;
  float* const s_go = smem;                 // [64][HP]
  float* const s_ct = smem + 64 * HP;       // [72][HP]  (70 KB together: within the 72.6 KB of phases 0-3, two workgroups per CU)
  float* const s_out = smem;                // [64][73] (after the MFMAs)
  float* const s_db = smem + 64 * 73;       // [4][64]
  const size_t wg = ((size_t)n * nkc + kc) * ntile + tile;
#pragma unroll 1
  for (int ob = 0; ob < (a.Cout >> 6); ++ob) {
    // wave -> (32-cout half ot, half kh of a step's pixels): 3 tiles per wave, ONE exchange between the two k halves
    const int ot = wave & 1, kh = wave >> 1;
    f32x16 dw[3];
#pragma unroll
    for (int j2 = 0; j2 < 3; ++j2)
#pragma unroll
      for (int r = 0; r < 16; ++r) dw[j2][r] = 0.f;
    float db0 = 0.f;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      if constexpr (SPLIT) {
        // K = the half's 128 pixels in eight chunks of 16 (chunk c: the half's pixel row c / 2 = tile row 2 (c / 2) + half, columns
        // 16 (c & 1) ..+15); wave (ot, kh)
        // takes chunks 4 kh ..+3 for its 32 couts and all three M-tiles of the samples.
        //  * A (gout) never passes the LDS: lane (lo, hi) IS row o = 32 ot + lo, pixels 8 hi ..+7 of a chunk -- two 16-byte loads
        //    straight from global memory, split in registers (the bias gradient sums them on the way);
        //  * B (the modulated samples the sampler left in the accumulator registers) is transposed through the LDS as bf16
        //    pieces, [piece][m 72][128 pixels] with rows of 272 bytes (a lane's fragment is one ds_read_b128; sixteen
        //    consecutive rows cover the 64 banks once): 58.7 KB.
        constexpr int SP = 136;   // halfwords per row of the sample image
        unsigned short* const s_s16 = reinterpret_cast<unsigned short*>(smem);
        f32x4 ga[4][2];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int c = 4 * kh + cc;
          const int gy_ = oy0 + 2 * (c >> 1) + half, gx_ = ox0 + 16 * (c & 1) + 8 * hi;
          const float* src = gon + (size_t)(ot * 32 + lo) * HW + (size_t)gy_ * a.W + gx_;
#pragma unroll
          for (int v = 0; v < 2; ++v)
            ga[cc][v] = (gy_ < a.H && gx_ + 4 * v < a.W) ? *reinterpret_cast<const f32x4*>(src + 4 * v) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();   // the previous user of this LDS (gradient window / previous half) is done
        {
          // (a half = one pixel row of every wave -- rows half, 2 + half, 4 + half, 6 + half -- so that all four waves
          // transpose their samples at once: pixel row `wave` of the half's image)
          const bool pvh = half ? pv[1] : pv[0];
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
              const int p = wave * 32 + lo;
              unsigned h[2], m[2], l[2];
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const float c0 = half ? COLR(tap, 1, 2 * q) : COLR(tap, 0, 2 * q);
                const float c1 = half ? COLR(tap, 1, 2 * q + 1) : COLR(tap, 0, 2 * q + 1);
                dcnb_split_pair(pvh ? c0 : 0.f, pvh ? c1 : 0.f, h[q], m[q], l[q]);
              }
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const int row = tap * 8 + 4 * hi + 2 * q;
                s_s16[(0 * 72 + row) * SP + p] = (unsigned short)h[q]; s_s16[(0 * 72 + row + 1) * SP + p] = (unsigned short)(h[q] >> 16);
                s_s16[(1 * 72 + row) * SP + p] = (unsigned short)m[q]; s_s16[(1 * 72 + row + 1) * SP + p] = (unsigned short)(m[q] >> 16);
                s_s16[(2 * 72 + row) * SP + p] = (unsigned short)l[q]; s_s16[(2 * 72 + row + 1) * SP + p] = (unsigned short)(l[q] >> 16);
              }
            }
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int c = 4 * kh + cc;
          dbu4 A[3];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v0 = ga[cc][q >> 1][2 * (q & 1)], v1 = ga[cc][q >> 1][2 * (q & 1) + 1];
            db0 += v0 + v1;
            unsigned h, m, l;
            dcnb_split_pair(v0, v1, h, m, l);
            A[0][q] = h; A[1][q] = m; A[2][q] = l;
          }
          dbu4 B[3][3];   // [piece][M-tile]
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
              B[pc][mt] = (mt < 2 || lo < 8) ? *reinterpret_cast<const dbu4*>(s_s16 + (pc * 72 + mt * 32 + lo) * SP + 16 * c + 8 * hi)
                                             : dbu4{0u, 0u, 0u, 0u};
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) dw[mt] = dcnb_mma(A[0], B[0][mt], dw[mt]);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) dw[mt] = dcnb_mma(A[0], B[1][mt], dw[mt]);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) dw[mt] = dcnb_mma(A[1], B[0][mt], dw[mt]);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) dw[mt] = dcnb_mma(A[1], B[1][mt], dw[mt]);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) dw[mt] = dcnb_mma(A[0], B[2][mt], dw[mt]);
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) dw[mt] = dcnb_mma(A[2], B[0][mt], dw[mt]);
        }
        continue;
      }
      __syncthreads();   // the previous user of this LDS (gradient window / previous half / previous block) is done
      // gout half tile, zero outside the image: 64 channels x 4 rows x 32 pixels (two batches of 16 loads per lane:
      // 32 at once spill)
      if (a.vec) {
        // 64 channels x 4 rows x 8 groups of 16 bytes: 8 loads per lane in flight (round 6; the 2 x 16 scalar loads below
        // paid two memory latencies per half)
        f32x4 rg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int idx = tid + 256 * e;               // = o * 32 + row * 8 + v
          const int o = ob * 64 + (idx >> 5), row = (idx >> 3) & 3, v = idx & 7;
          const int gy_ = oy0 + 4 * half + row, gx_ = ox0 + 4 * v;
          rg[e] = (gy_ < a.H && gx_ < a.W) ? *reinterpret_cast<const f32x4*>(gon + (size_t)o * HW + (size_t)gy_ * a.W + gx_)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int idx = tid + 256 * e;
          float* d = s_go + (idx >> 5) * HP + ((idx >> 3) & 3) * 32 + 4 * (idx & 7);
          d[0] = rg[e][0]; d[1] = rg[e][1]; d[2] = rg[e][2]; d[3] = rg[e][3];
        }
      } else
#pragma unroll 1
      for (int eb = 0; eb < 32; eb += 16) {
        float rg[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int idx = tid + 256 * (eb + e);        // = o * 128 + p
          const int o = ob * 64 + (idx >> 7), p = idx & 127;
          const int gy_ = oy0 + 4 * half + (p >> 5), gx_ = ox0 + (p & 31);
          rg[e] = (gy_ < a.H && gx_ < a.W) ? gon[(size_t)o * HW + (size_t)gy_ * a.W + gx_] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int idx = tid + 256 * (eb + e);
          s_go[(idx >> 7) * HP + (idx & 127)] = rg[e];
        }
      }
      if ((wave >> 1) == half) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int p = (2 * (wave & 1) + nt) * 32 + lo;
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
              s_ct[(tap * 8 + 4 * hi + cq) * HP + p] = pv[nt] ? COLR(tap, nt, cq) : 0.f;
          }
      }
      __syncthreads();
#pragma unroll 4
      for (int kk = 32 * kh; kk < 32 * kh + 32; ++kk) {
        const int p = 2 * kk + hi;
        const float av = s_go[(ot * 32 + lo) * HP + p];
        db0 += av;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
          const float bv = (mt < 2 || lo < 8) ? s_ct[(mt * 32 + lo) * HP + p] : 0.f;   // m = mt * 32 + lo < 72
          dw[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, dw[mt], 0, 0, 0);
        }
      }
    }
    // the two k halves meet in LDS: out[o][m], m < 72
    __syncthreads();
    s_db[(2 * kh + hi) * 64 + ot * 32 + lo] = db0;
#pragma unroll 1
    for (int w = 0; w < 2; ++w) {
      if (kh == w) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, m = mt * 32 + lo;
            if (m < 72) {
              float* q = s_out + o * 73 + m;
              *q = (w == 0 ? 0.f : *q) + dw[mt][r];
            }
          }
      }
      __syncthreads();
    }
    DCNB_STAMP(8);
    float* dst = a.dwp + (wg * (size_t)(a.Cout >> 6) + ob) * (size_t)(64 * 72);
    for (int idx = tid; idx < 64 * 72; idx += 256) dst[idx] = s_out[(idx / 72) * 73 + (idx % 72)];
    if (kc == 0 && tid < 64) {
      float sdb = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) sdb += s_db[q * 64 + tid];
      a.dbp[((size_t)n * ntile + tile) * a.Cout + ob * 64 + tid] = sdb;
    }
  }
  DCNB_STAMP(9);
#undef COLR
}

// The A operands of the fused kernel's first phase in the order its LDS image has them: wtp[set][kc][(mt * KST + kk) * 64 + l]
// = w[o = 2 kk + (l >> 5)][kc * 8 + (m & 7)][tap = m >> 3] with m = mt * 32 + (l & 31) (zero for m >= 72).  One launch per call
// (147 k floats for 64 -> 64): every workgroup of the fused kernel then stages its 24 KB with six 16-byte loads per lane.
__global__ void mdcn_bwd_wt_kernel(const float* __restrict__ w, long long w_gs, float* __restrict__ wtp, int C, int Cout, int nsets) {
  const int KST = Cout >> 1, per = 3 * KST * 64, nkc = C >> 3;
  const size_t total = (size_t)nsets * nkc * per;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int idx = (int)(i % per);
    const int kc = (int)((i / per) % nkc), set = (int)(i / ((size_t)per * nkc));
    const int l = idx & 63, kk = (idx >> 6) % KST, mt = idx / (64 * KST);
    const int m = mt * 32 + (l & 31), o = 2 * kk + (l >> 5);
    wtp[i] = m < 72 ? w[(size_t)set * w_gs + ((size_t)o * C + kc * 8 + (m & 7)) * 9 + (m >> 3)] : 0.f;
  }
}

// ... and as the three bf16 pieces of the SPLIT kernel's A fragments (Cout = 64): wtp16[set][kc][piece][mt][chunk j][lane l][i]
// = piece of w[o = 16 j + 8 (l >> 5) + i][kc * 8 + (m & 7)][tap = m >> 3], m = mt * 32 + (l & 31) -- 18,432 bf16 per image.
__global__ void mdcn_bwd_wt3_kernel(const float* __restrict__ w, long long w_gs, __bf16* __restrict__ wtp, int C, int nsets) {
  constexpr int per = 3 * 4 * 64 * 8;   // values per image (each becomes three pieces)
  const int nkc = C >> 3;
  const size_t total = (size_t)nsets * nkc * per;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int idx = (int)(i % per);
    const int kc = (int)((i / per) % nkc), set = (int)(i / ((size_t)per * nkc));
    const int ii = idx & 7, l = (idx >> 3) & 63, j = (idx >> 9) & 3, mt = idx >> 11;
    const int m = mt * 32 + (l & 31), o = 16 * j + 8 * (l >> 5) + ii;
    const float v = m < 72 ? w[(size_t)set * w_gs + ((size_t)o * C + kc * 8 + (m & 7)) * 9 + (m >> 3)] : 0.f;
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 mm = (__bf16)r1;
    const __bf16 ll = (__bf16)(r1 - (float)mm);
    __bf16* d = wtp + ((size_t)set * nkc + kc) * (size_t)(3 * per) + idx;
    d[0] = h; d[per] = mm; d[2 * per] = ll;
  }
}

// dW[g][o][kc * 8 + c][tap] += sum over a chunk of the (frame, tile) rows of batch group g of
// partial[n][kc][tile][ob][o][tap * 8 + c]; db[g][o] likewise.  grid (ceil(64 * 72 / 256), C / 8 * Cout / 64, groups * nsp):
// the rows of a group are split nsp ways (one thread summing all ~1000 rows of a 180x320 call serially took 800 us for
// 170 MB); the splits meet with fp32 atomics in the zeroed outputs.
constexpr int DW_ROWS = 32;
__global__ void mdcn_dw_reduce_kernel(const float* __restrict__ dwp, const float* __restrict__ dbp, float* __restrict__ gw,
                                      float* __restrict__ gb, int N, int nkc, int ntile, int Cout, int C, int groups,
                                      int nsp, long long gw_gs, long long gb_gs) {
  const int nob = Cout >> 6;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int kc = blockIdx.y / nob, ob = blockIdx.y - kc * nob;
  const int g = blockIdx.z / nsp, sp = blockIdx.z - g * nsp;
  const int per = N / groups, rows = per * ntile;
  const int r0 = sp * DW_ROWS, r1 = min(r0 + DW_ROWS, rows);
  if (e < 64 * 72) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    auto at = [&](int r) {
      const int n = g * per + r / ntile, t = r - (r / ntile) * ntile;
      return dwp[((((size_t)n * nkc + kc) * ntile + t) * nob + ob) * (size_t)(64 * 72) + e];
    };
    int r = r0;
    for (; r + 4 <= r1; r += 4) { s0 += at(r); s1 += at(r + 1); s2 += at(r + 2); s3 += at(r + 3); }
    for (; r < r1; ++r) s0 += at(r);
    const int o = ob * 64 + e / 72, m = e % 72;
    unsafeAtomicAdd(gw + (size_t)g * gw_gs + ((size_t)o * C + kc * 8 + (m & 7)) * 9 + (m >> 3), (s0 + s1) + (s2 + s3));
  }
  if (gb && kc == 0 && e < 64) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
      const int n = g * per + r / ntile, t = r - (r / ntile) * ntile;
      s += dbp[((size_t)n * ntile + t) * Cout + ob * 64 + e];
    }
    unsafeAtomicAdd(gb + (size_t)g * gb_gs + ob * 64 + e, s);
  }
}

// Workspace: the general path's column buffer [N][C*9][P] + the 1x1 weight gradient's slots; the fused path's per-
// workgroup weight / bias gradient partials.
size_t mdcn_backward_workspace_bytes(int N, int C, int H, int W, int Cout, int stride, int pad, int dil, int groups) {
  const int Ho = (H + 2 * pad - (dil * 2 + 1)) / stride + 1, Wo = (W + 2 * pad - (dil * 2 + 1)) / stride + 1;
  const size_t col = (size_t)N * C * 9 * Ho * Wo * sizeof(float);
  const size_t ntile = (size_t)ceil_div(Wo, 32) * ceil_div(Ho, 8);
  const size_t fused = ((size_t)N * (C / 8) * ntile * Cout * 72 + (size_t)N * ntile * Cout +
                        (size_t)std::max(groups, 1) * (C / 8) * 3 * (Cout / 2) * 64 * 3 / 2) * sizeof(float);   // + the W^T images (bf16 x 3)
  return std::max(col + conv2d_wgrad_workspace_bytes(N, C * 9, Ho, Wo, Cout, 1, 1, -1, groups), fused);
}

// gout: gradient w.r.t. the PRE-activation output.  gx is accumulated into (atomics) -- zero it
// first unless other contributions are already there; goff/gmsk/gw/gb are overwritten.
int mdcn_backward_run(const float* x, const float* off, long long off_bs, const float* msk, long long msk_bs,
                      int mask_logit, const float* w, const float* gout, float* gx, float* goff,
                      long long goff_bs, float* gmsk, long long gmsk_bs, float* gw, float* gb, int N, int C,
                      int H, int W, int Cout, int stride, int pad, int dil, int dg, void* ws,
                      size_t ws_bytes, hipStream_t st, int groups, long long gw_gs, long long gb_gs, long long w_gs) {
  DVSR_REQUIRE(x && off && msk && w && gout && goff && gmsk && ws, DVSR_ERR_INVALID,
               "mdcn_backward: null pointer");
  DVSR_REQUIRE(C % dg == 0, DVSR_ERR_INVALID, "mdcn_backward: C %% dg != 0");
  if (groups < 1) groups = 1;   // > 1: one weight / bias gradient per group of N / groups batch items (gw + g * gw_gs)
  const size_t need = mdcn_backward_workspace_bytes(N, C, H, W, Cout, stride, pad, dil, groups);
  DVSR_REQUIRE(ws_bytes >= need, DVSR_ERR_WORKSPACE, "mdcn_backward: workspace %zu < %zu", ws_bytes, need);
  DcnB a;
  a.x = x; a.off = off; a.msk = msk; a.mask_logit = mask_logit;
  a.N = N; a.C = C; a.H = H; a.W = W; a.stride = stride; a.pad = pad; a.dil = dil; a.dg = dg; a.cpg = C / dg;
  a.Ho = (H + 2 * pad - (dil * 2 + 1)) / stride + 1;
  a.Wo = (W + 2 * pad - (dil * 2 + 1)) / stride + 1;
  const size_t P = (size_t)a.Ho * a.Wo;
  a.off_bs = off_bs > 0 ? off_bs : (long long)dg * 18 * P;
  a.msk_bs = msk_bs > 0 ? msk_bs : (long long)dg * 9 * P;
  if (goff_bs <= 0) goff_bs = (long long)dg * 18 * P;
  if (gmsk_bs <= 0) gmsk_bs = (long long)dg * 9 * P;
  float* col = (float*)ws;
  void* ws2 = (char*)ws + (size_t)N * C * 9 * P * sizeof(float);
  const size_t ws2_bytes = ws_bytes - (size_t)N * C * 9 * P * sizeof(float);
  static int use_fused = -1;  // DVSR_DCN_BWD=unfused: the three-kernel path for every shape (A/B aid)
  if (use_fused < 0) {
    const char* v = getenv("DVSR_DCN_BWD");
    use_fused = (v && v[0] == 'u') ? 0 : 1;
  }
  if (use_fused && a.cpg % 8 == 0 && stride == 1 && pad == 1 && dil == 1 && Cout % 64 == 0 && Cout <= 128) {
    constexpr int HALO = 4, XPX = (8 + 2 + 2 * HALO) * (32 + 2 + 2 * HALO);
    DcnF f;
    f.x = x; f.off = off; f.msk = msk; f.w = w; f.gout = gout; f.gx = gx; f.goff = goff; f.gmsk = gmsk;
    const int ntile = ceil_div(W, 32) * ceil_div(H, 8);
    f.dwp = gw ? (float*)ws : nullptr;
    f.dbp = gw ? f.dwp + (size_t)N * (C / 8) * ntile * Cout * 72 : nullptr;
    f.off_bs = a.off_bs; f.msk_bs = a.msk_bs; f.goff_bs = goff_bs; f.gmsk_bs = gmsk_bs;
    f.mask_logit = mask_logit; f.N = N; f.C = C; f.H = H; f.W = W; f.Cout = Cout; f.dg = dg;
    f.tiles_x = ceil_div(W, 32); f.tiles_y = ceil_div(H, 8);
    f.sub = a.cpg / 8;
    f.wdiv = groups > 0 ? N / groups : N; f.w_gs = w_gs;
    if (f.wdiv < 1) f.wdiv = 1;
    bool split = false;
    {
      // the W^T images live behind the weight / bias gradient partials (sized for them with or without a weight gradient)
      float* wtp = (float*)ws + (size_t)N * (C / 8) * ntile * Cout * 72 + (size_t)N * ntile * Cout;
      const int nsets = w_gs ? ceil_div(N, f.wdiv) : 1;
      DVSR_REQUIRE(nsets <= std::max(groups, 1), DVSR_ERR_INVALID, "mdcn_backward: %d weight sets for %d groups", nsets, groups);
      f.vec = (W % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)gout & 15) == 0) ? 1 : 0;
      // DVSR_DCN_BWD=fp32: the fp32-MFMA contractions for every shape (A/B aid; read once per process)
      static const bool split_on = [] { const char* v = getenv("DVSR_DCN_BWD"); return !(v && v[0] == 'f'); }();
      split = split_on && f.vec && Cout == 64;
      if (split) {
        const size_t total = (size_t)nsets * (C / 8) * (3 * 4 * 64 * 8);
        hipLaunchKernelGGL(mdcn_bwd_wt3_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1024)), dim3(256), 0, st, w, w_gs,
                           (__bf16*)wtp, C, nsets);
      } else {
        const size_t total = (size_t)nsets * (C / 8) * 3 * (Cout / 2) * 64;
        hipLaunchKernelGGL(mdcn_bwd_wt_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1024)), dim3(256), 0, st, w, w_gs,
                           wtp, C, Cout, nsets);
      }
      int rcw = check_launch("mdcn_bwd_wt_kernel");
      if (rcw) return rcw;
      f.wtp = wtp;
    }
    if (f.sub > 1 && gmsk == goff + (size_t)dg * 18 * P && goff_bs == gmsk_bs && goff_bs == (long long)dg * 27 * P) {
      // offsets and masks are the two parts of one [N, 27 dg, H, W] tensor (the engine's layout): one memset
      DVSR_REQUIRE(hipMemsetAsync(goff, 0, (size_t)N * goff_bs * sizeof(float), st) == hipSuccess, DVSR_ERR_HIP,
                   "mdcn_backward: memset of goff/gmsk failed");
    } else if (f.sub > 1) {  // the chunks of a group add their offset / mask gradients into zeroed buffers
      for (int n = 0; n < N; ++n) {
        hipError_t e1 = hipMemsetAsync(goff + (size_t)n * goff_bs, 0, (size_t)dg * 18 * P * sizeof(float), st);
        hipError_t e2 = hipMemsetAsync(gmsk + (size_t)n * gmsk_bs, 0, (size_t)dg * 9 * P * sizeof(float), st);
        DVSR_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, DVSR_ERR_HIP, "mdcn_backward: memset of goff/gmsk failed");
      }
    }
    // LDS: input window + max(gradient window, W^T operands); the weight-gradient phase re-uses it ([64 + 72][129] floats)
    constexpr int XPXA = (8 + 2 + 2 * HALO) * 48 + 8;   // the input window's rows are 48 floats (aligned 16-byte groups)
    const size_t lds = (size_t)std::max(8 * XPXA + std::max(16 * XPX, 3 * (Cout / 2) * 64), (64 + 72) * 129) * sizeof(float);
    static PerDeviceOnce attr_once;
    static PerDeviceOnce attr_once_s;
    set_dyn_lds_once(attr_once, (const void*)mdcn_bwd_fused_kernel<HALO, false>, (std::max(8 * XPXA + std::max(16 * XPX, 3 * 64 * 64), (64 + 72) * 129) * sizeof(float)));
    set_dyn_lds_once(attr_once_s, (const void*)mdcn_bwd_fused_kernel<HALO, true>, (std::max(8 * XPXA + std::max(16 * XPX, 3 * 64 * 64), (64 + 72) * 129) * sizeof(float)));
#ifdef DVSR_CONV_TRACE
    f.trace = (g_dcnb_countdown == 0) ? g_dcnb_trace : nullptr;
    if (g_dcnb_countdown >= 0) --g_dcnb_countdown;
#endif
    const dim3 grid((unsigned)(8 * (C / 8) * ceil_div(N * f.tiles_x * f.tiles_y, 8)));
    if (split) hipLaunchKernelGGL((mdcn_bwd_fused_kernel<HALO, true>), grid, dim3(256), lds, st, f);
    else hipLaunchKernelGGL((mdcn_bwd_fused_kernel<HALO, false>), grid, dim3(256), lds, st, f);
    int rc = check_launch("mdcn_bwd_fused_kernel");
    if (rc || !gw) return rc;
    // dW / db: sum the per-workgroup partials (per batch group when per-group gradients are asked for) into zeroed outputs
    for (int g = 0; g < groups; ++g) {
      hipError_t e1 = hipMemsetAsync(gw + (size_t)g * gw_gs, 0, (size_t)Cout * C * 9 * sizeof(float), st);
      hipError_t e2 = gb ? hipMemsetAsync(gb + (size_t)g * gb_gs, 0, (size_t)Cout * sizeof(float), st) : hipSuccess;
      DVSR_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, DVSR_ERR_HIP, "mdcn_backward: memset of the weight gradient failed");
    }
    const int nsp = ceil_div((N / groups) * ntile, DW_ROWS);
    hipLaunchKernelGGL(mdcn_dw_reduce_kernel, dim3(ceil_div(64 * 72, 256), (C / 8) * (Cout / 64), groups * nsp), dim3(256), 0,
                       st, f.dwp, f.dbp, gw, gb, N, C / 8, ntile, Cout, C, groups, nsp, gw_gs, gb_gs);
    return check_launch("mdcn_dw_reduce_kernel");
  }
  DVSR_REQUIRE(w_gs == 0, DVSR_ERR_UNSUPPORTED, "mdcn_backward: per-sample weights need the fused path (C/dg %% 8 == 0, 3x3, "
               "stride / pad / dilation 1, Cout %% 64 == 0)");
  // 1) dcol[n][C*9][P] = W^T . gout  as a 1x1 conv with the transposed weight view
  dvsr_conv2d_desc g = {};
  g.x0 = gout; g.w = w; g.y = col; g.N = N; g.c0 = Cout; g.H = a.Ho; g.W = a.Wo; g.Cout = C * 9; g.ks = 1;
  g.stride = 1; g.pad = 0; g.act = ACT_NONE; g.x1_bdiv = 1;
  ConvExtra ex;
  ex.wt = 1; ex.w_ctot = C * 9; ex.w_coff = 0;
  int rc = conv2d_run(g, ex, st);
  if (rc) return rc;
  // 2) offset / mask / input gradients
  const size_t total = (size_t)N * dg * 9 * P;
  const int grid = (int)((total + 255) / 256 < 65535 * 4 ? (total + 255) / 256 : 65535 * 4);
  hipLaunchKernelGGL(mdcn_col2im_coord_kernel, dim3(grid), dim3(256), 0, st, a, col, goff, goff_bs, gmsk,
                     gmsk_bs, gx);
  rc = check_launch("mdcn_col2im_coord_kernel");
  if (rc) return rc;
  // 3) col = im2col(x) (overwrites dcol), dW = gout . col^T, db = gout . 1
  if (gw) {
    hipLaunchKernelGGL(mdcn_im2col_kernel, dim3(grid), dim3(256), 0, st, a, col);
    rc = check_launch("mdcn_im2col_kernel");
    if (rc) return rc;
    rc = conv2d_wgrad_run(col, 0, 1, gout, 0, gw, gb, N, C * 9, a.Ho, a.Wo, Cout, C * 9, 0, 1, 1, ws2,
                          ws2_bytes, st, 0, -1, nullptr, groups, gw_gs, gb_gs);
  }
  return rc;
}

}  // namespace dvsr

extern "C" size_t dvsr_mdcn_backward_workspace_bytes(int N, int C, int H, int W, int Cout, int kh, int kw,
                                                     int stride, int pad, int dil) {
  (void)kh; (void)kw;
  return dvsr::mdcn_backward_workspace_bytes(N, C, H, W, Cout, stride, pad, dil);
}

extern "C" int dvsr_mdcn_backward(const float* x, const float* offset, const float* mask, const float* w,
                                  const float* grad_out, float* gx, float* goffset, float* gmask, float* gw,
                                  float* gb, int N, int C, int H, int W, int Cout, int kh, int kw, int stride,
                                  int pad, int dil, int groups, int dg, void* workspace, size_t workspace_bytes,
                                  dvsr_stream_t stream) {
  DVSR_REQUIRE(kh == 3 && kw == 3 && groups == 1, DVSR_ERR_UNSUPPORTED,
               "mdcn_backward: 3x3, groups=1 only (got %dx%d, groups=%d)", kh, kw, groups);
  return dvsr::mdcn_backward_run(x, offset, 0, mask, 0, 0, w, grad_out, gx, goffset, 0, gmask, 0, gw, gb, N, C,
                                 H, W, Cout, stride, pad, dil, dg, workspace, workspace_bytes,
                                 (hipStream_t)stream);
}
