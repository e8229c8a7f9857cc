// Fused modulated deformable convolution (DCNv2) forward for gfx950.
//
// Reference: modulated_deform_conv_cuda_forward (deform_conv_cuda.cpp:486-564) =
// zero-fill + im2col kernel (deform_conv_cuda_kernel.cu:569-632, sampler :466-496) writing a
// [C*9, Ho*Wo] column buffer to HBM + addmm + bias.  Here the column tile never leaves the CU:
// per workgroup (8x32 output pixels x 64 output channels) and per deformable group the sampled,
// mask-weighted values of 3 taps x CPG channels are written to LDS and immediately contracted
// against the matching slice of W on v_mfma_f32_32x32x2_f32 (same operand roles as conv2d.hip:
// D rows = cout, D columns = pixels).  Offsets/masks are read once per (group, tap, pixel) and
// shared by the CPG channels of the group; the x gathers hit L1/L2 (a group's planes are
// CPG*H*W*4 bytes, e.g. 1.8 MB at 180x320).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dvsr {

struct DcnK {
  const float* x; const float* off; const float* msk; const float* w; const float* bias; float* out;
  long long off_bstride, msk_bstride;  // elements between consecutive batch items
  int mask_logit;                      // 1: msk holds pre-sigmoid values (packed conv output)
  int N, C, H, W, Cout, Ho, Wo, stride, pad, dil, dg, act;
  int tiles_x, tiles_y, ntiles, ncb;
};

template <int CPG>
__global__ __launch_bounds__(256, 2) void mdcn_fwd_kernel(DcnK a) {
  constexpr int TP = 3, KK = 9, WROW = 65, NPX = 256;
  __shared__ __attribute__((aligned(16))) float s_col[CPG * TP * NPX];
  __shared__ __attribute__((aligned(16))) float s_w[CPG * KK * WROW];

  const int id = blockIdx.x;
  const int tile = (id / (8 * a.ncb)) * 8 + (id & 7);
  const int cb = (id >> 3) % a.ncb;
  if (tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * 8, ox0 = tx_ * 32;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const size_t HW = (size_t)a.H * a.W, P = (size_t)a.Ho * a.Wo;
  // this thread's pixel for the sampling phases
  const int py = oy0 + (tid >> 5), px = ox0 + (tid & 31);
  const bool pvalid = py < a.Ho && px < a.Wo;
  const size_t pofs = (size_t)py * a.Wo + px;
  const float* offn = a.off + (size_t)n * a.off_bstride;
  const float* mskn = a.msk + (size_t)n * a.msk_bstride;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int g = 0; g < a.dg; ++g) {
    __syncthreads();  // previous group's MFMAs are done with s_w / s_col
    for (int idx = tid; idx < 64 * CPG * KK; idx += 256) {
      const int o = idx / (CPG * KK);
      const int rem = idx - o * (CPG * KK);  // = c*KK + tap, contiguous in OIHW
      const int co = cb * 64 + o;
      float v = 0.f;
      if (co < a.Cout) v = a.w[((size_t)co * a.C + g * CPG) * KK + rem];
      s_w[rem * WROW + o] = v;
    }
    const float* xg = a.x + ((size_t)n * a.C + g * CPG) * HW;
    for (int t0 = 0; t0 < KK; t0 += TP) {
      if (t0) __syncthreads();  // MFMAs of the previous tap triple have consumed s_col
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const int tap = t0 + t;
        const int ki = tap / 3, kj = tap - ki * 3;
        float vals[CPG];
#pragma unroll
        for (int c = 0; c < CPG; ++c) vals[c] = 0.f;
        if (pvalid) {
          const float oh = offn[(size_t)(g * 2 * KK + 2 * tap) * P + pofs];
          const float ow = offn[(size_t)(g * 2 * KK + 2 * tap + 1) * P + pofs];
          float m = mskn[(size_t)(g * KK + tap) * P + pofs];
          if (a.mask_logit) m = sigmoidf_(m);
          const float h_im = (float)(py * a.stride - a.pad + ki * a.dil) + oh;
          const float w_im = (float)(px * a.stride - a.pad + kj * a.dil) + ow;
          DcnTap tp;
          if (make_tap(h_im, w_im, a.H, a.W, tp)) {
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
              const float* pl = xg + (size_t)c * HW;
              const float v1 = tp.v1 ? pl[tp.o1] : 0.f;
              const float v2 = tp.v2 ? pl[tp.o2] : 0.f;
              const float v3 = tp.v3 ? pl[tp.o3] : 0.f;
              const float v4 = tp.v4 ? pl[tp.o4] : 0.f;
              vals[c] = (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * m;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < CPG; ++c) s_col[(c * TP + t) * NPX + tid] = vals[c];
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const int tap = t0 + t;
#pragma unroll
        for (int kk = 0; kk < CPG / 2; ++kk) {
          const int c = 2 * kk + hi;
          const float a0 = s_w[(c * KK + tap) * WROW + lo];
          const float a1 = s_w[(c * KK + tap) * WROW + 32 + lo];
          const float b0 = s_col[(c * TP + t) * NPX + (2 * wave) * 32 + lo];
          const float b1 = s_col[(c * TP + t) * NPX + (2 * wave + 1) * 32 + lo];
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      }
    }
  }

  const TileOut t{a.out, a.bias, nullptr, a.act, 0, 0, a.Cout, a.Ho, a.Wo};
  store_mfma_tile<2, 2>(acc, t, n, cb * 64, oy0, 8, ox0, oy0 + 2 * wave, lo, hi);
}

int mdcn_forward_run(const float* x, const float* off, long long off_bs, const float* msk,
                     long long msk_bs, int mask_logit, const float* w, const float* b, float* out,
                     int N, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad,
                     int dil, int groups, int dg, int act, hipStream_t st) {
  DVSR_REQUIRE(x && off && msk && w && out, DVSR_ERR_INVALID, "mdcn_forward: null pointer");
  DVSR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cout > 0 && dg > 0 && stride > 0 && dil > 0 &&
                   pad >= 0, DVSR_ERR_INVALID, "mdcn_forward: bad dimension");
  DVSR_REQUIRE(C % dg == 0, DVSR_ERR_INVALID, "mdcn_forward: C=%d not divisible by dg=%d", C, dg);
  DVSR_REQUIRE(kh == 3 && kw == 3, DVSR_ERR_UNSUPPORTED, "mdcn_forward: kernel %dx%d (3x3 only)", kh, kw);
  DVSR_REQUIRE(groups == 1, DVSR_ERR_UNSUPPORTED, "mdcn_forward: groups=%d (1 only)", groups);
  DVSR_REQUIRE(act >= 0 && act <= 2, DVSR_ERR_INVALID, "mdcn_forward: act=%d", act);
  DcnK k;
  k.x = x; k.off = off; k.msk = msk; k.w = w; k.bias = b; k.out = out;
  k.mask_logit = mask_logit;
  k.N = N; k.C = C; k.H = H; k.W = W; k.Cout = Cout; k.stride = stride; k.pad = pad; k.dil = dil;
  k.dg = dg; k.act = act;
  k.Ho = (H + 2 * pad - (dil * 2 + 1)) / stride + 1;
  k.Wo = (W + 2 * pad - (dil * 2 + 1)) / stride + 1;
  DVSR_REQUIRE(k.Ho > 0 && k.Wo > 0, DVSR_ERR_INVALID, "mdcn_forward: empty output");
  k.off_bstride = off_bs > 0 ? off_bs : (long long)dg * 18 * k.Ho * k.Wo;
  k.msk_bstride = msk_bs > 0 ? msk_bs : (long long)dg * 9 * k.Ho * k.Wo;
  k.tiles_x = ceil_div(k.Wo, 32);
  k.tiles_y = ceil_div(k.Ho, 8);
  k.ntiles = k.tiles_x * k.tiles_y * N;
  k.ncb = ceil_div(Cout, 64);
  const int grid = ceil_div(k.ntiles, 8) * 8 * k.ncb;
  const int cpg = C / dg;
  if (cpg == 8) hipLaunchKernelGGL(mdcn_fwd_kernel<8>, dim3(grid), dim3(256), 0, st, k);
  else if (cpg == 4) hipLaunchKernelGGL(mdcn_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, k);
  else if (cpg == 16) hipLaunchKernelGGL(mdcn_fwd_kernel<16>, dim3(grid), dim3(256), 0, st, k);
  else DVSR_REQUIRE(false, DVSR_ERR_UNSUPPORTED, "mdcn_forward: C/dg=%d (supported: 4, 8, 16)", cpg);
  return check_launch("mdcn_fwd_kernel");
}

}  // namespace dvsr

extern "C" int dvsr_mdcn_forward(const float* x, const float* offset, const float* mask,
                                 const float* w, const float* b, float* out, int N, int C, int H,
                                 int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                                 int groups, int dg, int act, dvsr_stream_t stream) {
  return dvsr::mdcn_forward_run(x, offset, 0, mask, 0, 0, w, b, out, N, C, H, W, Cout, kh, kw,
                                stride, pad, dil, groups, dg, act, (hipStream_t)stream);
}

extern "C" int dvsr_mdcn_pack_forward(const float* x, const float* om, const float* w,
                                      const float* b, float* out, int N, int C, int H, int W,
                                      int Cout, int kh, int kw, int stride, int pad, int dil,
                                      int groups, int dg, int act, dvsr_stream_t stream) {
  DVSR_REQUIRE(om, DVSR_ERR_INVALID, "mdcn_pack_forward: null om");
  DVSR_REQUIRE(kh == 3 && kw == 3 && stride > 0 && dil > 0, DVSR_ERR_UNSUPPORTED,
               "mdcn_pack_forward: 3x3 only");
  const int Ho = (H + 2 * pad - (dil * 2 + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * 2 + 1)) / stride + 1;
  const long long bs = (long long)dg * 27 * Ho * Wo;
  return dvsr::mdcn_forward_run(x, om, bs, om + (size_t)dg * 18 * Ho * Wo, bs, 1, w, b, out, N, C,
                                H, W, Cout, kh, kw, stride, pad, dil, groups, dg, act,
                                (hipStream_t)stream);
}

// =================================================================================================
// LDS-resident sampler variant (engine path, 3x3 / stride 1 / pad 1 / dil 1 / CPG = 8).
//
// The r01 profile showed mdcn_fwd_kernel bound by the ISSUE rate of its global gathers (2304
// wave-level gather instructions per wave per tile; 29 TFLOP/s).  Here, per deformable group, the
// group's 8 input planes over the tile + (1 + HALO)-pixel ring are staged ONCE into LDS in a
// pixel-major [y][x][8ch] layout (zero padded outside the image, which also reproduces the
// reference's (-1,H)x(-1,W) gate: a sample outside it has all four corners outside).  A bilinear
// corner is then TWO ds_read_b128 (8 channels) instead of 8 global gathers, i.e. 8 LDS reads per
// (pixel, tap) instead of 32 VMEM gathers.  Lanes whose sample leaves the staged window
// (|offset| > HALO) fall back to clamped global gathers, so the result is exact for any offset.
// Weights come from the conv weight pack (a conv chunk of 8 channels == one deformable group) by
// LDS-DMA; the column tile and both MFMA operands use the float4-of-k layouts of conv2d_v2.hip.
// The next group's planes are register-prefetched under the MFMA phases.
// =================================================================================================
namespace dvsr {

struct DcnK2 {
  const float* x; const float* off; const float* msk; const float* wp; const float* bias; float* out;
  long long off_bstride, msk_bstride;
  int mask_logit;
  int N, C, H, W, Cout, dg, act;
  int tiles_x, tiles_y, ntiles, ncb, nchunks;
#ifdef DVSR_CONV_TRACE
  long long* trace;  // debug build only (tools/dcn_trace.py): 64 cycle stamps per workgroup
#endif
};
#ifdef DVSR_CONV_TRACE
#define DCN_STAMP(i)                                                                               \
  do {                                                                                             \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
static long long* g_dcn_trace = nullptr;
static int g_dcn_countdown = -1;
extern "C" int dvsr_debug_dcn_trace(void* buf, int launch_index) {
  g_dcn_trace = (long long*)buf;
  g_dcn_countdown = launch_index;
  return 0;
}
#else
#define DCN_STAMP(i) \
  do {               \
  } while (0)
#endif

template <int HALO>
__global__ __launch_bounds__(256, 2) void mdcn_fwd_lds_kernel(DcnK2 a) {
  constexpr int CPG = 8, KK = 9, TP = 3, NPX = 256, TH = 8, TW = 32;
  constexpr int XH = TH + 2 + 2 * HALO, XW = TW + 2 + 2 * HALO, XPX = XH * XW;
  constexpr int XE = (XPX + 255) / 256;
  constexpr int HALF = KK * 2 * 32 * 4, WF = 2 * HALF, NPIECE = WF / 256;
  __shared__ __attribute__((aligned(16))) float s_x[XPX * 8];
  __shared__ __attribute__((aligned(16))) float s_col[TP * 2 * NPX * 4];
  __shared__ __attribute__((aligned(16))) float s_w[WF];

  const int id = blockIdx.x;
  const int tile = (id / (8 * a.ncb)) * 8 + (id & 7);
  const int cb = (id >> 3) % a.ncb;
  if (tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * TW;
  const int wy0 = oy0 - 1 - HALO, wx0 = ox0 - 1 - HALO;  // image coords of the LDS window origin

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const size_t HW = (size_t)a.H * a.W;
  const int py = oy0 + (tid >> 5), px = ox0 + (tid & 31);
  const bool pvalid = py < a.H && px < a.W;
  const size_t pofs = (size_t)py * a.W + px;
  const float* offn = a.off + (size_t)n * a.off_bstride;
  const float* mskn = a.msk + (size_t)n * a.msk_bstride;

  // window elements owned by this thread (fixed for all groups)
  int xoff[XE];
  bool xok[XE];
#pragma unroll
  for (int e = 0; e < XE; ++e) {
    const int idx = tid + 256 * e;
    const int ry = idx / XW, rx = idx - ry * XW;
    const int gy = wy0 + ry, gx = wx0 + rx;
    xok[e] = idx < XPX && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    xoff[e] = xok[e] ? gy * a.W + gx : 0;
  }
  float rx_[CPG][XE];
  auto prefetch_x = [&](int g) {
    const float* xg = a.x + ((size_t)n * a.C + g * CPG) * HW;
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
      for (int e = 0; e < XE; ++e) rx_[c][e] = xg[(size_t)c * HW + xoff[e]];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* wp_cb = a.wp + (size_t)cb * a.nchunks * WF;
  prefetch_x(0);
  for (int g = 0; g < a.dg; ++g) {
    __syncthreads();  // previous group's MFMAs are done with s_w / s_col, its sampling with s_x
    // weights of this group: LDS-DMA from the conv pack (chunk g), lands before the first MFMA phase
    {
      const float* wsrc = wp_cb + (size_t)g * WF;
#pragma unroll
      for (int j = 0; j < (NPIECE + 3) / 4; ++j) {
        const int piece = j * 4 + wave;
        if (piece < NPIECE)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(wsrc + piece * 256 + lane * 4),
              (__attribute__((address_space(3))) void*)(s_w + piece * 256), 16, 0, 0);
      }
    }
    // window of this group -> LDS, pixel-major, zero outside the image
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int idx = tid + 256 * e;
      if (idx < XPX) {
        const bool ok = xok[e];
        f32x4 v0 = {ok ? rx_[0][e] : 0.f, ok ? rx_[1][e] : 0.f, ok ? rx_[2][e] : 0.f, ok ? rx_[3][e] : 0.f};
        f32x4 v1 = {ok ? rx_[4][e] : 0.f, ok ? rx_[5][e] : 0.f, ok ? rx_[6][e] : 0.f, ok ? rx_[7][e] : 0.f};
        *reinterpret_cast<f32x4*>(s_x + (size_t)idx * 8) = v0;
        *reinterpret_cast<f32x4*>(s_x + (size_t)idx * 8 + 4) = v1;
      }
    }
    __syncthreads();
    const float* xg = a.x + ((size_t)n * a.C + g * CPG) * HW;
    for (int t0 = 0; t0 < KK; t0 += TP) {
      if (t0) __syncthreads();  // MFMAs of the previous tap triple have consumed s_col
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const int tap = t0 + t;
        const int ki = tap / 3, kj = tap - ki * 3;
        float vals[CPG];
#pragma unroll
        for (int c = 0; c < CPG; ++c) vals[c] = 0.f;
        if (pvalid) {
          const float oh = offn[(size_t)(g * 2 * KK + 2 * tap) * HW + pofs];
          const float ow = offn[(size_t)(g * 2 * KK + 2 * tap + 1) * HW + pofs];
          float m = mskn[(size_t)(g * KK + tap) * HW + pofs];
          if (a.mask_logit) m = sigmoidf_(m);
          const float h_im = (float)(py - 1 + ki) + oh;
          const float w_im = (float)(px - 1 + kj) + ow;
          const float hf = floorf(h_im), wf = floorf(w_im);
          const float lh = h_im - hf, lw = w_im - wf;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
          // window-relative corner coordinates; float compare first so that huge offsets cannot overflow
          const float ryf = hf - (float)wy0, rxf = wf - (float)wx0;
          if (ryf >= 0.f && ryf <= (float)(XH - 2) && rxf >= 0.f && rxf <= (float)(XW - 2)) {
            const float* p1 = s_x + ((size_t)((int)ryf * XW + (int)rxf)) * 8;
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(p1), b1 = *reinterpret_cast<const f32x4*>(p1 + 4);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(p1 + 8), b2 = *reinterpret_cast<const f32x4*>(p1 + 12);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(p1 + XW * 8),
                        b3 = *reinterpret_cast<const f32x4*>(p1 + XW * 8 + 4);
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(p1 + XW * 8 + 8),
                        b4 = *reinterpret_cast<const f32x4*>(p1 + XW * 8 + 12);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              vals[c] = (w1 * a1[c] + w2 * a2[c] + w3 * a3[c] + w4 * a4[c]) * m;
              vals[4 + c] = (w1 * b1[c] + w2 * b2[c] + w3 * b3[c] + w4 * b4[c]) * m;
            }
          } else {
            DcnTap tp;  // sample leaves the staged window: exact clamped global gathers
            if (make_tap(h_im, w_im, a.H, a.W, tp)) {
#pragma unroll
              for (int c = 0; c < CPG; ++c) {
                const float* pl = xg + (size_t)c * HW;
                const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
                const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
                vals[c] = (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * m;
              }
            }
          }
        }
        // column tile [t][hi][pixel] x float4(kk): channel c = 2kk + hi
        f32x4 c0 = {vals[0], vals[2], vals[4], vals[6]}, c1 = {vals[1], vals[3], vals[5], vals[7]};
        *reinterpret_cast<f32x4*>(s_col + ((size_t)((t * 2 + 0) * NPX + tid)) * 4) = c0;
        *reinterpret_cast<f32x4*>(s_col + ((size_t)((t * 2 + 1) * NPX + tid)) * 4) = c1;
      }
      __syncthreads();  // (first triple: also drains the weight DMA)
      if (t0 == 0 && g + 1 < a.dg) prefetch_x(g + 1);  // in flight under the MFMA/sampling below
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const int tap = t0 + t;
        f32x4 A[2], B[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          A[mt] = *reinterpret_cast<const f32x4*>(s_w + ((size_t)(((mt * KK + tap) * 2 + hi) * 32 + lo)) * 4);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          B[nt] = *reinterpret_cast<const f32x4*>(
              s_col + ((size_t)((t * 2 + hi) * NPX + (2 * wave + nt) * 32 + lo)) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0][j], B[0][j], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0][j], B[1][j], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1][j], B[0][j], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1][j], B[1][j], acc[1][1], 0, 0, 0);
        }
      }
    }
  }

  const TileOut t{a.out, a.bias, nullptr, a.act, 0, 0, a.Cout, a.H, a.W};
  store_mfma_tile<2, 2>(acc, t, n, cb * 64, oy0, 8, ox0, oy0 + 2 * wave, lo, hi);
}

// -------------------------------------------------------------------------------------------------
// Register-direct variant: no column tile at all.  Each lane samples exactly the MFMA B operands it
// will feed: lane (lo, hi) of wave w owns pixels (row 2w+nt, col lo), nt = 0..1, and channels
// c = 2kk + hi (kk = 0..3) of the group, so the staged window is laid out [y][x][hi][kk] and a
// bilinear corner is ONE ds_read_b128 (4 channels).  Per tap: 8 corner reads + 2 weight reads +
// 16 MFMAs, no barrier; per group only the two barriers around the window / weight refill remain
// (the LDS-tile kernel above needs seven).  Offsets and masks of tap t+1 are loaded while the MFMAs
// of tap t are in flight.  LDS drops to 42.6 KB -> 3 workgroups per CU.
// -------------------------------------------------------------------------------------------------
template <int HALO, bool MASK_LOGIT>
__global__ __launch_bounds__(256, 2) void mdcn_fwd_reg_kernel(DcnK2 a) {
  constexpr int CPG = 8, KK = 9, TH = 8, TW = 32;
  constexpr int XH = TH + 2 + 2 * HALO, XW = TW + 2 + 2 * HALO, XPX = XH * XW;
  constexpr int XE = (XPX + 255) / 256;
  constexpr int HALF = KK * 2 * 32 * 4, WF = 2 * HALF, NPIECE = WF / 256;
  __shared__ __attribute__((aligned(16))) float s_x[XPX * 8];
  __shared__ __attribute__((aligned(16))) float s_w[WF];

  // an XCD (id & 7) owns a band of tile rows (same mapping as conv2d_pipe_kernel): the 16 x 40 sampling windows of
  // vertically adjacent 8 x 32 tiles overlap by half and now meet in one L2
  const int id = blockIdx.x;
  const int tpx = (a.ntiles + 7) >> 3;
  const int q_ = id >> 3;
  const int cb = q_ % a.ncb;
  const int tile = (id & 7) * tpx + q_ / a.ncb;
  if (q_ / a.ncb >= tpx || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * TW;
  const int wy0 = oy0 - 1 - HALO, wx0 = ox0 - 1 - HALO;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const size_t HW = (size_t)a.H * a.W;
  const float* offn = a.off + (size_t)n * a.off_bstride;
  const float* mskn = a.msk + (size_t)n * a.msk_bstride;
  // the two pixels this lane samples
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

  int xoff[XE];
  bool xok[XE];
#pragma unroll
  for (int e = 0; e < XE; ++e) {
    const int idx = tid + 256 * e;
    const int ry = idx / XW, rx = idx - ry * XW;
    const int gy = wy0 + ry, gx = wx0 + rx;
    xok[e] = idx < XPX && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    xoff[e] = xok[e] ? gy * a.W + gx : 0;
  }
  float rx_[CPG][XE];
  auto prefetch_x = [&](int g) {
    const float* xg = a.x + ((size_t)n * a.C + g * CPG) * HW;
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
      for (int e = 0; e < XE; ++e) rx_[c][e] = xg[(size_t)c * HW + xoff[e]];
  };
  // offsets / masks of ALL nine taps of a group are fetched together, before the window is staged: a
  // per-tap prefetch made the compiler wait for vmcnt(0) behind the sampler's branches, i.e. one full
  // memory latency per tap (1.6-2.9 k cycles of "sampling" per tap in the cycle-stamp trace)
  float oh[KK][2], ow[KK][2], mm[KK][2];  // [tap][nt]
  auto load_off = [&](int g) {
#pragma unroll
    for (int tap = 0; tap < KK; ++tap)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        oh[tap][nt] = offn[(size_t)(g * 2 * KK + 2 * tap) * HW + pofs[nt]];
        ow[tap][nt] = offn[(size_t)(g * 2 * KK + 2 * tap + 1) * HW + pofs[nt]];
        mm[tap][nt] = mskn[(size_t)(g * KK + tap) * HW + pofs[nt]];
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* wp_cb = a.wp + (size_t)cb * a.nchunks * WF;
  DCN_STAMP(0);
  // One pass per 8-channel chunk; a deformable group spans a.nchunks / a.dg of them (1 for EDVR-M's 64
  // channels, 2 for EDVR-L's 128) which share the group's offsets and masks.
  const int sub = a.nchunks / a.dg;
  for (int kc = 0; kc < a.nchunks; ++kc) {
    const int g = kc / sub;
    if (kc < 2) DCN_STAMP(1 + 30 * kc);
    prefetch_x(kc);   // no cross-group register prefetch: 24 fewer VGPRs buy the third wave per SIMD
    __syncthreads();  // previous group's taps are done with s_x / s_w
    {
      const float* wsrc = wp_cb + (size_t)kc * WF;
#pragma unroll
      for (int j = 0; j < (NPIECE + 3) / 4; ++j) {
        const int piece = j * 4 + wave;
        if (piece < NPIECE)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(wsrc + piece * 256 + lane * 4),
              (__attribute__((address_space(3))) void*)(s_w + piece * 256), 16, 0, 0);
      }
    }
    // window -> LDS as [y][x][hi][kk]: channel c = 2kk + hi
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const int idx = tid + 256 * e;
      if (idx < XPX) {
        const bool ok = xok[e];
        f32x4 v0 = {ok ? rx_[0][e] : 0.f, ok ? rx_[2][e] : 0.f, ok ? rx_[4][e] : 0.f, ok ? rx_[6][e] : 0.f};
        f32x4 v1 = {ok ? rx_[1][e] : 0.f, ok ? rx_[3][e] : 0.f, ok ? rx_[5][e] : 0.f, ok ? rx_[7][e] : 0.f};
        *reinterpret_cast<f32x4*>(s_x + (size_t)idx * 8) = v0;
        *reinterpret_cast<f32x4*>(s_x + (size_t)idx * 8 + 4) = v1;
      }
    }
    load_off(g);
    if (kc < 2) DCN_STAMP(2 + 30 * kc);
    __syncthreads();  // window visible, weight DMA drained
    if (kc < 2) DCN_STAMP(3 + 30 * kc);
    const float* xg = a.x + ((size_t)n * a.C + kc * CPG) * HW;
    // Fast sampler of one tap, BRANCH-FREE: window coordinates are clamped (always a legal LDS read), lanes
    // whose footprint leaves the window get 0 and are flagged; the exact global-gather path for them runs
    // in a rare fix-up below.  Being straight-line code, the sampler of tap t+1 shares a basic block with
    // the 16 MFMAs of tap t, so the scheduler can interleave the two (VALU/LDS next to the matrix pipe).
    auto sample = [&](int tap, f32x4 (&B)[2], bool& fix) {
      const int ki = tap / 3, kj = tap - ki * 3;
      fix = false;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float m = mm[tap][nt];
        if (MASK_LOGIT) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));  // compile-time: keeps the sampler one basic block
        const float h_im = (float)(py[nt] - 1 + ki) + oh[tap][nt];
        const float w_im = (float)(px - 1 + kj) + ow[tap][nt];
        const float hf = floorf(h_im), wf = floorf(w_im);
        const float lh = h_im - hf, lw = w_im - wf;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const int ry = (int)hf - wy0, rx = (int)wf - wx0;
        const bool inwin = ry >= 0 && ry <= XH - 2 && rx >= 0 && rx <= XW - 2;
        const int ryc = min(max(ry, 0), XH - 2), rxc = min(max(rx, 0), XW - 2);
        const float ms = (pv[nt] && inwin) ? m : 0.f;
        const float w1 = hh * hw * ms, w2 = hh * lw * ms, w3 = lh * hw * ms, w4 = lh * lw * ms;
        const float* p1 = s_x + ((size_t)(ryc * XW + rxc)) * 8 + hi * 4;
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(p1);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(p1 + 8);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(p1 + XW * 8);
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(p1 + XW * 8 + 8);
        B[nt] = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
        // outside the window AND possibly inside the image gate: needs the exact path
        fix = fix || (pv[nt] && !inwin && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W);
      }
    };
    auto fixup = [&](int tap, f32x4 (&B)[2]) {
      const int ki = tap / 3, kj = tap - ki * 3;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (!pv[nt]) continue;
        const float h_im = (float)(py[nt] - 1 + ki) + oh[tap][nt];
        const float w_im = (float)(px - 1 + kj) + ow[tap][nt];
        const int ry = (int)floorf(h_im) - wy0, rx = (int)floorf(w_im) - wx0;
        if (ry >= 0 && ry <= XH - 2 && rx >= 0 && rx <= XW - 2) continue;
        float m = mm[tap][nt];
        if (MASK_LOGIT) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));
        DcnTap tp;  // sample leaves the staged window: exact clamped global gathers
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (make_tap(h_im, w_im, a.H, a.W, tp)) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float* pl = xg + (size_t)(2 * c + hi) * HW;
            const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
            const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
            b[c] = (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * m;
          }
        }
        B[nt] = b;
      }
    };
    f32x4 Bc[2], Bn[2];
    bool fix;
    sample(0, Bc, fix);
    if (fix) fixup(0, Bc);
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      if (kc < 2) DCN_STAMP(4 + 30 * kc + 2 * tap);
      f32x4 A[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        A[mt] = *reinterpret_cast<const f32x4*>(s_w + ((size_t)(((mt * KK + tap) * 2 + hi) * 32 + lo)) * 4);
      if (tap + 1 < KK) sample(tap + 1, Bn, fix);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0][j], Bc[0][j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0][j], Bc[1][j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1][j], Bc[0][j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1][j], Bc[1][j], acc[1][1], 0, 0, 0);
      }
      if (tap + 1 < KK) {
        // ask for the interleave explicitly: MFMAs of this tap between the sampler's geometry VALU ops, its
        // 8 corner reads (+2 weight reads of the next tap), and its blend VALU ops
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
        }
      }
      if (kc < 2) DCN_STAMP(5 + 30 * kc + 2 * tap);
      if (tap + 1 < KK) {
        if (fix) fixup(tap + 1, Bn);
        Bc[0] = Bn[0]; Bc[1] = Bn[1];
      }
    }
  }

  DCN_STAMP(62);
  const TileOut t{a.out, a.bias, nullptr, a.act, 0, 0, a.Cout, a.H, a.W};
  store_mfma_tile<2, 2>(acc, t, n, cb * 64, oy0, 8, ox0, oy0 + 2 * wave, lo, hi);
  DCN_STAMP(63);
}

// wp = weights packed by pack_weights_kernel with KK=9, CC=8, wt=0 (one chunk per deformable group).
int mdcn_forward_packed_run(const float* x, const float* off, long long off_bs, const float* msk,
                            long long msk_bs, int mask_logit, const float* wp, const float* b, float* out,
                            int N, int C, int H, int W, int Cout, int dg, int act, hipStream_t st) {
  DVSR_REQUIRE(x && off && msk && wp && out, DVSR_ERR_INVALID, "mdcn_forward_packed: null pointer");
  DVSR_REQUIRE(dg > 0 && C % (dg * 8) == 0, DVSR_ERR_UNSUPPORTED,
               "mdcn_forward_packed: needs C/dg to be a multiple of 8 (got %d/%d)", C, dg);
  DcnK2 k;
  k.x = x; k.off = off; k.msk = msk; k.wp = wp; k.bias = b; k.out = out;
  k.off_bstride = off_bs; k.msk_bstride = msk_bs; k.mask_logit = mask_logit;
  k.N = N; k.C = C; k.H = H; k.W = W; k.Cout = Cout; k.dg = dg; k.act = act;
  k.tiles_x = ceil_div(W, 32); k.tiles_y = ceil_div(H, 8); k.ntiles = k.tiles_x * k.tiles_y * N;
  k.ncb = ceil_div(Cout, 64); k.nchunks = C / 8;
#ifdef DVSR_CONV_TRACE
  k.trace = (g_dcn_countdown == 0) ? g_dcn_trace : nullptr;
  if (g_dcn_countdown >= 0) --g_dcn_countdown;
#endif
  const int grid = ceil_div(k.ntiles, 8) * 8 * k.ncb;
  static int variant = -1;  // DVSR_DCN_FWD=lds selects the LDS-column-tile kernel (A/B aid)
  if (variant < 0) { const char* v = getenv("DVSR_DCN_FWD"); variant = (v && v[0] == 'l') ? 1 : 0; }
  if (variant == 1 && C == dg * 8) {
    hipLaunchKernelGGL(mdcn_fwd_lds_kernel<4>, dim3(grid), dim3(256), 0, st, k);
    return check_launch("mdcn_fwd_lds_kernel");
  }
  // (window ring: 5 and 6 pixels were measured 4-5 % slower than 4 -- the staging cost outweighs the rarer
  // fall-back)
  if (mask_logit) hipLaunchKernelGGL((mdcn_fwd_reg_kernel<4, true>), dim3(grid), dim3(256), 0, st, k);
  else hipLaunchKernelGGL((mdcn_fwd_reg_kernel<4, false>), dim3(grid), dim3(256), 0, st, k);
  return check_launch("mdcn_fwd_reg_kernel");
}

}  // namespace dvsr

// Op-level entry to the LDS-sampler kernel: packs `w` into the caller's workspace first.
// Same contract as dvsr_mdcn_forward restricted to stride = pad = dil = 1, C/dg = 8.
extern "C" size_t dvsr_mdcn_forward_fast_workspace_bytes(int C, int Cout, int dg) {
  (void)dg;
  return (size_t)dvsr::ceil_div(Cout, 64) * dvsr::ceil_div(C, 8) * dvsr::conv2_pch(3, 1) * sizeof(float);
}

extern "C" int dvsr_mdcn_forward_fast(const float* x, const float* offset, const float* mask, const float* w,
                                      const float* b, float* out, int N, int C, int H, int W, int Cout, int dg,
                                      int act, void* workspace, size_t workspace_bytes, dvsr_stream_t stream) {
  using namespace dvsr;
  DVSR_REQUIRE(x && offset && mask && w && out && workspace, DVSR_ERR_INVALID, "mdcn_forward_fast: null pointer");
  DVSR_REQUIRE(dg > 0 && C % (dg * 8) == 0, DVSR_ERR_UNSUPPORTED, "mdcn_forward_fast: needs C/dg to be a multiple of 8");
  DVSR_REQUIRE(workspace_bytes >= dvsr_mdcn_forward_fast_workspace_bytes(C, Cout, dg), DVSR_ERR_WORKSPACE,
               "mdcn_forward_fast: workspace too small");
  PackTable t;
  t.n = 1;
  PackEntry& e = t.e[0];
  e.w = w; e.P = (float*)workspace; e.Cout = Cout; e.Ctot = C; e.KK = 9; e.CC = 8; e.wt = 0; e.w_ctot = 0;
  e.w_coff = 0; e.ncb = ceil_div(Cout, 64); e.nchunks = C / 8; e.pch = conv2_pch(3, 1); e.bf = 0;
  int rc = pack_weights_run(t, (hipStream_t)stream);
  if (rc) return rc;
  const long long P = (long long)H * W;
  return mdcn_forward_packed_run(x, offset, (long long)dg * 18 * P, mask, (long long)dg * 9 * P, 0,
                                 (const float*)workspace, b, out, N, C, H, W, Cout, dg, act, (hipStream_t)stream);
}
