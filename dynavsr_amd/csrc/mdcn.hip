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

// (DcnK2: kernels.h -- shared with mdcn_split.hip)
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
// (The first fused kernel of round 1 -- mdcn_fwd_lds_kernel: the group's window staged in LDS, a column tile sampled from it --
// lived here until round 6; the register-direct kernel below replaced it in round 2, the DMA-staged and the split kernels after
// that.  Retired with its switch value DVSR_DCN_FWD=lds.)

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

  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + (size_t)cb * a.nchunks * WF;
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
  const TileOut t{a.out, wset_ptr(a.bias, a.b_gs, n, a.wdiv), nullptr, a.act, 0, 0, a.Cout, a.H, a.W};
  store_mfma_tile<2, 2>(acc, t, n, cb * 64, oy0, 8, ox0, oy0 + 2 * wave, lo, hi);
  DCN_STAMP(63);
}


// -------------------------------------------------------------------------------------------------
// DMA-staged variant of the register-direct kernel (W % 4 == 0, 16-byte aligned x / offset / mask).
//
// mdcn_fwd_reg_kernel issues 83 vector-memory instructions per wave and 8-channel chunk (24 window loads,
// 54 offset / mask loads -- each pixel's 27 values, twice: both lane halves -- and the weight DMA) next to
// 144 MFMAs; at ~75 issue cycles apiece inside an MFMA stream (profiles/r02_z_conv_dma_ablation.txt) that is
// two thirds of the matrix time.  Here all three inputs go global -> LDS by 16-byte DMA:
//   * the sampling window: rows [oy0 - 5, oy0 + 13), the ALIGNED columns [ox0 - 8, ox0 + 40) (a ring of 4 rows,
//     7 columns), 8 channels x 18 rows x 12 groups = 27 KiB, planar [channel][row][48];
//   * the group's 27 offset / mask planes over the 8 x 32 tile: 27 x 64 groups = 27 KiB, [plane][row][32]
//     (a plane is exactly one DMA instruction; once per deformable group);
//   * the weight slice as before.
// 18 DMA instructions per wave and chunk, no staging registers.  A bilinear corner pair (x, x + 1) of one channel is
// a ds_read2_b32; offsets / masks are read from LDS in the sampler.  Groups outside the image are zeroed once (their
// lanes stay masked in every DMA).  Samples outside the window take the exact global-gather path, as before.
// -------------------------------------------------------------------------------------------------
struct DcnDmaShape {
  static constexpr int CPG = 8, KK = 9, TH = 8, TW = 32, HALO = 4, X0 = 8;
  static constexpr int XH = TH + 2 + 2 * HALO, XW = 48, XG = XW / 4, XCH = XH * XW;
  static constexpr int NXG = CPG * XH * XG, NXI = (NXG + 255) / 256;
  static constexpr int HALF = KK * 2 * 32 * 4, WF = 2 * HALF, NPIECE = WF / 256;
  static constexpr int OMP = 27, NOI = (OMP + 3) / 4;
  static constexpr int X_FLOATS = CPG * XCH, OM_FLOATS = OMP * TH * TW;
  static constexpr size_t LDS_BYTES = (size_t)(X_FLOATS + WF + OM_FLOATS) * sizeof(float);
};

template <bool MASK_LOGIT>
__global__ __launch_bounds__(256, 2) void mdcn_fwd_dma_kernel(DcnK2 a) {
  using Sh = DcnDmaShape;
  constexpr int KK = Sh::KK, TH = Sh::TH, TW = Sh::TW, XH = Sh::XH, XW = Sh::XW, XG = Sh::XG, XCH = Sh::XCH;
  extern __shared__ __attribute__((aligned(16))) float smem_dcn[];
  float* const s_x = smem_dcn;
  float* const s_w = smem_dcn + Sh::X_FLOATS;
  float* const s_om = s_w + Sh::WF;

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
  const int wy0 = oy0 - 1 - Sh::HALO, wx0 = ox0 - Sh::X0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const size_t HW = (size_t)a.H * a.W;
  const float* offn = a.off + (size_t)n * a.off_bstride;
  const float* mskn = a.msk + (size_t)n * a.msk_bstride;
  const int px = ox0 + lo;
  int py[2], prow[2];
  bool pv[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    py[nt] = oy0 + 2 * wave + nt;
    pv[nt] = py[nt] < a.H && px < a.W;
    prow[nt] = (2 * wave + nt) * TW + lo;
  }

  // window groups of this lane: L = 64 (wave + 4 jj) + lane = (channel, row, column group)
  unsigned xo[Sh::NXI];
  bool xv[Sh::NXI];
#pragma unroll
  for (int jj = 0; jj < Sh::NXI; ++jj) {
    const int L = 64 * (wave + 4 * jj) + lane;
    const int c = L / (XH * XG), r = L - c * (XH * XG);
    const int y = r / XG, g4 = r - y * XG;
    const int gy = wy0 + y, gx = wx0 + 4 * g4;
    const bool ok = L < Sh::NXG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    xo[jj] = ok ? (unsigned)(((size_t)c * HW + (size_t)gy * a.W + gx) * 4) : 0u;
    xv[jj] = ok;
    if (L < Sh::NXG && !ok) *reinterpret_cast<f32x4*>(s_x + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // offset / mask planes: plane p = wave + 4 jj is ONE instruction; this lane moves (row lane / 8, columns 4 (lane % 8) ..)
  const int omy = oy0 + (lane >> 3), omx = ox0 + 4 * (lane & 7);
  const bool omv = omy < a.H && omx < a.W;
  const unsigned omo = omv ? (unsigned)(((size_t)omy * a.W + omx) * 4) : 0u;
  if (!omv) {
#pragma unroll
    for (int jj = 0; jj < Sh::NOI; ++jj)
      if (wave + 4 * jj < Sh::OMP) *reinterpret_cast<f32x4*>(s_om + (wave + 4 * jj) * 256 + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS byte addresses of this lane's operands (inline-asm reads of the tap loop)
  auto lds_addr = [](const float* p) { return (unsigned)(size_t)((__attribute__((address_space(3))) const float*)p); };
  const unsigned a_x = lds_addr(s_x) + (unsigned)(hi * XCH) * 4u;    // channel 2 kk + hi: + 2 kk XCH floats
  const unsigned a_w = lds_addr(s_w) + (unsigned)(hi * 32 + lo) * 16u;
  unsigned a_om[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) a_om[nt] = lds_addr(s_om) + (unsigned)prow[nt] * 4u;

  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + (size_t)cb * a.nchunks * Sh::WF;
  const int sub = a.nchunks / a.dg;  // 8-channel chunks per deformable group (they share its offsets and masks)
  DCN_STAMP(0);
  for (int kc = 0; kc < a.nchunks; ++kc) {
    const int g = kc / sub;
    if (kc < 2) DCN_STAMP(1 + 30 * kc);
    __syncthreads();  // the previous chunk's taps are done with s_x / s_w / s_om
    {
      const float* wsrc = wp_cb + (size_t)kc * Sh::WF;
#pragma unroll
      for (int j = 0; j < (Sh::NPIECE + 3) / 4; ++j) {
        const int piece = j * 4 + wave;
        if (piece < Sh::NPIECE)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + piece * 256 + lane * 4),
                                           (__attribute__((address_space(3))) void*)(s_w + piece * 256), 16, 0, 0);
      }
    }
    const float* xg = a.x + ((size_t)n * a.C + kc * Sh::CPG) * HW;
    {
      const char* xb = reinterpret_cast<const char*>(xg);
#pragma unroll
      for (int jj = 0; jj < Sh::NXI; ++jj)
        if (xv[jj])
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + xo[jj]),
                                           (__attribute__((address_space(3))) void*)(s_x + 256 * (wave + 4 * jj)), 16, 0, 0);
    }
    if (kc % sub == 0) {
#pragma unroll
      for (int jj = 0; jj < Sh::NOI; ++jj) {
        const int p = wave + 4 * jj;
        if (p < Sh::OMP) {
          const float* pb = p < 18 ? offn + (size_t)(g * 18 + p) * HW : mskn + (size_t)(g * 9 + p - 18) * HW;
          if (omv)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(pb) + omo),
                                             (__attribute__((address_space(3))) void*)(s_om + 256 * p), 16, 0, 0);
        }
      }
    }
    if (kc < 2) DCN_STAMP(2 + 30 * kc);
    __syncthreads();  // everything landed (the barrier's vmcnt(0) covers the DMAs)
    if (kc < 2) DCN_STAMP(3 + 30 * kc);

    // ---- the nine taps, software-pipelined by hand ---------------------------------------------------------
    // The sampler of tap t+1 is cut into pieces that sit BETWEEN the 16 MFMAs of tap t, one piece per 64-cycle
    // MFMA.  Left to the compiler, the sampler's LDS round trips (offsets, then corners) and its ~60 geometry
    // instructions ran IN FRONT of the MFMAs of every tap: ~45 % of the matrix time exposed (70 TFLOP/s at
    // 5x64x180x320, and 80 us per call whatever the size on the small inner-step grids).  Pinning takes three
    // devices, because SelectionDAG places pure nodes by register pressure, i.e. at their use:
    //   * every VALU piece ends in an empty asm volatile that "modifies" its results, every MFMA in one on its
    //     accumulator, and sched_barrier(0) separates the slots;
    //   * the LDS reads are inline asm (a volatile load would be waited for on the spot) and their results pass
    //     through ONE "s_waitcnt lgkmcnt(0)" asm before the blends -- three MFMAs after the last read was issued.
    //     (The compiler's own lgkmcnt bookkeeping stays valid: LDS returns in order, extra reads in flight only
    //     make its waits conservative.)
    //   * operands ping-pong between two register sets by tap parity (no copies that could be hoisted above the wait).
    //   M0 M1: mask sigmoid + sample position of rows 0 / 1     M2 M3: geometry of row 0     M4 M5: its 16 corner reads
    //   M6 M7: geometry of row 1     M8 M9: its corner reads    M10: offsets / mask of tap t+2, weights of tap t+1
    //   M11 M12: blends of row 0     M13 M14: wait for the rest, blends of row 1      after M15: rare exact fix-up
    // (~160 non-MFMA instructions per tap, at most ~12 in one slot: about what a wave can issue in the shadow of a
    // 64-cycle MFMA; the cycle-stamp trace of the first version -- pieces of up to 35 instructions, 200 in all --
    // showed 1950 cycles per tap against 1060 for the sampler-free last tap, profiles/r02_z_dcn_dma_trace.txt)
    struct Geo { float w1, w2, w3, w4, h_im, w_im, m, lh, lw; int ry, rx; unsigned addr; bool inwin; };
    auto om_issue = [&](auto TAP, float (&o)[2][3]) {
      constexpr int tap = decltype(TAP)::value;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const unsigned ad = a_om[nt];  // (a local: asm operands of a generic lambda do not capture)
        f32x2 hw2;  // (dh, dw): planes 2 tap and 2 tap + 1, 1 KiB = 4 x 64 dwords apart
        asm volatile("ds_read2st64_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read_b32 %1, %2 offset:%5"
                     : "=&v"(hw2), "=&v"(o[nt][2])
                     : "v"(ad), "i"(2 * tap * 4), "i"((2 * tap + 1) * 4), "i"((18 + tap) * 1024));
        o[nt][0] = hw2[0]; o[nt][1] = hw2[1];
      }
    };
    auto a_issue = [&](auto TAP, f32x4 (&A)[2]) {
      constexpr int tap = decltype(TAP)::value;
      const unsigned ad = a_w;
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                   : "=&v"(A[0]), "=&v"(A[1]) : "v"(ad), "i"(tap * 1024), "i"((KK + tap) * 1024));
    };
    // c[4 kk + {0,1,2,3}] = (y,x) (y,x+1) (y+1,x) (y+1,x+1) of channel 2 kk + hi; half 0 = kk 0,1, half 1 = kk 2,3
    auto corners_issue = [&](unsigned addr, float (&c)[16], int half) {
      if (half == 0)
        asm volatile(
            "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:192\n\tds_read_b32 %3, %8 offset:196\n\t"
            "ds_read_b32 %4, %8 offset:6912\n\tds_read_b32 %5, %8 offset:6916\n\tds_read_b32 %6, %8 offset:7104\n\tds_read_b32 %7, %8 offset:7108"
            : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7])
            : "v"(addr));
      else
        asm volatile(
            "ds_read_b32 %0, %8 offset:13824\n\tds_read_b32 %1, %8 offset:13828\n\tds_read_b32 %2, %8 offset:14016\n\tds_read_b32 %3, %8 offset:14020\n\t"
            "ds_read_b32 %4, %8 offset:20736\n\tds_read_b32 %5, %8 offset:20740\n\tds_read_b32 %6, %8 offset:20928\n\tds_read_b32 %7, %8 offset:20932"
            : "=&v"(c[8]), "=&v"(c[9]), "=&v"(c[10]), "=&v"(c[11]), "=&v"(c[12]), "=&v"(c[13]), "=&v"(c[14]), "=&v"(c[15])
            : "v"(addr));
    };
    static_assert(XW * 4 == 192 && 2 * XCH * 4 == 6912, "corners_issue hard-codes the window pitch");
#define DCN_PIN8(c, o) asm volatile("" : "+v"(c[o]), "+v"(c[o + 1]), "+v"(c[o + 2]), "+v"(c[o + 3]), "+v"(c[o + 4]), "+v"(c[o + 5]), "+v"(c[o + 6]), "+v"(c[o + 7]))
    // row 0's corners: everything but the 15 newest LDS reads has returned (row 1's and the operand prefetch follow it)
    auto landed0 = [&](float (&c0)[16]) {
      asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c0[4]), "+v"(c0[5]), "+v"(c0[6]), "+v"(c0[7]));
      DCN_PIN8(c0, 8);
    };
    auto landed1 = [&](float (&c1)[16]) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]), "+v"(c1[4]), "+v"(c1[5]), "+v"(c1[6]), "+v"(c1[7]));
      DCN_PIN8(c1, 8);
    };
    auto landed_om = [&](float (&o)[2][3]) {
      asm volatile("" : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[0][2]), "+v"(o[1][0]), "+v"(o[1][1]), "+v"(o[1][2]));
    };
    auto landed_a = [&](f32x4 (&A)[2]) { asm volatile("" : "+v"(A[0]), "+v"(A[1])); };
    auto geom_a = [&](auto TAP, int nt, const float (&o)[3], Geo& q) {
      constexpr int tap = decltype(TAP)::value;
      constexpr int ki = tap / 3, kj = tap - ki * 3;
      float m = o[2];
      if (MASK_LOGIT) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));
      q.m = m;
      q.h_im = (float)(py[nt] - 1 + ki) + o[0];
      q.w_im = (float)(px - 1 + kj) + o[1];
      asm volatile("" : "+v"(q.m), "+v"(q.h_im), "+v"(q.w_im));
    };
    auto geom_b1 = [&](Geo& q) {
      const float hf = floorf(q.h_im), wf = floorf(q.w_im);
      q.lh = q.h_im - hf; q.lw = q.w_im - wf;
      q.ry = (int)hf - wy0; q.rx = (int)wf - wx0;
      const int ryc = min(max(q.ry, 0), XH - 2), rxc = min(max(q.rx, 0), XW - 2);
      q.addr = a_x + (unsigned)(ryc * XW + rxc) * 4u;
      asm volatile("" : "+v"(q.lh), "+v"(q.lw), "+v"(q.ry), "+v"(q.rx), "+v"(q.addr));
    };
    auto geom_b2 = [&](int nt, Geo& q, int& fix) {
      // (bitwise on purpose: a short-circuit splits the tap into basic blocks)
      const int inwin = ((unsigned)q.ry <= (unsigned)(XH - 2)) & ((unsigned)q.rx <= (unsigned)(XW - 2));
      q.inwin = inwin;
      const float hh = 1.f - q.lh, hw = 1.f - q.lw;
      const float ms = ((int)pv[nt] & inwin) ? q.m : 0.f;
      q.w1 = hh * hw * ms; q.w2 = hh * q.lw * ms; q.w3 = q.lh * hw * ms; q.w4 = q.lh * q.lw * ms;
      fix |= (int)pv[nt] & (inwin ^ 1);  // outside the window: the exact path decides (it applies the image gate itself)
      asm volatile("" : "+v"(q.w1), "+v"(q.w2), "+v"(q.w3), "+v"(q.w4), "+v"(fix));
    };
    auto blend2 = [&](const Geo& q, const float (&c)[16], int k0, f32x4& B) {
      float b0 = q.w1 * c[4 * k0] + q.w2 * c[4 * k0 + 1] + q.w3 * c[4 * k0 + 2] + q.w4 * c[4 * k0 + 3];
      float b1 = q.w1 * c[4 * k0 + 4] + q.w2 * c[4 * k0 + 5] + q.w3 * c[4 * k0 + 6] + q.w4 * c[4 * k0 + 7];
      asm volatile("" : "+v"(b0), "+v"(b1));
      B[k0] = b0; B[k0 + 1] = b1;
    };
    auto fixup = [&](const Geo (&q)[2], f32x4 (&B)[2]) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (!pv[nt] || q[nt].inwin) continue;
        DcnTap tp;  // sample leaves the staged window: exact clamped global gathers
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (make_tap(q[nt].h_im, q[nt].w_im, a.H, a.W, tp)) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float* pl = xg + (size_t)(2 * c + hi) * HW;
            const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
            const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
            b[c] = (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * q[nt].m;
          }
        }
        B[nt] = b;
      }
    };

    f32x4 Bop[2][2], Aop[2][2];  // [tap parity][pixel row | cout half]
    float om[2][2][3];           // [tap parity][pixel row][dh, dw, mask]
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    {  // tap 0 of the chunk: nothing to hide behind
      Geo gq[2];
      float c0[16], c1[16];
      int fix = 0;
      om_issue(I0{}, om[0]);
      a_issue(I0{}, Aop[0]);
      om_issue(I1{}, om[1]);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(om[0][0][0]), "+v"(om[0][0][1]), "+v"(om[0][0][2]), "+v"(om[0][1][0]), "+v"(om[0][1][1]), "+v"(om[0][1][2]));
      landed_om(om[1]); landed_a(Aop[0]);
      geom_a(I0{}, 0, om[0][0], gq[0]); geom_a(I0{}, 1, om[0][1], gq[1]);
      geom_b1(gq[0]); geom_b2(0, gq[0], fix); corners_issue(gq[0].addr, c0, 0); corners_issue(gq[0].addr, c0, 1);
      geom_b1(gq[1]); geom_b2(1, gq[1], fix); corners_issue(gq[1].addr, c1, 0); corners_issue(gq[1].addr, c1, 1);
      landed1(c0); landed1(c1);
      blend2(gq[0], c0, 0, Bop[0][0]); blend2(gq[0], c0, 2, Bop[0][0]);
      blend2(gq[1], c1, 0, Bop[0][1]); blend2(gq[1], c1, 2, Bop[0][1]);
      if (fix) fixup(gq, Bop[0]);
    }
    if (kc < 2) DCN_STAMP(4 + 30 * kc);
#define DCN_SB __builtin_amdgcn_sched_barrier(0)
#define DCN_MF(j, mt, nt)                                                                                     \
  do {                                                                                                        \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Aop[cur][mt][j], Bop[cur][nt][j], acc[mt][nt], 0, 0, 0); \
    asm volatile("" : "+v"(acc[mt][nt]));                                                                     \
    DCN_SB;                                                                                                   \
  } while (0)
    auto tap_body = [&](auto TAP) {
      constexpr int tap = decltype(TAP)::value;
      constexpr int cur = tap & 1, nxt = cur ^ 1;
      constexpr bool nx = tap + 1 < KK;
      using TN = std::integral_constant<int, (tap + 1 < KK ? tap + 1 : tap)>;
      using TNN = std::integral_constant<int, (tap + 2 < KK ? tap + 2 : tap)>;
      Geo gq[2];
      float c0[16], c1[16];
      int fix = 0;
      DCN_SB;
      if (nx) geom_a(TN{}, 0, om[nxt][0], gq[0]);
      DCN_MF(0, 0, 0);
      if (nx) geom_a(TN{}, 1, om[nxt][1], gq[1]);
      DCN_MF(0, 0, 1);
      if (nx) geom_b1(gq[0]);
      DCN_MF(0, 1, 0);
      if (nx) geom_b2(0, gq[0], fix);
      DCN_MF(0, 1, 1);
      if (nx) corners_issue(gq[0].addr, c0, 0);
      DCN_MF(1, 0, 0);
      if (nx) corners_issue(gq[0].addr, c0, 1);
      DCN_MF(1, 0, 1);
      if (nx) geom_b1(gq[1]);
      DCN_MF(1, 1, 0);
      if (nx) geom_b2(1, gq[1], fix);
      DCN_MF(1, 1, 1);
      if (nx) corners_issue(gq[1].addr, c1, 0);
      DCN_MF(2, 0, 0);
      if (nx) corners_issue(gq[1].addr, c1, 1);
      DCN_MF(2, 0, 1);
      if (tap + 2 < KK) om_issue(TNN{}, om[cur]);  // (om[cur] held this tap's values: consumed one tap ago)
      if (nx) a_issue(TN{}, Aop[nxt]);
      DCN_MF(2, 1, 0);
      if (nx) { landed0(c0); blend2(gq[0], c0, 0, Bop[nxt][0]); }
      DCN_MF(2, 1, 1);
      if (nx) blend2(gq[0], c0, 2, Bop[nxt][0]);
      DCN_MF(3, 0, 0);
      if (nx) {
        landed1(c1);
        if (tap + 2 < KK) landed_om(om[cur]);
        landed_a(Aop[nxt]);
        blend2(gq[1], c1, 0, Bop[nxt][1]);
      }
      DCN_MF(3, 0, 1);
      if (nx) blend2(gq[1], c1, 2, Bop[nxt][1]);
      DCN_MF(3, 1, 0);
      DCN_MF(3, 1, 1);
      if (nx) {
        if (fix) fixup(gq, Bop[nxt]);
      }
      if (kc < 2) DCN_STAMP(5 + 30 * kc + tap);
    };
    tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{});
    tap_body(std::integral_constant<int, 2>{}); tap_body(std::integral_constant<int, 3>{});
    tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
    tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{});
    tap_body(std::integral_constant<int, 8>{});
#undef DCN_MF
#undef DCN_SB
#undef DCN_PIN8
  }

  DCN_STAMP(62);
  const TileOut t{a.out, wset_ptr(a.bias, a.b_gs, n, a.wdiv), nullptr, a.act, 0, 0, a.Cout, a.H, a.W};
  store_mfma_tile<2, 2>(acc, t, n, cb * 64, oy0, 8, ox0, oy0 + 2 * wave, lo, hi);
  DCN_STAMP(63);
}

// wp = weights packed by pack_weights_kernel with KK=9, CC=8, wt=0 (one chunk per deformable group).
int mdcn_forward_packed_run(const float* x, const float* off, long long off_bs, const float* msk,
                            long long msk_bs, int mask_logit, const float* wp, const float* b, float* out,
                            int N, int C, int H, int W, int Cout, int dg, int act, hipStream_t st, int wdiv,
                            long long w_gs, int b_gs, int pack_perm) {
  DVSR_REQUIRE(x && off && msk && wp && out, DVSR_ERR_INVALID, "mdcn_forward_packed: null pointer");
  DVSR_REQUIRE(dg > 0 && C % (dg * 8) == 0, DVSR_ERR_UNSUPPORTED,
               "mdcn_forward_packed: needs C/dg to be a multiple of 8 (got %d/%d)", C, dg);
  DcnK2 k;
  k.x = x; k.off = off; k.msk = msk; k.wp = wp; k.bias = b; k.out = out;
  k.off_bstride = off_bs; k.msk_bstride = msk_bs; k.mask_logit = mask_logit;
  k.N = N; k.C = C; k.H = H; k.W = W; k.Cout = Cout; k.dg = dg; k.act = act;
  k.tiles_x = ceil_div(W, 32); k.tiles_y = ceil_div(H, 8); k.ntiles = k.tiles_x * k.tiles_y * N;
  k.ncb = ceil_div(Cout, 64); k.nchunks = C / 8;
  k.wdiv = wdiv > 0 ? wdiv : 1; k.w_gs = w_gs; k.b_gs = b_gs;
#ifdef DVSR_CONV_TRACE
  k.trace = (g_dcn_countdown == 0) ? g_dcn_trace : nullptr;
  if (g_dcn_countdown >= 0) --g_dcn_countdown;
#endif
  const int grid = ceil_div(k.ntiles, 8) * 8 * k.ncb;
  // DVSR_DCN_FWD: (default) the contraction on the bf16 pipe under the exact 3-way split (mdcn_split.hip; the pack is in its
  // own layout, mdcn_pack_perm()); dma = the fp32-MFMA DMA-staged kernel below, reg = its register-staged form (A/B aids; reg
  // is also what unaligned tensors run).  Read once per process: the packs and the kernels must agree.
  const int variant = mdcn_fwd_variant();
  // DMA-staged kernel when the 16-byte groups line up (DVSR_DCN_FWD=reg keeps the register-staged one, A/B aid)
  const bool aligned = W % 4 == 0 && (((uintptr_t)x | (uintptr_t)off | (uintptr_t)msk) & 15) == 0 && off_bs % 4 == 0 &&
                       msk_bs % 4 == 0;
  if (pack_perm == 6) {   // the pack is in the split kernel's layout (mdcn_pack_perm: chosen where the pack was made)
    DVSR_REQUIRE(aligned, DVSR_ERR_UNSUPPORTED, "mdcn_forward_packed: the split kernel needs W %% 4 == 0 and 16-byte aligned tensors (W=%d)", W);
    return mdcn_fwd_split_launch(k, grid, mask_logit, st);
  }
  if ((variant == 0 || variant == 3) && aligned) {
    static PerDeviceOnce attr_once_t, attr_once_f;
    set_dyn_lds_once(attr_once_t, (const void*)mdcn_fwd_dma_kernel<true>, DcnDmaShape::LDS_BYTES);
    set_dyn_lds_once(attr_once_f, (const void*)mdcn_fwd_dma_kernel<false>, DcnDmaShape::LDS_BYTES);
    if (mask_logit) hipLaunchKernelGGL(mdcn_fwd_dma_kernel<true>, dim3(grid), dim3(256), DcnDmaShape::LDS_BYTES, st, k);
    else hipLaunchKernelGGL(mdcn_fwd_dma_kernel<false>, dim3(grid), dim3(256), DcnDmaShape::LDS_BYTES, st, k);
    return check_launch("mdcn_fwd_dma_kernel");
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
  return (size_t)dvsr::ceil_div(Cout, 64) * dvsr::ceil_div(C, 8) * dvsr::mdcn_pack_floats() * sizeof(float);
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
  const bool aligned = (((uintptr_t)x | (uintptr_t)offset | (uintptr_t)mask) & 15) == 0 && ((size_t)H * W) % 4 == 0;
  e.w_coff = 0; e.ncb = ceil_div(Cout, 64); e.nchunks = C / 8; e.bf = 0;
  e.perm = aligned ? mdcn_pack_perm(W) : 0;   // (unaligned tensors: the register-staged fp32 kernel and its pack)
  e.pch = e.perm == 6 ? mdcn_pack_floats() : conv2_pch(3, 1);   // the chunk pitch of the layout (the workspace fits either)
  int rc = pack_weights_run(t, (hipStream_t)stream);
  if (rc) return rc;
  const long long P = (long long)H * W;
  return mdcn_forward_packed_run(x, offset, (long long)dg * 18 * P, mask, (long long)dg * 9 * P, 0,
                                 (const float*)workspace, b, out, N, C, H, W, Cout, dg, act, (hipStream_t)stream, 1, 0, 0, e.perm);
}
