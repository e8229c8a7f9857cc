// Pipelined variant of the MFMA convolution used by the EDVR engine (forward and data-gradient).
//
// Same math and tiling as conv2d.hip (8x32-pixel x 64-cout tile per 4-wave workgroup,
// v_mfma_f32_32x32x2_f32, D rows = cout / columns = pixels) with the K loop software-pipelined:
//   * weights are PRE-PACKED once per forward into the exact LDS image of every (cout block,
//     channel chunk) -- [c*KK + tap][65] floats, zero padded -- by pack_weights_kernel, so staging
//     them is a straight 16-byte-per-lane LDS-DMA (global_load_lds_dwordx4): no index math, no
//     VGPRs, no ds_write;
//   * the input halo tile of chunk k+1 is fetched into registers and the weight DMA of chunk k+1
//     is issued BEFORE the MFMAs of chunk k; both LDS images are double buffered, so there is one
//     barrier per chunk and HBM/L2 latency hides under the 144 MFMAs per wave of a chunk;
//   * per-thread halo offsets / validity are computed once, not per chunk.
#include "common.h"
#include "kernels.h"

namespace dvsr {

struct ConvK2 {
  const float* x0; const float* x1; const float* wp; const float* bias; const float* res; float* y;
  int N, c0, c1, H, W, Cout, Ho, Wo, pad, act, ps, x1_bdiv;
  long long x0_bs, x1_bs;
  int tiles_x, tiles_y, ntiles, ncb, nchunks;
  int in_ps, in_dil, Hs, Ws, accum;
};

template <int KS, int S, int CC>
struct Conv2Shape {
  static constexpr int TH = 8, TW = 32, KK = KS * KS;
  static constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
  static constexpr int PLANE = IH * IW, E = (PLANE + 255) / 256;
  static constexpr int WROW = 65;
  static constexpr int IN_FLOATS = CC * PLANE;
  static constexpr int PCH = ((CC * KK * WROW + 1023) / 1024) * 1024;  // packed chunk, DMA granularity
  static constexpr int NDMA = PCH / 1024;
  static constexpr int BUF_FLOATS = ((IN_FLOATS + 3) & ~3) + PCH;
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF_FLOATS * sizeof(float);
};

int conv2_pch(int ks, int stride) {
  if (ks == 1) return Conv2Shape<1, 1, 32>::PCH;
  return stride == 2 ? Conv2Shape<3, 2, 4>::PCH : Conv2Shape<3, 1, 8>::PCH;
}
int conv2_cc(int ks, int stride) { return ks == 1 ? 32 : (stride == 2 ? 4 : 8); }

// ---- weight packing ---------------------------------------------------------------------------
// P[cb][k][(c*KK + tap)*65 + o] = W(cout = cb*64 + o, cin = k*CC + c, tap), zero outside.
// wt = 1 (data gradient): this conv's (cin, cout, tap) = original (cout, cin slice, mirrored tap).
__global__ void pack_weights_kernel(PackTable t) {
  const PackEntry& e = t.e[blockIdx.y];
  const int rows = e.CC * e.KK;
  const size_t per_chunk = (size_t)e.pch;
  const size_t total = (size_t)e.ncb * e.nchunks * per_chunk;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i % per_chunk);
    const size_t ck = i / per_chunk;
    const int k = (int)(ck % e.nchunks), cb = (int)(ck / e.nchunks);
    float v = 0.f;
    const int row = r / 65, o = r - row * 65;
    if (row < rows && o < 64) {
      const int c = row / e.KK, tap = row - c * e.KK;
      const int co = cb * 64 + o, ci = k * e.CC + c;
      if (co < e.Cout && ci < e.Ctot) {
        if (!e.wt) v = e.w[((size_t)co * e.Ctot + ci) * e.KK + tap];
        else v = e.w[((size_t)ci * e.w_ctot + e.w_coff + co) * e.KK + (e.KK - 1 - tap)];
      }
    }
    e.P[i] = v;
  }
}

int pack_weights_run(const PackTable& t, hipStream_t st) {
  if (t.n <= 0) return DVSR_OK;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(48, t.n), dim3(256), 0, st, t);
  return check_launch("pack_weights_kernel");
}

template <int KS, int S, int CC>
__global__ __launch_bounds__(256, 2) void conv2d_pipe_kernel(ConvK2 a) {
  using Sh = Conv2Shape<KS, S, CC>;
  constexpr int KK = Sh::KK, IW = Sh::IW, PLANE = Sh::PLANE, WROW = Sh::WROW, E = Sh::E;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_in0 = smem;
  float* const s_w0 = smem + ((Sh::IN_FLOATS + 3) & ~3);

  const int id = blockIdx.x;
  const int tile = (id / (8 * a.ncb)) * 8 + (id & 7);
  const int cb = (id >> 3) % a.ncb;
  if (tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * Sh::TH, ox0 = tx_ * Sh::TW;
  const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int Ctot = a.c0 + a.c1;
  const size_t HW = (size_t)a.H * a.W;
  const float* x0n = a.x0 + (size_t)n * a.x0_bs;
  const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : nullptr;
  const size_t cstride0 = a.in_dil ? (size_t)a.Hs * a.Ws : HW;  // x0 channel stride (in_ps: handled below)

  // per-thread halo elements: offset inside a channel plane + validity, fixed for all chunks
  int eoff[E];
  bool evalid[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = tid + 256 * e;
    const int iy = idx / IW, ix = idx - iy * IW;
    const int gy = iy0 + iy, gx = ix0 + ix;
    bool ok = idx < PLANE && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    int off = gy * a.W + gx;
    if (a.in_ps) off = (2 * gy) * (2 * a.W) + 2 * gx;
    if (a.in_dil) {
      ok = ok && !((gy | gx) & 1) && (gy >> 1) < a.Hs && (gx >> 1) < a.Ws;
      off = (gy >> 1) * a.Ws + (gx >> 1);
    }
    eoff[e] = ok ? off : 0;
    evalid[e] = ok;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float rin[CC][E];
  const float* wp_cb = a.wp + (size_t)cb * a.nchunks * Sh::PCH;

  auto prefetch = [&](int k, int buf) {
    const int cbase = k * CC;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const int ci = cbase + c;
      const float* src;
      if (a.in_ps) src = x0n + (size_t)(ci >> 2) * (4 * HW) + ((ci >> 1) & 1) * (2 * a.W) + (ci & 1);
      else if (ci < a.c0) src = x0n + (size_t)ci * cstride0;
      else src = x1n + (size_t)(ci - a.c0) * HW;
      const bool cok = ci < Ctot;
#pragma unroll
      for (int e = 0; e < E; ++e) rin[c][e] = (cok && evalid[e]) ? src[eoff[e]] : 0.f;
    }
    // weights of chunk k: LDS-DMA, 16 B per lane, destination = wave-uniform base + lane*16
    const float* wsrc = wp_cb + (size_t)k * Sh::PCH;
    float* wdst = s_w0 + buf * Sh::BUF_FLOATS;
#pragma unroll
    for (int j = 0; j < Sh::NDMA; ++j) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(wsrc + (j * 256 + tid) * 4),
          (__attribute__((address_space(3))) void*)(wdst + (j * 256 + wave * 64) * 4), 16, 0, 0);
    }
  };

  prefetch(0, 0);
  for (int k = 0; k < a.nchunks; ++k) {
    const int buf = k & 1;
    float* s_in = s_in0 + buf * Sh::BUF_FLOATS;
    const float* s_w = s_w0 + buf * Sh::BUF_FLOATS;
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int idx = tid + 256 * e;
        if (idx < PLANE) s_in[c * PLANE + idx] = rin[c][e];
      }
    __syncthreads();  // also drains this chunk's weight DMA (vmcnt(0) before the barrier)
    if (k + 1 < a.nchunks) prefetch(k + 1, buf ^ 1);
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      const int ty = tap / KS, tx = tap - ty * KS;
      const float* pin = s_in + ((2 * wave) * S + ty) * IW + lo * S + tx;
#pragma unroll
      for (int kk = 0; kk < CC / 2; ++kk) {
        const int c = 2 * kk + hi;
        const float a0 = s_w[(c * KK + tap) * WROW + lo];
        const float a1 = s_w[(c * KK + tap) * WROW + 32 + lo];
        const float b0 = pin[c * PLANE];
        const float b1 = pin[c * PLANE + S * IW];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
  }

  const int ox = ox0 + lo;
  if (ox >= a.Wo) return;
  const size_t HWo = (size_t)a.Ho * a.Wo;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int oy = oy0 + 2 * wave + nt;
      if (oy >= a.Ho) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cb * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co >= a.Cout) continue;
        float v = acc[mt][nt][r];
        if (a.bias) v += a.bias[co];
        v = apply_act(v, a.act);
        if (a.ps == 0) {
          const size_t o = ((size_t)n * a.Cout + co) * HWo + (size_t)oy * a.Wo + ox;
          if (a.res) v += a.res[o];
          if (a.accum) v += a.y[o];
          a.y[o] = v;
        } else {
          const int cq = co >> 2, dy = (co >> 1) & 1, dx = co & 1;
          a.y[(((size_t)n * (a.Cout >> 2) + cq) * (2 * a.Ho) + (2 * oy + dy)) * (size_t)(2 * a.Wo) +
              (2 * ox + dx)] = v;
        }
      }
    }
  }
}

template <int KS, int S, int CC>
static int launch_conv2(const ConvK2& k, hipStream_t st) {
  using Sh = Conv2Shape<KS, S, CC>;
  auto kern = conv2d_pipe_kernel<KS, S, CC>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Sh::LDS_BYTES);
    attr_done = true;
  }
  const int grid = ceil_div(k.ntiles, 8) * 8 * k.ncb;
  const size_t lds = Sh::LDS_BYTES;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, k);
  return check_launch("conv2d_pipe_kernel");
}

// `wp` = weights packed by pack_weights_kernel for this exact (ks, stride, wt) combination.
int conv2d_packed_run(const dvsr_conv2d_desc& d, const float* wp, const ConvExtra& ex, hipStream_t st) {
  DVSR_REQUIRE(d.x0 && wp && d.y, DVSR_ERR_INVALID, "conv2d_packed: null x0/wp/y");
  DVSR_REQUIRE((d.ks == 1 && d.stride == 1) || (d.ks == 3 && (d.stride == 1 || d.stride == 2)),
               DVSR_ERR_UNSUPPORTED, "conv2d_packed: ks=%d stride=%d", d.ks, d.stride);
  ConvK2 k;
  k.x0 = d.x0; k.x1 = d.x1; k.wp = wp; k.bias = d.bias; k.res = d.res; k.y = d.y;
  k.N = d.N; k.c0 = d.c0; k.c1 = d.c1; k.H = d.H; k.W = d.W; k.Cout = d.Cout;
  k.pad = d.pad; k.act = d.act; k.ps = d.pixel_shuffle; k.x1_bdiv = d.x1_bdiv > 0 ? d.x1_bdiv : 1;
  k.x0_bs = d.x0_bstride > 0 ? d.x0_bstride : (long long)d.c0 * d.H * d.W;
  k.x1_bs = d.x1_bstride > 0 ? d.x1_bstride : (long long)d.c1 * d.H * d.W;
  k.Ho = (d.H + 2 * d.pad - d.ks) / d.stride + 1;
  k.Wo = (d.W + 2 * d.pad - d.ks) / d.stride + 1;
  k.tiles_x = ceil_div(k.Wo, 32); k.tiles_y = ceil_div(k.Ho, 8); k.ntiles = k.tiles_x * k.tiles_y * d.N;
  k.ncb = ceil_div(d.Cout, 64);
  k.nchunks = ceil_div(d.c0 + d.c1, conv2_cc(d.ks, d.stride));
  k.in_ps = ex.in_ps; k.in_dil = ex.in_dil; k.Hs = ex.Hs; k.Ws = ex.Ws; k.accum = ex.accum;
  if (k.in_ps) k.x0_bs = (long long)d.c0 * d.H * d.W;
  if (k.in_dil) k.x0_bs = (long long)d.c0 * ex.Hs * ex.Ws;
  if (d.ks == 3 && d.stride == 1) return launch_conv2<3, 1, 8>(k, st);
  if (d.ks == 3) return launch_conv2<3, 2, 4>(k, st);
  return launch_conv2<1, 1, 32>(k, st);
}

}  // namespace dvsr
