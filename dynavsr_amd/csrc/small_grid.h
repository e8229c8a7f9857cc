// Device code of the small-grid launches (the 44x80 / 22x40 / 11x20 levels of the inner MAML step): the K-split 3x3
// convolution (forward / data gradient) and the kernel-row-split weight gradient, as __device__ item functions
// with their argument structs, so that one translation unit can host both roles.
// (Tried and dropped in r02: ONE launch per layer whose grid is [data-gradient items | weight-gradient items],
// which removes the per-layer fork event -- every event recorded on the main stream delays its next kernel by
// ~6 us -- but makes the next layer wait for the LONGER role: EDVR fwd+bwd at 44x80 6.15 ms fused vs 5.69 ms
// with the weight gradients on the side stream, gpurun_out/r02c.)
#pragma once
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace dvsr {

struct ConvK2 {
  const float* x0; const float* x1; const float* wp; const float* bias; const float* res; float* y;
  int N, c0, c1, H, W, Cout, Ho, Wo, pad, act, ps, x1_bdiv;
  long long x0_bs, x1_bs;
  int tiles_x, tiles_y, ntiles, ncb, nchunks, nitems, tiles_per_xcd;
  int in_ps, in_dil, Hs, Ws, accum;
  const float* gmask; int gmask_act;
  // per-sample weight sets (r03): batch item n uses the packed weights wp + (n / wdiv) * w_gs and the bias
  // bias + (n / wdiv) * b_gs -- K frames whose private copies of the network have DIVERGED (after their first inner
  // step; the adapted forwards) still run as one batch.  w_gs == 0: one set for the whole batch.
  int wdiv = 1; long long w_gs = 0; int b_gs = 0;
#ifdef DVSR_CONV_TRACE
  int ablate;  // DVSR_CONV_ABLATE measurement aid (conv2d_dma_kernel), debug build only
  long long* trace;  // debug build only (tools/conv_trace.py): 64 cycle stamps per workgroup
#endif
};

struct WgradK {
  const float* x; const float* gy; float* partial; float* dbp;
  long long x_bs;
  int x_bdiv;
  int N, Cin, H, W, Cout, Ho, Wo, pad, gy_ps;
  int tiles_x, tiles_y, ntiles, nsplit, nob, ncb, nslot;
  // per-group gradients (one dW per group of N / ngroups consecutive batch items: the per-FRAME gradients of a batch of
  // frames adapted from the same weights).  nsplit / nslot count per group; gtiles = tiles of one group.
  int ngroups = 1, gtiles = 0;
  int vx = 0;   // wide staging (conv2d_wgrad_wide_item): 4 / 2 = float4 / float2 x vectors (and float4 gy), 0 = scalar loads
#ifdef DVSR_CONV_TRACE
  int noflush = 0;  // measurement aid of the debug build (DVSR_WGRAD_NOFLUSH=1): skip the atomic flush, results are wrong
  long long* trace = nullptr;   // tools/wgrad_trace.py: 64 cycle stamps per workgroup (conv2d_wgrad_split3v_kernel)
#endif
};

// The tiles and the slot of pixel split `split` (block index along x, kernel-row factor removed): group g owns the splits
// [g * nsplit, (g + 1) * nsplit), walks its own tiles only and flushes into its own nslot slots.
struct WgSpan { int tile0, tile_end, slot; };
__device__ __forceinline__ WgSpan wg_span(const WgradK& a, int split) {
  int g = 0, s = split;
  if (a.ngroups > 1) { g = split / a.nsplit; s = split - g * a.nsplit; }
  return WgSpan{g * a.gtiles + s, (g + 1) * a.gtiles, g * a.nslot + s % a.nslot};
}

// host side: fill the argument structs without launching (conv2d_v2.hip / conv2d_wgrad.hip)
int conv2d_packed_prepare(const dvsr_conv2d_desc& d, const float* wp, const ConvExtra& ex, const ConvGeo& geo, ConvK2* k);
struct WgradLaunch {
  WgradK k;
  dim3 grid;
  int ks, stride, kys;
  int bf = 0;  // 1: bf16 operands on v_mfma_f32_32x32x16_bf16 (conv2d_wgrad_bf16.hip; 3x3 stride 1 only); 2: the exact
               // 3-way split of both operands on the same pipe (fp32 accuracy)
};
int conv2d_wgrad_prepare(const float* x, long long x_bs, int x_bdiv, const float* gy, int gy_ps, float* dW, float* db,
                         int N, int Cin, int H, int W, int Cout, int Ctot, int c_off, int ks, int stride, void* ws,
                         size_t ws_bytes, hipStream_t st, int scratch_is_zero, int pad, WgradReduceEntry* defer,
                         WgradLaunch* out, int bf16 = 0, int groups = 1, long long dW_gs = 0, long long db_gs = 0);
int conv2d_wgrad_launch(const WgradLaunch& l, hipStream_t st);
int conv2d_wino_launch(const ConvK2& k, int th, hipStream_t st);   // conv2d_wino.hip (ConvGeo::dma == 3)
int conv2d_wino3_launch(const ConvK2& k, int th, hipStream_t st);  // conv2d_wino3.hip (ConvGeo::dma == 4)
int conv2d_wino4_launch(const ConvK2& k, int th, hipStream_t st);  // conv2d_wino4.hip (the same image, B operand in registers)
int conv2d_wino5_launch(const ConvK2& k, int th, hipStream_t st);  // conv2d_wino5.hip (ConvGeo::dma == 5: F(4x4, 3x3), th = 8 | 16)
int conv2d_wgrad_bf16_launch(const WgradLaunch& l, hipStream_t st);
int conv2d_wgrad_split3_launch(const WgradLaunch& l, hipStream_t st);   // bf = 2: exact 3-way bf16 split (fp32 accuracy)

// -------------------------------------------------------------------------------------------------
// K-split variant for SMALL grids (the 44x80 / 22x40 / 11x20 levels of the inner MAML step).
//
// There a 3x3 64->64 layer is 33..165 four-row tiles: the kernel above leaves most CUs idle and its
// time is the serial MFMA chain of ONE workgroup (8 chunks x 36..72 MFMAs x 64 cycles = 8..15 us)
// plus prologue and epilogue, whatever the layer's FLOPs (profiles/r02a: 16 us for the 0.26 GFLOP of
// a reconstruction conv).  Here a workgroup owns a small tile -- NT pixel rows x 32 pixels x 32*MT
// output channels with MT*NT = 2 -- and its four waves split the REDUCTION: a chunk is 32 input
// channels, wave w contracts channels 8w..8w+7 of it (operand group q = w of the packed layouts), so
// the chain per wave is Ctot/32 x 9 taps x 8 MFMAs (144 for 64 channels) and a layer becomes 4x as
// many workgroups.  The partial accumulators are summed through LDS at the end.
//   * B operands: halo tile of the chunk in LDS, same [q][row][hi][x] x float4 image as above
//     (double buffered, 13 KB per buffer: ~5 workgroups per CU);
//   * A operands: never in LDS.  Each wave needs only its own 1/4 of the chunk's weights: one 16-byte
//     load per lane per (tap, 32-cout half) straight from the packed image (1 KiB per wave
//     instruction, L2-resident), two taps ahead in a ring of three register sets.
// Restricted to 3x3 / stride 1 / pad 1 / plain inputs; with two inputs the first has a multiple of 32 channels
// (the host checks).  A last chunk that runs past the input's channels re-reads its last channel against zero
// weights (the pack zero-fills).
// -------------------------------------------------------------------------------------------------
template <int MT, int NT>
struct KsShape {
  static constexpr int KK = 9, CC = 32, KQ4 = 4, TW = 32;
  static constexpr int IH = NT + 2, IW = TW + 2, PLANE = IH * IW;
  static constexpr int E = (PLANE + 127) / 128;               // plane elements per thread of a 128-thread half
  static constexpr int IN_FLOATS = KQ4 * PLANE * 8;       // [q][row][hi][x] x float4
  static constexpr int HALF = KK * KQ4 * 2 * 32 * 4;      // packed floats of one 32-cout half of a chunk
  static constexpr int RED_FLOATS = 4 * MT * NT * 16 * 64;
  static constexpr int LDS_FLOATS = 2 * IN_FLOATS > RED_FLOATS ? 2 * IN_FLOATS : RED_FLOATS;
  static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

template <int MT, int NT>
__device__ __forceinline__ void conv2d_ksplit_item(const ConvK2& a, const int id, float* const smem) {
  using Sh = KsShape<MT, NT>;
  constexpr int KK = Sh::KK, CC = Sh::CC, IH = Sh::IH, IW = Sh::IW, PLANE = Sh::PLANE;

  // same XCD-aware item order as conv2d_pipe_item: XCD (id & 7) owns a band of tile rows
  const int q_ = id >> 3;
  const int cbi = q_ % a.ncb;
  const int j_ = q_ / a.ncb;
  const int tile = (id & 7) * a.tiles_per_xcd + j_;
  if (j_ >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * NT, ox0 = tx_ * Sh::TW;
  const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int Ctot = a.c0 + a.c1;
  const size_t HW = (size_t)a.H * a.W;
  const float* x0n = a.x0 + (size_t)n * a.x0_bs;
  const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : x0n;

  // halo staging: thread = (channel half: waves 0,1 -> channels 0..15 of the chunk, waves 2,3 -> 16..31;
  // plane elements (tid & 127) + 128 m), 16 E loads in flight per thread
  constexpr int E = Sh::E;
  const int chalf = wave >> 1;
  unsigned eoffb[E];
  int elds[E];
  bool e_in[E], e_ok[E];
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const int e_ = (tid & 127) + 128 * m;
    const int eiy = e_ / IW, eix = e_ - eiy * IW;
    const int gy_ = iy0 + eiy, gx_ = ix0 + eix;
    e_in[m] = e_ < PLANE;
    e_ok[m] = e_in[m] && (unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W;
    eoffb[m] = e_ok[m] ? (unsigned)(gy_ * a.W + gx_) * 4u : 0u;
    elds[m] = (eiy * 2) * IW + eix;  // float4 index of (row, hi 0, x) inside one q-slab
  }

  float rin[16][E];
  auto issue_halo = [&](int k) {
    const int cbase = k * CC;
    const bool second = cbase >= a.c0;
    const float* b = second ? x1n : x0n;
    const int nci = second ? a.c1 : a.c0;  // channels of this input; the last chunk may run past them: those
    const int ci = (second ? cbase - a.c0 : cbase) + 16 * chalf;  // re-read the last channel (their weights are 0)
    const char* p = reinterpret_cast<const char*>(b + (size_t)(ci < nci ? ci : nci - 1) * HW);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
#pragma unroll
      for (int m = 0; m < E; ++m) rin[c][m] = *reinterpret_cast<const float*>(p + eoffb[m]);
      p += (ci + c + 1 < nci) ? HW * 4 : 0;
    }
  };
  auto write_halo = [&](int buf) {
    float* s_in = smem + buf * Sh::IN_FLOATS;
#pragma unroll
    for (int m = 0; m < E; ++m) {
      if (!e_in[m]) continue;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const bool ok = e_ok[m];
          const f32x4 v = {ok ? rin[8 * qq + h2][m] : 0.f, ok ? rin[8 * qq + 2 + h2][m] : 0.f,
                           ok ? rin[8 * qq + 4 + h2][m] : 0.f, ok ? rin[8 * qq + 6 + h2][m] : 0.f};
          *reinterpret_cast<f32x4*>(s_in + ((size_t)((2 * chalf + qq) * IH * 2 * IW) + elds[m] + h2 * IW) * 4) = v;
        }
    }
  };

  // weights of this workgroup's 32*MT-cout block; this wave's operand group (q = wave) of every tap
  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + ((size_t)((cbi * MT) >> 1) * a.nchunks * 2 + ((cbi * MT) & 1)) * Sh::HALF +
                       (size_t)(wave * 64 + lane) * 4;
  f32x4 Ag[3][MT];
  auto load_a = [&](int k, int tap, int slot) {
    const float* b = wp_cb + (size_t)k * (2 * Sh::HALF) + tap * (Sh::KQ4 * 64 * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) Ag[slot][mt] = *reinterpret_cast<const f32x4*>(b + mt * Sh::HALF);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int AD = 2;  // weight prefetch distance in taps (ring of 3 = 9 % 3 keeps the slot static)
  load_a(0, 0, 0);
  load_a(0, 1, 1);
  issue_halo(0);
  write_halo(0);
  __syncthreads();

  auto chunk = [&](int k, auto has_next_tag) {
    constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
    const float* s_in = smem + (k & 1) * Sh::IN_FLOATS + (size_t)(wave * IH * 2 * IW) * 4;
    if (HAS_NEXT) {
      issue_halo(k + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 Bv[2][NT];
    auto load_b = [&](int tap, int rb) {
      const int ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        Bv[rb][nt] = *reinterpret_cast<const f32x4*>(s_in + ((size_t)(((nt + ty) * 2 + hi) * IW + lo + tx)) * 4);
    };
    load_b(0, 0);
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      if (tap + AD < KK) load_a(k, tap + AD, (tap + AD) % 3);
      else if (HAS_NEXT) load_a(k + 1, tap + AD - KK, (tap + AD) % 3);
      if (tap + 1 < KK) load_b(tap + 1, (tap + 1) & 1);
      if (HAS_NEXT && tap == 6) {
        __builtin_amdgcn_sched_barrier(0);
        write_halo((k + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ag[tap % 3][mt][j], Bv[tap & 1][nt][j], acc[mt][nt], 0, 0, 0);
    }
    if (HAS_NEXT) __syncthreads();
  };
  for (int k = 0; k + 1 < a.nchunks; ++k) chunk(k, std::true_type{});
  chunk(a.nchunks - 1, std::false_type{});

  // ---- sum the four waves' partial tiles through LDS; wave s < MT*NT finishes sub-tile s
  __syncthreads();  // every wave is done with the halo buffers (the reduction image aliases them)
  float* red = smem;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((((wave * MT + mt) * NT + nt) * 16 + r) << 6) + lane] = acc[mt][nt][r];
  __syncthreads();
  if (wave >= MT * NT) return;
  const int smt = wave / NT, snt = wave - smt * NT;
  f32x16 sum[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[((((w * MT + smt) * NT + snt) * 16 + r) << 6) + lane];
    sum[0][0][r] = v;
  }
  const TileOut t{a.y, wset_ptr(a.bias, a.b_gs, n, a.wdiv), a.res, a.act, a.ps, a.accum, a.Cout, a.Ho, a.Wo, a.gmask, a.gmask_act};
  store_mfma_tile<1, 1>(sum, t, n, (cbi * MT + smt) * 32, oy0 + snt, 1, ox0, oy0 + snt, lo, hi);
}

// KYS (small pixel grids for 3x3; always for the 7x7 / 9x9 of TOFlow, whose 49 / 81 accumulators would not fit the
// register file): a workgroup handles ONE kernel row ky (3 of the 9 taps) -- three times as
// many workgroups, each with a third of the MFMA chain, of the accumulators (48 instead of 144 registers), of the
// flush and a two-row x tile (2 x 34 KB of LDS: two workgroups per CU).  At 44x80 a 64->64 layer is 66 (B = 1) to
// 330 (5 frames) two-row tiles: without the split 66..256 workgroups run 288 MFMAs per wave and tile.
template <int KS, bool KYS>
struct WgPipeShape {
  static constexpr int KR = KYS ? 1 : KS, NTAP = KR * KS;
  static constexpr int IW = 31 + KS, IH = 1 + KR, PLANE = IH * IW, PLANEP = PLANE | 1, GROW = 65, NPX = 64;
  static constexpr int BUF = 64 * GROW + 64 * PLANEP;
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * sizeof(float);
};

// FAST: every wave of the workgroup owns a populated 32 x 32 block (more than 32 channels on both sides of this
// (cout, cin) block: kinc == 1) -- the MFMA loop is straight-line, software-pipelined code.  The two variants are two
// separate instantiations of the WHOLE body (the kernel branches once, on block-uniform values): with both loops in
// one function the accumulators got different registers on the two paths and every tile paid ~600 v_accvgpr_mov.
template <int KS, bool KYS, bool FAST>
__device__ __forceinline__ void conv2d_wgrad_pipe_item(const WgradK& a, const int bx, const int by, const int bz,
                                                       float* const smem) {
  using Sh = WgPipeShape<KS, KYS>;
  constexpr int KK = Sh::NTAP, IW = Sh::IW, PLANE = Sh::PLANE, PLANEP = Sh::PLANEP, GROW = Sh::GROW, NPX = Sh::NPX;
  constexpr int XM = (PLANE + 63) / 64;  // wave-instructions per channel plane of the x tile
  constexpr int BUF = Sh::BUF;

  const int split = KYS ? bx / KS : bx, ky = KYS ? bx % KS : 0;
  const int ob = by, cbk = bz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  // wave -> (cout half, cin half) of the 64 x 64 block.  A half that holds no channels of the layer (first / last
  // layers: 3 or 9 inputs, 3 outputs; SpyNet: 8..32 channels) would multiply zeros: the waves that would own it
  // share the populated half instead and split the pixel reduction among them (k-steps kpar, kpar + kinc, ...;
  // the flush is atomic anyway).
  const bool cin_small = a.Cin - cbk * 64 <= 32, cout_small = a.Cout - ob * 64 <= 32;
  const int ot = cout_small ? 0 : (wave >> 1);
  const int ct = cin_small ? 0 : (wave & 1);
  const int kinc = (cin_small ? 2 : 1) * (cout_small ? 2 : 1);
  const int kpar = (cin_small && cout_small) ? wave : (cin_small ? (wave & 1) : (cout_small ? (wave >> 1) : 0));
  const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;

  // lane-fixed parts of the staging addresses
  const int gpy = lane >> 5, gpx = lane & 31;  // gy tile: lane = pixel
  const unsigned g_lane = a.gy_ps ? (unsigned)((2 * gpy) * (2 * a.Wo) + 2 * gpx) : (unsigned)(gpy * a.Wo + gpx);
  int xiy[XM], xix[XM];
#pragma unroll
  for (int m = 0; m < XM; ++m) {
    const int e = lane + 64 * m;
    xiy[m] = e / IW;
    xix[m] = e - xiy[m] * IW;
  }

  f32x16 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float db = 0.f;
  const bool do_db = cbk == 0 && (cin_small || ct == 0) && ky == 0;  // (every wave of cin half 0, each its own k-steps)

  float rg[16], rx[16][XM];
  bool g_ok, x_ok[XM];
  auto issue_loads = [&](int tile) {
    const int tx_ = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_ = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int oy0 = ty_ * 2, ox0 = tx_ * 32;
    // gy: 16 channels per wave, one pixel per lane
    g_ok = oy0 + gpy < a.Ho && ox0 + gpx < a.Wo;
    const unsigned g_off = g_ok ? g_lane * 4u : 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int co = ob * 64 + wave * 16 + j;
      co = co < a.Cout ? co : a.Cout - 1;  // clamped channels are masked at the LDS write
      const float* base;
      if (a.gy_ps)
        base = a.gy + (((size_t)n * (a.Cout >> 2) + (co >> 2)) * (2 * a.Ho) + 2 * oy0 + ((co >> 1) & 1)) *
                          (size_t)(2 * a.Wo) + 2 * ox0 + (co & 1);
      else
        base = a.gy + ((size_t)n * a.Cout + co) * HWo + (size_t)oy0 * a.Wo + ox0;
      rg[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + g_off);
    }
    // x halo: 16 channels per wave, XM x 64 plane elements per channel
    const int iy0 = oy0 - a.pad + ky, ix0 = ox0 - a.pad;
    unsigned x_off[XM];
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      const int gy_ = iy0 + xiy[m], gx_ = ix0 + xix[m];
      x_ok[m] = lane + 64 * m < PLANE && (unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W;
      x_off[m] = x_ok[m] ? (unsigned)(gy_ * a.W + gx_) * 4u : 0u;
    }
    const float* xn = a.x + (size_t)(n / a.x_bdiv) * a.x_bs;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int ci = cbk * 64 + wave * 16 + j;
      ci = ci < a.Cin ? ci : a.Cin - 1;
      const char* base = reinterpret_cast<const char*>(xn + (size_t)ci * HW);
#pragma unroll
      for (int m = 0; m < XM; ++m) rx[j][m] = *reinterpret_cast<const float*>(base + x_off[m]);
    }
  };
  auto write_lds = [&](int buf) {
    float* s_g = smem + buf * BUF;
    float* s_x = s_g + 64 * GROW;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int o = wave * 16 + j;
      s_g[o * GROW + lane] = (g_ok && ob * 64 + o < a.Cout) ? rg[j] : 0.f;
#pragma unroll
      for (int m = 0; m < XM; ++m)
        if (lane + 64 * m < PLANE) s_x[o * PLANEP + lane + 64 * m] = (x_ok[m] && cbk * 64 + o < a.Cin) ? rx[j][m] : 0.f;
    }
  };
  auto mfma_steps = [&](int buf, int k0, int k1) {
    const float* s_g = smem + buf * BUF;
    const float* s_x = s_g + 64 * GROW;
#pragma unroll 4
    for (int kk = k0 + kpar; kk < k1; kk += kinc) {
      const int p = 2 * kk + hi;
      const int py = p >> 5, px = p & 31;
      const float av = s_g[(ot * 32 + lo) * GROW + p];
      const float* bx = s_x + (ct * 32 + lo) * PLANEP + py * IW + px;
      db += av;
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const int ty = t / KS, tx = t - ty * KS;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bx[ty * IW + tx], acc[t], 0, 0, 0);
      }
    }
  };
  // The common case (every wave owns a populated 32 x 32 block: kinc == 1) as straight-line code (r03).  The loop above
  // does not unroll (run-time step), so each k-step waited for its own LDS reads in front of the MFMAs that need them --
  // four exposed LDS latencies per nine MFMAs for 3x3, one per MFMA for 1x1 (the ISA: ds_read, s_waitcnt lgkmcnt(0),
  // v_mfma; 84 TFLOP/s at best for 3x3, 15 for 1x1).  Here the k range is a compile-time constant: every operand address
  // is the lane's base plus an immediate, and the operands of k-step kk + D are read before the MFMAs of k-step kk
  // (ring of D + 1 register sets), so a read has D x KK x 64 cycles to land.
  auto mfma_fast = [&](int buf, auto k0c, auto k1c) {
    constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
    constexpr int D = KK >= 9 ? 1 : (KK >= 3 ? 2 : 6);
    const float* ga = smem + buf * BUF + (ot * 32 + lo) * GROW + hi;
    const float* xb = smem + buf * BUF + 64 * GROW + (ct * 32 + lo) * PLANEP + hi;
    float av[D + 1], bv[D + 1][KK];
    auto load = [&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      constexpr int slot = (kk - K0) % (D + 1);
      av[slot] = ga[2 * kk];
#pragma unroll
      for (int t = 0; t < KK; ++t) bv[slot][t] = xb[(kk >> 4) * IW + 2 * (kk & 15) + (t / KS) * IW + (t % KS)];
    };
    static_for<0, (D < K1 - K0 ? D : K1 - K0)>([&](auto i) { load(std::integral_constant<int, K0 + decltype(i)::value>{}); });
    static_for<K0, K1>([&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      constexpr int slot = (kk - K0) % (D + 1);
      if constexpr (kk + D < K1) load(std::integral_constant<int, kk + D>{});
      db += av[slot];
#pragma unroll
      for (int t = 0; t < KK; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[slot], bv[slot][t], acc[t], 0, 0, 0);
    });
  };
  auto mfma_range = [&](int buf, auto k0c, auto k1c) {
    if constexpr (FAST) mfma_fast(buf, k0c, k1c);
    else mfma_steps(buf, decltype(k0c)::value, decltype(k1c)::value);
  };

  const WgSpan sp = wg_span(a, split);
  int tile = sp.tile0;
  if (tile < sp.tile_end) {
    issue_loads(tile);
    write_lds(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < sp.tile_end; tile += a.nsplit) {
    const bool has_next = tile + a.nsplit < sp.tile_end;
#ifdef DVSR_CONV_TRACE   // ablations of the debug build (DVSR_WGRAD_NOFLUSH bits 1 / 2 / 3: no global loads / no LDS writes / no barrier)
    if (has_next && !(a.noflush & 2)) issue_loads(tile + a.nsplit);
    mfma_range(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, 3 * NPX / 8>{});
    if (has_next && !(a.noflush & 4)) write_lds(buf ^ 1);
    mfma_range(buf, std::integral_constant<int, 3 * NPX / 8>{}, std::integral_constant<int, NPX / 2>{});
    if (!(a.noflush & 8)) __syncthreads();
#else
    if (has_next) issue_loads(tile + a.nsplit);
    mfma_range(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, 3 * NPX / 8>{});
    if (has_next) write_lds(buf ^ 1);
    mfma_range(buf, std::integral_constant<int, 3 * NPX / 8>{}, std::integral_constant<int, NPX / 2>{});
    __syncthreads();
#endif
    buf ^= 1;
  }

  // ---- partial[slot][tap][o][c]  (o, c padded to the 64-blocks of the grid), slot = split % nslot
  const int OP = a.nob * 64, CP = a.ncb * 64;
  const int slot = sp.slot;
#ifdef DVSR_CONV_TRACE
  if ((a.noflush & 1) && acc[0][0] != 12345.f) return;
#endif
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * 64 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int c = cbk * 64 + ct * 32 + lo;
      unsafeAtomicAdd(a.partial + (((size_t)slot * (KS * KS) + ky * KS + t) * OP + o) * CP + c, acc[t][r]);
    }
  // lane (lo, hi) summed gy[o = ot*32 + lo] over the pixels of parity hi
  if (do_db) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + ob * 64 + ot * 32 + lo, db);
}

// -------------------------------------------------------------------------------------------------
// Wide staging (r03).  Ablations of the debug build (DVSR_WGRAD_NOFLUSH bits, profiles/r03_b_wgrad_ablation.txt) price
// the staging of the kernel above: with its 64 global_load_dword per lane and tile compiled out the 40x64x176x320 layer
// runs at 122 instead of 95 TFLOP/s, without its 64 ds_write_b32 at 106, without both at 140 -- the MFMA loop itself is
// at 0.89 of the peak.  Here both operands move as VECTORS whenever the tensors allow it:
//   gy   float4 along a row (Wo % 4 == 0, 16-byte aligned): 4 loads + 4 ds_write_b128 per lane and tile instead of 16 + 16;
//   x    VX = 4: the tile's window widened to the 16-byte aligned columns [ox0 - pad - s, ..) (pad 1: [ox0 - 4, ox0 + 36),
//        ten float4 per row -- every group lies entirely inside or outside the image since W % 4 == 0): 10 loads + 10
//        ds_write_b128 instead of 48 + 48;  VX = 2: float2 (the estimator's explicitly padded tensors: even pitch, pad 0).
// The LDS images keep the window as it was loaded ([channel][row][WWIN]), so a vector is ONE aligned LDS write; the
// channel pitch is an odd number of vectors: lanes (= channels) of an operand read spread over 32 / VX banks, a VX-way
// conflict that 10 reads per 9 MFMAs do not notice.  Every wave owns a populated 32 x 32 block (the FAST case above).
// -------------------------------------------------------------------------------------------------
template <int KS, bool KYS, int VX>
struct WgWideShape {
  static constexpr int KR = KYS ? 1 : KS, NTAP = KR * KS, IW = 31 + KS, IH = 1 + KR, NPX = 64;
  static constexpr int WWIN = ((IW + 2 * (VX - 1)) / VX) * VX;   // window floats per row: start rounded down, end up
  static constexpr int RV = WWIN / VX;                           // vectors per row
  static constexpr int XV = IH * RV;                             // vectors per channel
  static constexpr int PX = ((XV % 2) ? XV : XV + 1) * VX;       // channel pitch: an odd number of vectors
  static constexpr int GP = 68;                                  // gy: 16 float4 per channel + 1
  static constexpr int BUF = 64 * GP + 64 * PX;
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * sizeof(float);
};

template <int KS, bool KYS, int VX>
__device__ __forceinline__ void conv2d_wgrad_wide_item(const WgradK& a, const int bx, const int by, const int bz,
                                                       float* const smem) {
  using Sh = WgWideShape<KS, KYS, VX>;
  typedef float xvec __attribute__((ext_vector_type(VX)));
  constexpr int KK = Sh::NTAP, WWIN = Sh::WWIN, RV = Sh::RV, XV = Sh::XV, PX = Sh::PX, GP = Sh::GP, NPX = Sh::NPX, BUF = Sh::BUF;
  constexpr int XM = (16 * XV + 63) / 64;   // x vectors per lane and tile (16 channels per wave)

  const int split = KYS ? bx / KS : bx, ky = KYS ? bx % KS : 0;
  const int ob = by, cbk = bz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int ot = wave >> 1, ct = wave & 1;
  const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;
  const int shift = (VX - a.pad % VX) % VX;   // the window starts `shift` columns left of the first one the taps need

  // lane-fixed parts of the staging: gy vector m = (channel gch, row, group); x vector m = (channel xch, row, group)
  int gch[4], grow[4], gcol[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int idx = lane + 64 * m;
    gch[m] = idx >> 4; grow[m] = (idx >> 3) & 1; gcol[m] = (idx & 7) * 4;
  }
  int xch[XM], xrow[XM], xcol[XM], xlds[XM];
#pragma unroll
  for (int m = 0; m < XM; ++m) {
    const int idx = lane + 64 * m;
    const int c = idx / XV, r = idx - c * XV;
    xch[m] = c < 16 ? c : 15; xrow[m] = r / RV; xcol[m] = (r - xrow[m] * RV) * VX;
    xlds[m] = c < 16 ? (wave * 16 + c) * PX + r * VX : -1;
  }

  f32x16 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float db = 0.f;
  const bool do_db = cbk == 0 && ct == 0 && ky == 0;

  f32x4 rg[4];
  xvec rx[XM];
  bool g_ok[4], x_ok[XM];
  auto issue_loads = [&](int tile) {
    const int tx_ = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_ = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int oy0 = ty_ * 2, ox0 = tx_ * 32;
    // (an invalid vector reads the first one of the image instead: channels past Cout / Cin of a partly filled block
    // would otherwise address memory behind the tensor)
    const float* gn = a.gy + (size_t)n * a.Cout * HWo;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int co = ob * 64 + wave * 16 + gch[m];
      g_ok[m] = co < a.Cout && oy0 + grow[m] < a.Ho && ox0 + gcol[m] < a.Wo;
      const size_t off = g_ok[m] ? (size_t)co * HWo + (size_t)(oy0 + grow[m]) * a.Wo + ox0 + gcol[m] : 0;
      rg[m] = *reinterpret_cast<const f32x4*>(gn + off);
    }
    const int iy0 = oy0 - a.pad + ky, ix0 = ox0 - a.pad - shift;
    const float* xn = a.x + (size_t)(n / a.x_bdiv) * a.x_bs;
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      const int ci = cbk * 64 + wave * 16 + xch[m];
      const int gy_ = iy0 + xrow[m], gx_ = ix0 + xcol[m];
      x_ok[m] = xlds[m] >= 0 && ci < a.Cin && (unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W;
      const size_t off = x_ok[m] ? (size_t)ci * HW + (size_t)gy_ * a.W + gx_ : 0;
      rx[m] = *reinterpret_cast<const xvec*>(xn + off);
    }
  };
  auto write_lds = [&](int buf) {
    float* s_g = smem + buf * BUF;
    float* s_x = s_g + 64 * GP;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 v = g_ok[m] ? rg[m] : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(s_g + (wave * 16 + gch[m]) * GP + grow[m] * 32 + gcol[m]) = v;
    }
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      xvec v = rx[m];
      if (!x_ok[m]) {
#pragma unroll
        for (int e = 0; e < VX; ++e) v[e] = 0.f;
      }
      if (xlds[m] >= 0) *reinterpret_cast<xvec*>(s_x + xlds[m]) = v;
    }
  };
  auto mfma_fast = [&](int buf, auto k0c, auto k1c) {
    constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
    constexpr int D = KK >= 9 ? 1 : (KK >= 3 ? 2 : 6);
    const float* ga = smem + buf * BUF + (ot * 32 + lo) * GP + hi;
    const float* xb = smem + buf * BUF + 64 * GP + (ct * 32 + lo) * PX + hi + shift;
    float av[D + 1], bv[D + 1][KK];
    auto load = [&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      constexpr int slot = (kk - K0) % (D + 1);
      av[slot] = ga[2 * kk];
#pragma unroll
      for (int t = 0; t < KK; ++t) bv[slot][t] = xb[(kk >> 4) * WWIN + 2 * (kk & 15) + (t / KS) * WWIN + (t % KS)];
    };
    static_for<0, (D < K1 - K0 ? D : K1 - K0)>([&](auto i) { load(std::integral_constant<int, K0 + decltype(i)::value>{}); });
    static_for<K0, K1>([&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      constexpr int slot = (kk - K0) % (D + 1);
      if constexpr (kk + D < K1) load(std::integral_constant<int, kk + D>{});
      db += av[slot];
#pragma unroll
      for (int t = 0; t < KK; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[slot], bv[slot][t], acc[t], 0, 0, 0);
    });
  };

  const WgSpan sp = wg_span(a, split);
  int tile = sp.tile0;
  if (tile < sp.tile_end) {
    issue_loads(tile);
    write_lds(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < sp.tile_end; tile += a.nsplit) {
    const bool has_next = tile + a.nsplit < sp.tile_end;
    if (has_next) issue_loads(tile + a.nsplit);
    mfma_fast(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, 3 * NPX / 8>{});
    if (has_next) write_lds(buf ^ 1);
    mfma_fast(buf, std::integral_constant<int, 3 * NPX / 8>{}, std::integral_constant<int, NPX / 2>{});
    __syncthreads();
    buf ^= 1;
  }

  const int OP = a.nob * 64, CP = a.ncb * 64;
  const int slot = sp.slot;
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * 64 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int c = cbk * 64 + ct * 32 + lo;
      unsafeAtomicAdd(a.partial + (((size_t)slot * (KS * KS) + ky * KS + t) * OP + o) * CP + c, acc[t][r]);
    }
  if (do_db) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + ob * 64 + ot * 32 + lo, db);
}

}  // namespace dvsr
