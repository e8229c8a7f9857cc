// Weight gradient of the 3x3 stride-1 convolutions on the bf16 MFMA (network_G.bf16_mfma = 1, BASELINE configs[4]).
//
// With the forward and data-gradient convolutions on v_mfma_f32_32x32x16_bf16 the fp32 weight gradient was 51 % of the
// GPU time of an EDVR-L forward+backward (profiles: conv2d_wgrad_pipe_kernel<3,true> 82 us x 128 launches of a
// 16.7 ms step).  Same decomposition as conv2d_wgrad.hip -- D[o 64][c 64] per tap with K = pixels, each wave one
// (o-half, c-half) 32x32 tile for all 9 taps, workgroups walk strided 2x32-pixel tiles, slot flush with fp32 atomics --
// but both operands are rounded to bf16 when they are written to LDS and a (tap, 16-pixel block) is ONE MFMA
// (32 cycles) instead of eight fp32 MFMAs (512): 36 MFMAs per tile and wave instead of 288.
//   A (lane l): gy[o = l&31][pixels 16kb + 8(l>>5) .. +7]          <- s_g[o][px]  bf16, rows padded to 144 B
//   B (lane l): x [c = l&31][the same pixels shifted by the tap]   <- s_x[tx][c][row][px] bf16: THREE copies of the
//       tile, shifted by tx = 0, 1, 2 columns, so that every 16-byte operand read is aligned (channel stride 272 B:
//       both images are conflict-free for ds_read_b128).
// The kernel is bound by staging, not by the matrix pipe: per tile a wave loads 16 channels of both operands (64
// loads per lane, issued for tile i+1 before the MFMAs of tile i), converts and writes them (three 4-byte LDS writes per
// lane and channel for the x copies: lanes hold column pairs, the neighbour's pair comes by DPP).  Accumulation and the
// flush stay fp32; the bias gradient is the fp32 sum of the (bf16-rounded) A operands.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x2 __attribute__((ext_vector_type(2)));

constexpr int WB_GROW = 72;            // bf16 per gy row: 64 pixels + 8 pad (144 B)
constexpr int WB_XCH = 136;            // bf16 per channel of one x copy: 4 rows x 32 + 8 pad (272 B)
constexpr int WB_XCOPY = 64 * WB_XCH;  // one shifted copy
constexpr size_t WB_LDS_BYTES = (size_t)(64 * WB_GROW + 3 * WB_XCOPY) * 2;

// Staging map of the x tile (4 rows x 34 columns per channel): lane = (row = lane >> 4, pair p = lane & 15) holds columns
// 2p, 2p+1 (ONE 8-byte load); the two halo columns 32, 33 of all 16 channels x 4 rows of a wave are one more 8-byte load
// (lane = channel*4 + row).  The three shifted copies are then three 4-byte LDS writes per lane and channel:
//   copy 0 [2p, 2p+1] = (v0, v1)             copy 1 = (v1, next v0)             copy 2 = (next v0, next v1)
// with `next` = the pair of lane + 1 (a DPP row shift inside the 16-lane row) or, for p = 15, the halo pair.
__device__ __forceinline__ float dpp_next(float v) {   // value of lane + 1 within its row of 16 lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101 /* row_shl:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(256, 1) void conv2d_wgrad_bf16_kernel(WgradK a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
  __bf16* const s_g = smem16;
  __bf16* const s_x = smem16 + 64 * WB_GROW;

  const int split = blockIdx.x, ob = blockIdx.y, cbk = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int ot = wave >> 1, ct = wave & 1;
  const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;

  const int gpy = lane >> 5, gpx = lane & 31;  // gy tile: lane = pixel
  const unsigned g_lane = a.gy_ps ? (unsigned)((2 * gpy) * (2 * a.Wo) + 2 * gpx) : (unsigned)(gpy * a.Wo + gpx);
  const int xrow = lane >> 4, xp = lane & 15;       // x tile: lane = (row, column pair)
  const int trow = lane & 3, tch = lane >> 2;       // halo pair: lane = (channel of the wave, row)

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float db = 0.f;

  float rg[16];
  float2 rx[16], rt;
  bool g_ok, x_ok0, x_ok1, t_ok0, t_ok1;
  auto issue_loads = [&](int tile) {
    const int tx_ = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_ = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int oy0 = ty_ * 2, ox0 = tx_ * 32;
    g_ok = oy0 + gpy < a.Ho && ox0 + gpx < a.Wo;
    const unsigned g_off = g_ok ? g_lane * 4u : 0u;
    {  // one scalar base pointer walked over the wave's 16 channels (channels past the end re-read the last one and
       // are masked at the LDS write); a pixel-shuffled gy walks +1, +2W-1, +1, +4HW-2W-1 like the conv kernel
      const int co0 = ob * 64 + wave * 16;
      const int cc = co0 < a.Cout ? co0 : a.Cout - 1;
      const char* p;
      if (a.gy_ps)
        p = reinterpret_cast<const char*>(a.gy + (((size_t)n * (a.Cout >> 2) + (cc >> 2)) * (2 * a.Ho) + 2 * oy0 + ((cc >> 1) & 1)) *
                                                     (size_t)(2 * a.Wo) + 2 * ox0 + (cc & 1));
      else
        p = reinterpret_cast<const char*>(a.gy + ((size_t)n * a.Cout + cc) * HWo + (size_t)oy0 * a.Wo + ox0);
      const size_t inc0 = a.gy_ps ? 4 : HWo * 4;
      const size_t inc1 = a.gy_ps ? ((size_t)2 * a.Wo - 1) * 4 : HWo * 4;
      const size_t inc3 = a.gy_ps ? ((size_t)4 * HWo - 2 * a.Wo - 1) * 4 : HWo * 4;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        rg[j] = *reinterpret_cast<const float*>(p + g_off);
        const size_t inc = (j & 1) == 0 ? inc0 : ((j & 3) == 1 ? inc1 : inc3);   // (co0 is a multiple of 16)
        p += (co0 + j + 1 < a.Cout) ? inc : 0;
      }
    }
    const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
    const float* xn = a.x + (size_t)(n / a.x_bdiv) * a.x_bs;
    {  // main pairs: columns 2p, 2p+1 of row xrow; the pair is loaded as two dwords when it straddles the image edge
      const int gy_ = iy0 + xrow, gx_ = ix0 + 2 * xp;
      const bool rok = (unsigned)gy_ < (unsigned)a.H;
      x_ok0 = rok && (unsigned)gx_ < (unsigned)a.W;
      x_ok1 = rok && (unsigned)(gx_ + 1) < (unsigned)a.W;
      const unsigned o0 = x_ok0 ? (unsigned)(gy_ * a.W + gx_) * 4u : 0u, o1 = x_ok1 ? (unsigned)(gy_ * a.W + gx_ + 1) * 4u : 0u;
      const int ci0 = cbk * 64 + wave * 16;
      const char* base = reinterpret_cast<const char*>(xn + (size_t)(ci0 < a.Cin ? ci0 : a.Cin - 1) * HW);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        rx[j].x = *reinterpret_cast<const float*>(base + o0);
        rx[j].y = *reinterpret_cast<const float*>(base + o1);
        base += (ci0 + j + 1 < a.Cin) ? HW * 4 : 0;
      }
    }
    {  // halo pairs (columns 32, 33) of the wave's 16 channels x 4 rows
      const int gy_ = iy0 + trow, gx_ = ix0 + 32;
      int ci = cbk * 64 + wave * 16 + tch;
      ci = ci < a.Cin ? ci : a.Cin - 1;
      const bool rok = (unsigned)gy_ < (unsigned)a.H;
      t_ok0 = rok && (unsigned)gx_ < (unsigned)a.W;
      t_ok1 = rok && (unsigned)(gx_ + 1) < (unsigned)a.W;
      const float* base = xn + (size_t)ci * HW;
      rt.x = base[t_ok0 ? (size_t)gy_ * a.W + gx_ : 0];
      rt.y = base[t_ok1 ? (size_t)gy_ * a.W + gx_ + 1 : 0];
    }
  };
  auto write_lds = [&]() {
    const float t0 = t_ok0 ? rt.x : 0.f, t1 = t_ok1 ? rt.y : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int o = wave * 16 + j;
      s_g[o * WB_GROW + lane] = (__bf16)((g_ok && ob * 64 + o < a.Cout) ? rg[j] : 0.f);
      const bool cok = cbk * 64 + o < a.Cin;
      const float v0 = (x_ok0 && cok) ? rx[j].x : 0.f, v1 = (x_ok1 && cok) ? rx[j].y : 0.f;
      // pair of lane + 1, or (p = 15) the halo pair of (channel j, this row), which lane j*4 + row holds
      const float h0 = __shfl(cok ? t0 : 0.f, j * 4 + xrow, 64), h1 = __shfl(cok ? t1 : 0.f, j * 4 + xrow, 64);
      float n0 = dpp_next(v0), n1 = dpp_next(v1);
      if (xp == 15) { n0 = h0; n1 = h1; }
      unsigned* row = reinterpret_cast<unsigned*>(s_x + o * WB_XCH + xrow * 32) + xp;
      row[0] = pack_bf16(v0, v1);
      row[WB_XCOPY / 2] = pack_bf16(v1, n0);
      row[WB_XCOPY] = pack_bf16(n0, n1);
    }
  };

  const WgSpan sp = wg_span(a, split);
  int tile = sp.tile0;
  if (tile < sp.tile_end) issue_loads(tile);
  for (; tile < sp.tile_end; tile += a.nsplit) {
    write_lds();
    if (tile + a.nsplit < sp.tile_end) issue_loads(tile + a.nsplit);  // in flight under the MFMAs below
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int py = kb >> 1, px0 = (kb & 1) * 16 + 8 * hi;
      const wbf16x8 A = *reinterpret_cast<const wbf16x8*>(s_g + (ot * 32 + lo) * WB_GROW + 16 * kb + 8 * hi);
      if (ct == 0) {  // bias gradient from the operand this wave reads anyway (bf16-rounded values, fp32 sum)
#pragma unroll
        for (int i = 0; i < 8; ++i) db += (float)A[i];
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ty = t / 3, tx = t - ty * 3;
        const wbf16x8 B = *reinterpret_cast<const wbf16x8*>(s_x + tx * WB_XCOPY + (ct * 32 + lo) * WB_XCH + (py + ty) * 32 + px0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- partial[slot][tap][o][c], as conv2d_wgrad_pipe_kernel
  const int OP = a.nob * 64, CP = a.ncb * 64;
  const int slot = sp.slot;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * 64 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int c = cbk * 64 + ct * 32 + lo;
      unsafeAtomicAdd(a.partial + (((size_t)slot * 9 + t) * OP + o) * CP + c, acc[t][r]);
    }
  // lane (lo, hi) summed gy[o = ot*32 + lo] over the pixel blocks of half hi
  if (cbk == 0 && ct == 0) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + ob * 64 + ot * 32 + lo, db);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same weight gradient at fp32 ACCURACY on the bf16 pipe (r04): both operands are split exactly into three bf16 pieces
// (x = hi + mid + lo, 8 + 8 + 8 significand bits) as they are staged, and the six partial products above 2^-24
// (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) are accumulated in fp32 -- the arithmetic of the forward's split kernels
// (conv2d_v2.hip BF = 2, conv2d_wino3.hip).  Per (tap, 16-pixel block) six v_mfma_f32_32x32x16_bf16 (192 cycles) take the
// place of eight v_mfma_f32_32x32x2_f32 (512 cycles); 216 MFMAs per tile and wave instead of 288 at half the cycles each.
// Why: the fp32 kernel (conv2d_wgrad_pipe_kernel) is the largest item of the batched inner MAML step (22 % of its kernel
// time, matrix pipe 57 % busy) and its pipe has no faster exact mode.
//   * LDS: ONE copy of the x tile per piece (4 rows x 34 columns per channel, rows padded to 40, channels to 168 bf16: 16-byte
//     operand reads conflict-free across the 32 channels of a lane group); the column shift of a tap is made in registers
//     (tx = 1: four v_alignbit over the 16-byte operand and the next pixel pair; tx = 2: the dwords one further) instead of by
//     three shifted copies -- 3 x (9.2 + 21.5) KB = 90 KB;
//   * per 16-pixel block and kernel row: 3 + 3 LDS reads per piece set, 18 MFMAs on three accumulators (the three taps of
//     the row) so that no MFMA waits for its predecessor;
//   * the bias gradient is the fp32 sum of the three pieces of gy, i.e. of the exact values.
constexpr int S3_GROW = 72;                  // bf16 per gy row: 64 pixels + 8 pad (144 B)
constexpr int S3_GP = 64 * S3_GROW;          // one piece of the gy tile
constexpr int S3_XROW = 40;                  // bf16 per x row: 34 columns + 6 pad (80 B)
constexpr int S3_XCH = 4 * S3_XROW + 8;      // 168 bf16 (336 B) per channel
constexpr int S3_XP = 64 * S3_XCH;           // one piece of the x tile
constexpr size_t S3_LDS_BYTES = (size_t)(3 * S3_GP + 3 * S3_XP) * 2;

// exact 3-way split of a pair: words {bf16(a) | bf16(b) << 16} of the hi / mid / lo pieces
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
  const float r0 = a - __builtin_bit_cast(float, h << 16), r1 = b - __builtin_bit_cast(float, h & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
  const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l) : "v"(q0), "v"(q1));
}

typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));

// VX = 0: scalar staging (any geometry: pixel-shuffled gy, odd widths).  VX = 4 / 2: the x window as float4 / float2 vectors
// and the gy rows as float4 -- the window starts SHIFT = (VX - pad % VX) % VX columns left of the first column the taps need,
// so that every vector is aligned and lies entirely inside or outside the image (as conv2d_wgrad_wide_item does for the fp32
// kernel): 14 / 22 vector loads and 42 / 66 8-byte (4-byte) LDS stores per lane and tile instead of 50 + 99.
// KS = 3: the 3x3 stride-1 layers; KS = 2: the estimators' 4x4 stride-2 convolutions, re-expressed as 2x2 stride-1 over the
// space-to-depth input (engine.hip: conv4s2).
template <int KS, int VX, int SHIFT>
__global__ __launch_bounds__(256, 1) void conv2d_wgrad_split3_kernel(WgradK a) {
  constexpr int NT = KS * KS, XR = 1 + KS, XC = 31 + KS;   // taps; rows / columns of x a 2 x 32-pixel tile needs
  extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
  __bf16* const s_g = smem16;
  __bf16* const s_x = smem16 + 3 * S3_GP;
  constexpr int WWIN = VX ? ((XC + 2 * (VX - 1)) / VX) * VX : XC;   // window columns per row (3x3: 40 / 36 / 34)
  constexpr int RV = VX ? WWIN / VX : 1, XV = XR * RV;               // vectors per row / per channel
  constexpr int XM = VX ? (16 * XV + 63) / 64 : 1;                   // x vectors per lane and tile (16 channels per wave)
  typedef float xvec __attribute__((ext_vector_type(VX ? VX : 1)));

  const int split = blockIdx.x, ob = blockIdx.y, cbk = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int ot = wave >> 1, ct = wave & 1;
  const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float db = 0.f;

  // ---- scalar staging (VX = 0)
  const int gpy = lane >> 5, gpx = lane & 31;  // gy tile: lane = pixel
  const unsigned g_lane = a.gy_ps ? (unsigned)((2 * gpy) * (2 * a.Wo) + 2 * gpx) : (unsigned)(gpy * a.Wo + gpx);
  const int xrow = lane >> 4, xp = lane & 15;       // x tile: lane = (row, column pair)
  const int trow = lane & 3, tch = lane >> 2;       // halo pair (columns 32, 33): lane = (channel of the wave, row)
  float rg[VX ? 1 : 16];
  float2 rx[VX ? 1 : 16], rt;
  bool g_ok, x_ok0, x_ok1, t_ok0, t_ok1;
  // ---- vector staging (VX = 2, 4): lane-fixed (channel, row, column) of its vectors
  int gch[4], grow[4], gcol[4], xpk[XM], xlds[XM];   // xpk = channel | row << 4 | column << 8 of x vector m
  f32x4 vg[4];
  xvec vx[XM];
  bool vg_ok[4], vx_ok[XM];
  if (VX) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int idx = lane + 64 * m;
      gch[m] = idx >> 4; grow[m] = (idx >> 3) & 1; gcol[m] = (idx & 7) * 4;
    }
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      const int idx = lane + 64 * m;
      const int c = idx / XV, r = idx - c * XV;
      const int xr = r / RV, xc = (r - xr * RV) * VX;
      xpk[m] = (c < 16 ? c : 15) | (xr << 4) | (xc << 8);
      xlds[m] = c < 16 ? (wave * 16 + c) * S3_XCH + xr * S3_XROW + xc : -1;
    }
  }

  auto issue_loads = [&](int tile) {
    const int tx_ = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_ = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int oy0 = ty_ * 2, ox0 = tx_ * 32;
    if constexpr (VX != 0) {
      // (an invalid vector reads the first one of the image instead: channels past Cout / Cin of a partly filled block
      // would otherwise address memory behind the tensor)
      const float* gn = a.gy + (size_t)n * a.Cout * HWo;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int co = ob * 64 + wave * 16 + gch[m];
        vg_ok[m] = co < a.Cout && oy0 + grow[m] < a.Ho && ox0 + gcol[m] < a.Wo;
        const size_t off = vg_ok[m] ? (size_t)co * HWo + (size_t)(oy0 + grow[m]) * a.Wo + ox0 + gcol[m] : 0;
        vg[m] = *reinterpret_cast<const f32x4*>(gn + off);
      }
      const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad - SHIFT;
      const float* xn = a.x + (size_t)(n / a.x_bdiv) * a.x_bs;
#pragma unroll
      for (int m = 0; m < XM; ++m) {
        const int ci = cbk * 64 + wave * 16 + (xpk[m] & 15);
        const int gy_ = iy0 + ((xpk[m] >> 4) & 15), gx_ = ix0 + (xpk[m] >> 8);
        vx_ok[m] = xlds[m] >= 0 && ci < a.Cin && (unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W;
        const size_t off = vx_ok[m] ? (size_t)ci * HW + (size_t)gy_ * a.W + gx_ : 0;
        vx[m] = *reinterpret_cast<const xvec*>(xn + off);
      }
    } else {
    g_ok = oy0 + gpy < a.Ho && ox0 + gpx < a.Wo;
    const unsigned g_off = g_ok ? g_lane * 4u : 0u;
    {
      const int co0 = ob * 64 + wave * 16;
      const int cc = co0 < a.Cout ? co0 : a.Cout - 1;
      const char* p;
      if (a.gy_ps)
        p = reinterpret_cast<const char*>(a.gy + (((size_t)n * (a.Cout >> 2) + (cc >> 2)) * (2 * a.Ho) + 2 * oy0 + ((cc >> 1) & 1)) *
                                                     (size_t)(2 * a.Wo) + 2 * ox0 + (cc & 1));
      else
        p = reinterpret_cast<const char*>(a.gy + ((size_t)n * a.Cout + cc) * HWo + (size_t)oy0 * a.Wo + ox0);
      const size_t inc0 = a.gy_ps ? 4 : HWo * 4;
      const size_t inc1 = a.gy_ps ? ((size_t)2 * a.Wo - 1) * 4 : HWo * 4;
      const size_t inc3 = a.gy_ps ? ((size_t)4 * HWo - 2 * a.Wo - 1) * 4 : HWo * 4;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        rg[j] = *reinterpret_cast<const float*>(p + g_off);
        const size_t inc = (j & 1) == 0 ? inc0 : ((j & 3) == 1 ? inc1 : inc3);   // (co0 is a multiple of 16)
        p += (co0 + j + 1 < a.Cout) ? inc : 0;
      }
    }
    const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
    const float* xn = a.x + (size_t)(n / a.x_bdiv) * a.x_bs;
    {
      const int gy_ = iy0 + xrow, gx_ = ix0 + 2 * xp;
      const bool rok = (unsigned)gy_ < (unsigned)a.H && xrow < XR;
      x_ok0 = rok && (unsigned)gx_ < (unsigned)a.W;
      x_ok1 = rok && (unsigned)(gx_ + 1) < (unsigned)a.W;
      const unsigned o0 = x_ok0 ? (unsigned)(gy_ * a.W + gx_) * 4u : 0u, o1 = x_ok1 ? (unsigned)(gy_ * a.W + gx_ + 1) * 4u : 0u;
      const int ci0 = cbk * 64 + wave * 16;
      const char* base = reinterpret_cast<const char*>(xn + (size_t)(ci0 < a.Cin ? ci0 : a.Cin - 1) * HW);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        rx[j].x = *reinterpret_cast<const float*>(base + o0);
        rx[j].y = *reinterpret_cast<const float*>(base + o1);
        base += (ci0 + j + 1 < a.Cin) ? HW * 4 : 0;
      }
    }
    {
      const int gy_ = iy0 + trow, gx_ = ix0 + 32;
      int ci = cbk * 64 + wave * 16 + tch;
      ci = ci < a.Cin ? ci : a.Cin - 1;
      const bool rok = (unsigned)gy_ < (unsigned)a.H && trow < XR;
      t_ok0 = rok && (unsigned)gx_ < (unsigned)a.W;
      t_ok1 = rok && (unsigned)(gx_ + 1) < (unsigned)a.W;
      const float* base = xn + (size_t)ci * HW;
      rt.x = base[t_ok0 ? (size_t)gy_ * a.W + gx_ : 0];
      rt.y = base[t_ok1 ? (size_t)gy_ * a.W + gx_ + 1 : 0];
    }
    }
  };
  auto write_lds = [&]() {
    if constexpr (VX == 0) {
    // gy: channels in pairs (one v_cvt_pk per piece and pair), 2-byte stores at [piece][o][pixel]
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
      const int o = wave * 16 + j;
      const float v0 = (g_ok && ob * 64 + o < a.Cout) ? rg[j] : 0.f, v1 = (g_ok && ob * 64 + o + 1 < a.Cout) ? rg[j + 1] : 0.f;
      unsigned w[3];
      split3_pair(v0, v1, w[0], w[1], w[2]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const wbf16x2 pr = __builtin_bit_cast(wbf16x2, w[q]);
        __bf16* d = s_g + q * S3_GP + o * S3_GROW + lane;
        d[0] = pr[0];
        d[S3_GROW] = pr[1];
      }
    }
    // x: the column pair of (row, p) as one word per piece; the halo pair (columns 32, 33) of (channel tch, row trow)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = wave * 16 + j;
      const bool cok = cbk * 64 + c < a.Cin;
      const float v0 = (x_ok0 && cok) ? rx[j].x : 0.f, v1 = (x_ok1 && cok) ? rx[j].y : 0.f;
      unsigned w[3];
      split3_pair(v0, v1, w[0], w[1], w[2]);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        reinterpret_cast<wbf16x2*>(s_x + q * S3_XP + c * S3_XCH + xrow * S3_XROW)[xp] = __builtin_bit_cast(wbf16x2, w[q]);
    }
    {
      const int c = wave * 16 + tch;
      const bool cok = cbk * 64 + c < a.Cin;
      unsigned w[3];
      split3_pair((t_ok0 && cok) ? rt.x : 0.f, (t_ok1 && cok) ? rt.y : 0.f, w[0], w[1], w[2]);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        reinterpret_cast<wbf16x2*>(s_x + q * S3_XP + c * S3_XCH + trow * S3_XROW)[16] = __builtin_bit_cast(wbf16x2, w[q]);
    }
    }
  };

  // ---- vector staging: the conversion of the NEXT tile (its loads were issued before this tile's MFMAs) is cut into pieces
  // that sit between the MFMA steps; only the LDS stores of the finished words are left for the gap between two tiles
  unsigned wg0[4][3], wg1[4][3], wx0[XM][3], wx1[XM][3];
  float dbl[4] = {0.f, 0.f, 0.f, 0.f};   // bias gradient: fp32 sums of the staged gy vectors (cbk == 0 workgroups)
  auto convert_g = [&](int m) {
    const f32x4 v = vg_ok[m] ? vg[m] : f32x4{0.f, 0.f, 0.f, 0.f};
    dbl[m] += (v[0] + v[1]) + (v[2] + v[3]);
    split3_pair(v[0], v[1], wg0[m][0], wg0[m][1], wg0[m][2]);
    split3_pair(v[2], v[3], wg1[m][0], wg1[m][1], wg1[m][2]);
  };
  auto convert_x = [&](int m) {
    xvec v = vx[m];
    if (!vx_ok[m]) {
#pragma unroll
      for (int e = 0; e < (VX ? VX : 1); ++e) v[e] = 0.f;
    }
    split3_pair(v[0], v[VX ? 1 : 0], wx0[m][0], wx0[m][1], wx0[m][2]);
    if (VX == 4) split3_pair(v[VX == 4 ? 2 : 0], v[VX == 4 ? 3 : 0], wx1[m][0], wx1[m][1], wx1[m][2]);
  };
  auto store_words = [&]() {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        *reinterpret_cast<wbf16x4*>(s_g + q * S3_GP + (wave * 16 + gch[m]) * S3_GROW + grow[m] * 32 + gcol[m]) =
            __builtin_bit_cast(wbf16x4, uint2{wg0[m][q], wg1[m][q]});
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      if (xlds[m] < 0) continue;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (VX == 4) *reinterpret_cast<wbf16x4*>(s_x + q * S3_XP + xlds[m]) = __builtin_bit_cast(wbf16x4, uint2{wx0[m][q], wx1[m][q]});
        else *reinterpret_cast<wbf16x2*>(s_x + q * S3_XP + xlds[m]) = __builtin_bit_cast(wbf16x2, wx0[m][q]);
      }
    }
  };

  // ---- MFMA loop of one tile: 4 KS steps (16-pixel block kb, kernel row ty) of 6 KS MFMAs; the operands of step i + 1 are read
  // from LDS before the MFMAs of step i (two register sets), `fill(i)` is the piece of other work placed behind step i
  wbf16x8 A[2][3], R0[2][3], R1[2][3];
  wbf16x2 R4[2][3];
  auto load_step = [&](int st, int rb) {
    const int kb = st / KS, ty = st - kb * KS;
    const int py = kb >> 1, px0 = (kb & 1) * 16 + 8 * hi;
    if (ty == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        A[kb & 1][q] = *reinterpret_cast<const wbf16x8*>(s_g + q * S3_GP + (ot * 32 + lo) * S3_GROW + 16 * kb + 8 * hi);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const __bf16* bp = s_x + q * S3_XP + (ct * 32 + lo) * S3_XCH + (py + ty) * S3_XROW + px0;
      R0[rb][q] = *reinterpret_cast<const wbf16x8*>(bp);
      if (VX != 0) R1[rb][q] = *reinterpret_cast<const wbf16x8*>(bp + 8);
      else R4[rb][q] = *reinterpret_cast<const wbf16x2*>(bp + 8);
    }
  };
  auto mfma_tile = [&](auto&& fill) __attribute__((always_inline)) {
    load_step(0, 0);
    static_for<0, 4 * KS>([&](auto st_) {
      constexpr int st = decltype(st_)::value;
      constexpr int kb = st / KS, ty = st - kb * KS, rb = st & 1;
      if (st + 1 < 4 * KS) load_step(st + 1, rb ^ 1);
      if (VX == 0 && ct == 0 && ty == 0) {  // (scalar staging: the bias gradient from the A operands, whose pieces sum to the exact values)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < 8; ++i) db += (float)A[kb & 1][q][i];
      }
      wbf16x8 B[3][3];   // [piece][tx]
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const wbf16x8 r0 = R0[rb][q];
        if constexpr (VX != 0) {
          const wbf16x8 r1 = R1[rb][q];
          // tap tx reads the window columns px0 + SHIFT + tx .. + 7 (element shuffles: v_alignbit / v_perm / moves)
          B[q][0] = __builtin_shufflevector(r0, r1, SHIFT, SHIFT + 1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7);
          B[q][1] = __builtin_shufflevector(r0, r1, SHIFT + 1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7, SHIFT + 8);
          B[q][2] = __builtin_shufflevector(r0, r1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7, SHIFT + 8, SHIFT + 9);
        } else {
          const wbf16x2 r4 = R4[rb][q];
          B[q][0] = r0;
          B[q][1] = wbf16x8{r0[1], r0[2], r0[3], r0[4], r0[5], r0[6], r0[7], r4[0]};
          B[q][2] = wbf16x8{r0[2], r0[3], r0[4], r0[5], r0[6], r0[7], r4[0], r4[1]};
        }
      }
      // hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the three taps of the row alternate (independent accumulators)
      constexpr int QA[6] = {0, 0, 1, 0, 2, 1}, QB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
          acc[ty * KS + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kb & 1][QA[pr]], B[QB[pr]][tx], acc[ty * KS + tx], 0, 0, 0);
      fill(st_);
    });
  };

  const WgSpan sp = wg_span(a, split);
  int tile = sp.tile0;
  if constexpr (VX != 0) {
    if (tile < sp.tile_end) {
      issue_loads(tile);
#pragma unroll
      for (int m = 0; m < 4; ++m) convert_g(m);
#pragma unroll
      for (int m = 0; m < XM; ++m) convert_x(m);
    }
    for (; tile < sp.tile_end; tile += a.nsplit) {
      store_words();
      const bool has_next = tile + a.nsplit < sp.tile_end;
      if (has_next) issue_loads(tile + a.nsplit);  // in flight under the first MFMA steps below
      __syncthreads();
      mfma_tile([&](auto st_) __attribute__((always_inline)) {
        constexpr int st = decltype(st_)::value;
        // the last eight steps: the next tile's vectors have landed long ago -- convert them (registers only)
        if (has_next && st >= 4 * KS - 8) {
          constexpr int j = st - (4 * KS - 8);   // 0 .. 7
          if (j < 4) convert_g(j);
          constexpr int per = (XM + 7) / 8;
#pragma unroll
          for (int u = 0; u < per; ++u)
            if (j * per + u < XM) convert_x(j * per + u);
        }
      });
      __syncthreads();
    }
    if (cbk != 0) { dbl[0] = dbl[1] = dbl[2] = dbl[3] = 0.f; }
  } else {
    if (tile < sp.tile_end) issue_loads(tile);
    for (; tile < sp.tile_end; tile += a.nsplit) {
      write_lds();
      if (tile + a.nsplit < sp.tile_end) issue_loads(tile + a.nsplit);  // in flight under the MFMAs below
      __syncthreads();
      mfma_tile([&](auto) __attribute__((always_inline)) {});
      __syncthreads();
    }
  }

  // ---- partial[slot][tap][o][c], as conv2d_wgrad_pipe_kernel
  const int OP = a.nob * 64, CP = a.ncb * 64;
  const int slot = sp.slot;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * 64 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int c = cbk * 64 + ct * 32 + lo;
      unsafeAtomicAdd(a.partial + (((size_t)slot * NT + t) * OP + o) * CP + c, acc[t][r]);
    }
  if constexpr (VX != 0) {
    if (cbk == 0) {   // lanes 16 k .. 16 k + 15 staged channel (lane >> 4) + 4 m of this wave's 16
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float v = dbl[m];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
        const int co = ob * 64 + wave * 16 + gch[m];
        if ((lane & 15) == 0 && co < OP) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + co, v);
      }
    }
  } else {
    if (cbk == 0 && ct == 0) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + ob * 64 + ot * 32 + lo, db);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the vector-staging forms (VX = 4 / 2) of the kernel above, re-scheduled.  The arithmetic, the LDS images and the
// flush are the same; what changed is WHEN things are issued -- one wave per SIMD has nobody to hide its stalls behind:
//   * the compiler had sunk every operand read of the MFMA loop down to its first use (ds_read_b128 x2, s_waitcnt, v_alignbit
//     ... in the middle of the MFMA stream: ~36 exposed LDS round trips per tile).  The reads of step i + 1 now sit above a
//     scheduling fence at the TOP of step i and are waited for by an lgkmcnt(0) closed by a fence at the top of step i + 1
//     (conv2d_wino4.hip's wait_lds): a whole step of MFMAs (576 cycles) covers them;
//   * the next tile's global loads are issued from INSIDE the loop (behind the first step's MFMAs) instead of between the LDS
//     stores and the barrier, with 32-bit lane offsets off a scalar sample base: validity is two compares and a select per
//     vector (before: 64-bit multiply-adds under exec masks, ~500 instructions with the matrix pipe idle);
//   * a workgroup's last tile loads / converts its own tile again instead of branching around the staging code (the loop body
//     is one straight line);
//   * v_cvt_pk_bf16_f32 as a vector conversion, not inline asm (no s_nop padding behind it).
__device__ __forceinline__ void split3_pair_v(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  typedef float sf2 __attribute__((ext_vector_type(2)));
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(sf2{a, b}, wbf16x2));
  const float r0 = a - __builtin_bit_cast(float, h << 16), r1 = b - __builtin_bit_cast(float, h & 0xffff0000u);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(sf2{r0, r1}, wbf16x2));
  const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(sf2{q0, q1}, wbf16x2));
}

#ifdef DVSR_CONV_TRACE
#define S3_STAMP(i)                                                                                                          \
  do {                                                                                                                       \
    if (a.trace && threadIdx.x == 0)                                                                                         \
      a.trace[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define S3_STAMP(i) \
  do {              \
  } while (0)
#endif

// KYS: one kernel ROW per workgroup (block index = split * KS + ky): KS accumulator tiles and two x rows instead of KS * KS
// and KS + 1 -- under 256 registers and 62 KB of LDS, so that TWO workgroups share a CU.  One wave per SIMD issues its MFMAs
// and everything else one after the other (tools/wgrad_trace.py: 47 cycles per MFMA with the shuffles between them against
// 37.5 alone; LDS stores, barriers and the first operand reads of a tile with the pipe idle); a second wave on the SIMD fills
// those gaps (tools/mfma_overlap.hip: MFMAs of one wave and vector / LDS instructions of the other run at full rate).  The
// price is staging: gy three times, x one and a half times.
template <int KS, int VX, int SHIFT, bool KYS>
__global__ __launch_bounds__(256, KYS ? 2 : 1) void conv2d_wgrad_split3v_kernel(WgradK a) {
  static_assert(VX == 2 || VX == 4, "vector staging only");
  constexpr int KR = KYS ? 1 : KS, NT = KR * KS, XR = 1 + KR, XC = 31 + KS;   // kernel rows / taps / x rows of a workgroup
  constexpr int XCH0 = XR * S3_XROW + 8, XCH = (XCH0 / 8) % 2 ? XCH0 : XCH0 + 8, XP = 64 * XCH;                        // bf16 per channel / per piece of the x tile
  extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
  __bf16* const s_g = smem16;
  __bf16* const s_x = smem16 + 3 * S3_GP;
  static_assert((XCH * 2 / 16) % 2 == 1, "channel stride: an odd number of 16-byte units");
  constexpr int WWIN = ((XC + 2 * (VX - 1)) / VX) * VX;
  constexpr int RV = WWIN / VX, XV = XR * RV;
  constexpr int XM = (16 * XV + 63) / 64;
  typedef float xvec __attribute__((ext_vector_type(VX)));

  const int split = KYS ? blockIdx.x / KS : blockIdx.x, ky = KYS ? blockIdx.x % KS : 0, ob = blockIdx.y, cbk = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int ot = wave >> 1, ct = wave & 1;
  const unsigned HW = (unsigned)a.H * a.W, HWo = (unsigned)a.Ho * a.Wo;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- lane-fixed part of the staging map: byte offsets inside a sample, (row, column) inside the tile / window
  unsigned g_rel[4], x_rel[XM];
  int g_row[4], g_col[4], x_row[XM], x_col[XM], g_lds[4], x_lds[XM];
  bool g_cok[4], x_cok[XM];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int idx = lane + 64 * m;
    const int ch = idx >> 4;
    g_row[m] = (idx >> 3) & 1; g_col[m] = (idx & 7) * 4;
    const int co = ob * 64 + wave * 16 + ch;
    g_cok[m] = co < a.Cout;
    g_rel[m] = ((unsigned)(g_cok[m] ? co : 0) * HWo + (unsigned)g_row[m] * a.Wo + g_col[m]) * 4u;
    g_lds[m] = (wave * 16 + ch) * S3_GROW + g_row[m] * 32 + g_col[m];
  }
#pragma unroll
  for (int m = 0; m < XM; ++m) {
    const int idx = lane + 64 * m;
    const int c = idx / XV, r = idx - c * XV;
    x_row[m] = r / RV; x_col[m] = (r - x_row[m] * RV) * VX;
    const int ci = cbk * 64 + wave * 16 + c;
    x_cok[m] = c < 16 && ci < a.Cin;
    x_rel[m] = ((unsigned)(x_cok[m] ? ci : 0) * HW + (unsigned)x_row[m] * a.W + x_col[m]) * 4u;
    x_lds[m] = c < 16 ? (wave * 16 + c) * XCH + x_row[m] * S3_XROW + x_col[m] : -1;
  }

  f32x4 vg[4];
  xvec vx[XM];
  bool vg_ok[4], vx_ok[XM];
  auto issue_loads = [&](int tile) __attribute__((always_inline)) {
    const int tx_ = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_ = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int oy0 = ty_ * 2, ox0 = tx_ * 32;
    const char* gn = reinterpret_cast<const char*>(a.gy + (size_t)n * a.Cout * HWo);
    const unsigned g_tile = ((unsigned)oy0 * a.Wo + ox0) * 4u;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      vg_ok[m] = g_cok[m] && oy0 + g_row[m] < a.Ho && ox0 + g_col[m] < a.Wo;
      vg[m] = *reinterpret_cast<const f32x4*>(gn + (vg_ok[m] ? g_rel[m] + g_tile : 0u));
    }
    const int iy0 = oy0 - a.pad + ky, ix0 = ox0 - a.pad - SHIFT;
    const char* xn = reinterpret_cast<const char*>(a.x + (size_t)(n / a.x_bdiv) * a.x_bs);
    const unsigned x_tile = (unsigned)(iy0 * a.W + ix0) * 4u;   // (may wrap: the sum with a valid lane's offset does not)
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      vx_ok[m] = x_cok[m] && (unsigned)(iy0 + x_row[m]) < (unsigned)a.H && (unsigned)(ix0 + x_col[m]) < (unsigned)a.W;
      vx[m] = *reinterpret_cast<const xvec*>(xn + (vx_ok[m] ? x_rel[m] + x_tile : 0u));
    }
  };

  unsigned wg0[4][3], wg1[4][3], wx0[XM][3], wx1[XM][3];
  float dbl[4] = {0.f, 0.f, 0.f, 0.f};
  float db_on = 1.f;   // 0 while a workgroup's last tile is staged a second time (see above)
  auto convert_g = [&](int m) __attribute__((always_inline)) {
    const f32x4 v = vg_ok[m] ? vg[m] : f32x4{0.f, 0.f, 0.f, 0.f};
    dbl[m] = __builtin_fmaf(db_on, (v[0] + v[1]) + (v[2] + v[3]), dbl[m]);
    split3_pair_v(v[0], v[1], wg0[m][0], wg0[m][1], wg0[m][2]);
    split3_pair_v(v[2], v[3], wg1[m][0], wg1[m][1], wg1[m][2]);
  };
  auto convert_x = [&](int m) __attribute__((always_inline)) {
    xvec v = vx[m];
    if (!vx_ok[m]) {
#pragma unroll
      for (int e = 0; e < VX; ++e) v[e] = 0.f;
    }
    split3_pair_v(v[0], v[1], wx0[m][0], wx0[m][1], wx0[m][2]);
    if (VX == 4) split3_pair_v(v[VX == 4 ? 2 : 0], v[VX == 4 ? 3 : 0], wx1[m][0], wx1[m][1], wx1[m][2]);
  };
  auto store_words = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        *reinterpret_cast<wbf16x4*>(s_g + q * S3_GP + g_lds[m]) = __builtin_bit_cast(wbf16x4, uint2{wg0[m][q], wg1[m][q]});
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      if (x_lds[m] < 0) continue;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (VX == 4) *reinterpret_cast<wbf16x4*>(s_x + q * XP + x_lds[m]) = __builtin_bit_cast(wbf16x4, uint2{wx0[m][q], wx1[m][q]});
        else *reinterpret_cast<wbf16x2*>(s_x + q * XP + x_lds[m]) = __builtin_bit_cast(wbf16x2, wx0[m][q]);
      }
    }
  };

  wbf16x8 A[2][3], R0[2][3], R1[2][3];
  const __bf16* const a_base = s_g + (ot * 32 + lo) * S3_GROW + 8 * hi;
  const __bf16* const b_base = s_x + (ct * 32 + lo) * XCH + 8 * hi;
  auto load_step = [&](auto st_, int rb) __attribute__((always_inline)) {
    constexpr int st = decltype(st_)::value;
    constexpr int kb = st / KR, ty = st - kb * KR;
    constexpr int py = kb >> 1, px0 = (kb & 1) * 16;
    // A of pixel block kb goes into A[kb & 1]: with KS steps per block its reads ride with the block's first step (two steps
    // ahead, while block kb - 1 -- the other set -- is in use); with ONE step per block (KYS) two steps ahead would overwrite
    // the set in use, so they are issued one step ahead instead (load_a below)
    if (KR > 1 && ty == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) A[kb & 1][q] = *reinterpret_cast<const wbf16x8*>(a_base + q * S3_GP + 16 * kb);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const __bf16* bp = b_base + q * XP + (py + ty) * S3_XROW + px0;
      R0[rb][q] = *reinterpret_cast<const wbf16x8*>(bp);
      R1[rb][q] = *reinterpret_cast<const wbf16x8*>(bp + 8);
    }
  };

  auto load_a = [&](auto kb_) __attribute__((always_inline)) {   // (KR == 1 only)
    constexpr int kb = decltype(kb_)::value;
#pragma unroll
    for (int q = 0; q < 3; ++q) A[kb & 1][q] = *reinterpret_cast<const wbf16x8*>(a_base + q * S3_GP + 16 * kb);
  };

  const WgSpan sp = wg_span(a, split);
  int tile = sp.tile0;
  S3_STAMP(0);
  if (tile < sp.tile_end) {
    issue_loads(tile);
#pragma unroll
    for (int m = 0; m < 4; ++m) convert_g(m);
#pragma unroll
    for (int m = 0; m < XM; ++m) convert_x(m);
  }
  S3_STAMP(1);
  [[maybe_unused]] int it = 0;
  for (; tile < sp.tile_end; tile += a.nsplit) {
    if (it == 2) S3_STAMP(4);
    store_words();
    const bool has_next = tile + a.nsplit < sp.tile_end;
    const int tnext = has_next ? tile + a.nsplit : tile;
    db_on = has_next ? 1.f : 0.f;
    if (it == 2) S3_STAMP(5);
    __syncthreads();
    if (it == 2) S3_STAMP(6);
    // two steps of operand reads in flight, the B fragments of step i + 1 shuffled together between the MFMAs of step i
    wbf16x8 B[2][3][3];   // [step parity][piece][tx]
    auto build_b = [&](int rb) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const wbf16x8 r0 = R0[rb][q], r1 = R1[rb][q];
        B[rb][q][0] = __builtin_shufflevector(r0, r1, SHIFT, SHIFT + 1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7);
        B[rb][q][1] = __builtin_shufflevector(r0, r1, SHIFT + 1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7, SHIFT + 8);
        B[rb][q][2] = __builtin_shufflevector(r0, r1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7, SHIFT + 8, SHIFT + 9);
      }
    };
    if constexpr (KR == 1) load_a(std::integral_constant<int, 0>{});
    load_step(std::integral_constant<int, 0>{}, 0);
    load_step(std::integral_constant<int, 1>{}, 1);
    build_b(0);
    static_for<0, 4 * KR>([&](auto st_) __attribute__((always_inline)) {
      constexpr int st = decltype(st_)::value;
      constexpr int kb = st / KR, ty = st - kb * KR, rb = st & 1;
      // the reads of step st + 2 go out HERE (R[rb] was consumed by build_b in the previous step): nothing of this step may be
      // scheduled above them and they cannot sink to their first use; the compiler's own lgkmcnt(n) covers the older reads
      if constexpr (st + 2 < 4 * KR) load_step(std::integral_constant<int, st + 2>{}, rb);
      if constexpr (KR == 1 && st + 1 < 4) load_a(std::integral_constant<int, st + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (st == 0) issue_loads(tnext);
      constexpr int QA[6] = {0, 0, 1, 0, 2, 1}, QB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
          acc[ty * KS + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kb & 1][QA[pr]], B[rb][QB[pr]][tx], acc[ty * KS + tx], 0, 0, 0);
      if constexpr (st + 1 < 4 * KR) build_b(rb ^ 1);
      // the next tile's vectors landed steps ago: convert them (registers only) behind the last CS steps
      constexpr int CS = 4 * KR >= 8 ? 8 : 4 * KR - 1;
      if constexpr (st >= 4 * KR - CS) {
        constexpr int j = st - (4 * KR - CS);
        constexpr int perg = (4 + CS - 1) / CS, perx = (XM + CS - 1) / CS;
#pragma unroll
        for (int u = 0; u < perg; ++u)
          if (j * perg + u < 4) convert_g(j * perg + u);
#pragma unroll
        for (int u = 0; u < perx; ++u)
          if (j * perx + u < XM) convert_x(j * perx + u);
      }
      // one MFMA (32 cycles of the pipe), then up to four of the vector instructions above
#ifndef S3_MG
#define S3_MG 1
#define S3_IL 4
#endif
#if S3_MG > 0
#pragma unroll
      for (int i = 0; i < 6 * KS / S3_MG; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, S3_MG, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, S3_IL, 0);
      }
#endif
#ifdef DVSR_CONV_TRACE
      __builtin_amdgcn_sched_barrier(0);
      if (it == 2) S3_STAMP(10 + st);
      __builtin_amdgcn_sched_barrier(0);
#endif
    });
    if (it == 2) S3_STAMP(7);
    __syncthreads();
    if (it == 2) S3_STAMP(8);
    ++it;
  }
  S3_STAMP(2);

  // ---- partial[slot][tap][o][c], as conv2d_wgrad_pipe_kernel
  const int OP = a.nob * 64, CP = a.ncb * 64;
  const int slot = sp.slot;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * 64 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int c = cbk * 64 + ct * 32 + lo;
      unsafeAtomicAdd(a.partial + (((size_t)slot * (KS * KS) + ky * KS + t) * OP + o) * CP + c, acc[t][r]);
    }
  S3_STAMP(3);
  if (cbk == 0 && ky == 0) {   // lanes 16 k .. 16 k + 15 staged channel (lane >> 4) + 4 m of this wave's 16
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float v = dbl[m];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
      const int co = ob * 64 + wave * 16 + (lane >> 4) + 4 * m;
      if ((lane & 15) == 0 && co < OP) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + co, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5 (late): the same tile on EIGHT waves -- two per SIMD WITHOUT staging anything twice (the row split above pays for
// its second wave with gy three times and x one and a half times).  Wave (ot = wave & 1, ct = wave >> 1) owns 32 output
// channels x 16 input channels on v_mfma_f32_16x16x32_bf16: K = 32 = one pixel row of the tile, two A blocks (16 couts each)
// share a B block, 9 taps x 2 x 4 = 72 accumulator registers.  Six steps per tile (pixel row py, kernel row ty) of 36 MFMAs;
// per step six 16-byte operand reads for B (+ six for A when py changes) and the same 36 shuffle instructions as the
// four-wave form -- half the vector work per wave, and what is left runs beside the partner wave's MFMAs.  A wave stages 8
// channels of either operand.  Same LDS images, same arithmetic, same flush.
template <int KS, int VX, int SHIFT>
__global__ __launch_bounds__(512, 2) void conv2d_wgrad_split3w_kernel(WgradK a) {
  static_assert(VX == 2 || VX == 4, "vector staging only");
  constexpr int NT = KS * KS, XR = 1 + KS, XC = 31 + KS;
  constexpr int XCH0 = XR * S3_XROW + 8, XCH = (XCH0 / 8) % 2 ? XCH0 : XCH0 + 8, XP = 64 * XCH;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
  __bf16* const s_g = smem16;
  __bf16* const s_x = smem16 + 3 * S3_GP;
  constexpr int WWIN = ((XC + 2 * (VX - 1)) / VX) * VX;
  constexpr int RV = WWIN / VX, XV = XR * RV;
  constexpr int XM = (8 * XV + 63) / 64;
  typedef float xvec __attribute__((ext_vector_type(VX)));
  typedef float f32x4w __attribute__((ext_vector_type(4)));

  const int split = blockIdx.x, ob = blockIdx.y, cbk = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kg = lane >> 4;
  const int ot = wave & 1, ct = wave >> 1;
  const unsigned HW = (unsigned)a.H * a.W, HWo = (unsigned)a.Ho * a.Wo;

  f32x4w acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[t][b] = f32x4w{0.f, 0.f, 0.f, 0.f};

  // ---- lane-fixed staging map (8 channels of gy and of x per wave)
  unsigned g_rel[2], x_rel[XM];
  int g_row[2], g_col[2], x_row[XM], x_col[XM], g_lds[2], x_lds[XM];
  bool g_cok[2], x_cok[XM];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int idx = lane + 64 * m;
    const int ch = idx >> 4;
    g_row[m] = (idx >> 3) & 1; g_col[m] = (idx & 7) * 4;
    const int co = ob * 64 + wave * 8 + ch;
    g_cok[m] = co < a.Cout;
    g_rel[m] = ((unsigned)(g_cok[m] ? co : 0) * HWo + (unsigned)g_row[m] * a.Wo + g_col[m]) * 4u;
    g_lds[m] = (wave * 8 + ch) * S3_GROW + g_row[m] * 32 + g_col[m];
  }
#pragma unroll
  for (int m = 0; m < XM; ++m) {
    const int idx = lane + 64 * m;
    const int c = idx / XV, r = idx - c * XV;
    x_row[m] = r / RV; x_col[m] = (r - x_row[m] * RV) * VX;
    const int ci = cbk * 64 + wave * 8 + c;
    x_cok[m] = c < 8 && ci < a.Cin;
    x_rel[m] = ((unsigned)(x_cok[m] ? ci : 0) * HW + (unsigned)x_row[m] * a.W + x_col[m]) * 4u;
    x_lds[m] = c < 8 ? (wave * 8 + c) * XCH + x_row[m] * S3_XROW + x_col[m] : -1;
  }

  f32x4 vg[2];
  xvec vx[XM];
  bool vg_ok[2], vx_ok[XM];
  auto issue_loads = [&](int tile) __attribute__((always_inline)) {
    const int tx_ = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_ = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int oy0 = ty_ * 2, ox0 = tx_ * 32;
    const char* gn = reinterpret_cast<const char*>(a.gy + (size_t)n * a.Cout * HWo);
    const unsigned g_tile = ((unsigned)oy0 * a.Wo + ox0) * 4u;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      vg_ok[m] = g_cok[m] && oy0 + g_row[m] < a.Ho && ox0 + g_col[m] < a.Wo;
      vg[m] = *reinterpret_cast<const f32x4*>(gn + (vg_ok[m] ? g_rel[m] + g_tile : 0u));
    }
    const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad - SHIFT;
    const char* xn = reinterpret_cast<const char*>(a.x + (size_t)(n / a.x_bdiv) * a.x_bs);
    const unsigned x_tile = (unsigned)(iy0 * a.W + ix0) * 4u;
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      vx_ok[m] = x_cok[m] && (unsigned)(iy0 + x_row[m]) < (unsigned)a.H && (unsigned)(ix0 + x_col[m]) < (unsigned)a.W;
      vx[m] = *reinterpret_cast<const xvec*>(xn + (vx_ok[m] ? x_rel[m] + x_tile : 0u));
    }
  };

  unsigned wg0[2][3], wg1[2][3], wx0[XM][3], wx1[XM][3];
  float dbl[2] = {0.f, 0.f};
  float db_on = 1.f;
  auto convert_g = [&](int m) __attribute__((always_inline)) {
    const f32x4 v = vg_ok[m] ? vg[m] : f32x4{0.f, 0.f, 0.f, 0.f};
    dbl[m] = __builtin_fmaf(db_on, (v[0] + v[1]) + (v[2] + v[3]), dbl[m]);
    split3_pair_v(v[0], v[1], wg0[m][0], wg0[m][1], wg0[m][2]);
    split3_pair_v(v[2], v[3], wg1[m][0], wg1[m][1], wg1[m][2]);
  };
  auto convert_x = [&](int m) __attribute__((always_inline)) {
    xvec v = vx[m];
    if (!vx_ok[m]) {
#pragma unroll
      for (int e = 0; e < VX; ++e) v[e] = 0.f;
    }
    split3_pair_v(v[0], v[1], wx0[m][0], wx0[m][1], wx0[m][2]);
    if (VX == 4) split3_pair_v(v[VX == 4 ? 2 : 0], v[VX == 4 ? 3 : 0], wx1[m][0], wx1[m][1], wx1[m][2]);
  };
  auto store_words = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        *reinterpret_cast<wbf16x4*>(s_g + q * S3_GP + g_lds[m]) = __builtin_bit_cast(wbf16x4, uint2{wg0[m][q], wg1[m][q]});
#pragma unroll
    for (int m = 0; m < XM; ++m) {
      if (x_lds[m] < 0) continue;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (VX == 4) *reinterpret_cast<wbf16x4*>(s_x + q * XP + x_lds[m]) = __builtin_bit_cast(wbf16x4, uint2{wx0[m][q], wx1[m][q]});
        else *reinterpret_cast<wbf16x2*>(s_x + q * XP + x_lds[m]) = __builtin_bit_cast(wbf16x2, wx0[m][q]);
      }
    }
  };

  // operands: A[block][piece] = gy[ot*32 + block*16 + l16][row py, pixels 8 kg ..]; R0 / R1 [piece] = x[ct*16 + l16][row py + ty,
  // window columns 8 kg .. 8 kg + 15]
  wbf16x8 A[2][3], R0[2][3], R1[2][3];
  const __bf16* const a_base = s_g + (ot * 32 + l16) * S3_GROW + 8 * kg;
  const __bf16* const b_base = s_x + (ct * 16 + l16) * XCH + 8 * kg;
  auto load_a = [&](int py) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 3; ++q) A[b][q] = *reinterpret_cast<const wbf16x8*>(a_base + q * S3_GP + b * 16 * S3_GROW + py * 32);
  };
  auto load_r = [&](int row, int rb) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const __bf16* bp = b_base + q * XP + row * S3_XROW;
      R0[rb][q] = *reinterpret_cast<const wbf16x8*>(bp);
      R1[rb][q] = *reinterpret_cast<const wbf16x8*>(bp + 8);
    }
  };

  const WgSpan sp = wg_span(a, split);
  int tile = sp.tile0;
  if (tile < sp.tile_end) {
    issue_loads(tile);
#pragma unroll
    for (int m = 0; m < 2; ++m) convert_g(m);
#pragma unroll
    for (int m = 0; m < XM; ++m) convert_x(m);
  }
  for (; tile < sp.tile_end; tile += a.nsplit) {
    store_words();
    const bool has_next = tile + a.nsplit < sp.tile_end;
    const int tnext = has_next ? tile + a.nsplit : tile;
    db_on = has_next ? 1.f : 0.f;
    __syncthreads();
    load_a(0);
    load_r(0, 0);
    static_for<0, 2 * KS>([&](auto st_) __attribute__((always_inline)) {
      constexpr int st = decltype(st_)::value;
      constexpr int py = st / KS, ty = st - py * KS, rb = st & 1;
      // this step's B fragments out of R[rb] (read one step ago); then the next step's reads go out, pinned above the MFMAs
      wbf16x8 B[3][3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const wbf16x8 r0 = R0[rb][q], r1 = R1[rb][q];
        B[q][0] = __builtin_shufflevector(r0, r1, SHIFT, SHIFT + 1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7);
        B[q][1] = __builtin_shufflevector(r0, r1, SHIFT + 1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7, SHIFT + 8);
        B[q][2] = __builtin_shufflevector(r0, r1, SHIFT + 2, SHIFT + 3, SHIFT + 4, SHIFT + 5, SHIFT + 6, SHIFT + 7, SHIFT + 8, SHIFT + 9);
      }
      if constexpr (st + 1 < 2 * KS) load_r((st + 1) / KS + (st + 1) % KS, rb ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (st == 0) issue_loads(tnext);
      constexpr int QA[6] = {0, 0, 1, 0, 2, 1}, QB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
      for (int pr = 0; pr < 6; ++pr)
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[ty * KS + tx][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[b][QA[pr]], B[QB[pr]][tx], acc[ty * KS + tx][b], 0, 0, 0);
      // (A of the second pixel row: after the last MFMAs that read the first row's)
      if constexpr (st == KS - 1) load_a(1);
      // the next tile's vectors: converted behind the last steps (registers only)
      constexpr int CS = 2 * KS - 2;
      if constexpr (st >= 2 * KS - CS) {
        constexpr int j = st - (2 * KS - CS);
        constexpr int perx = (XM + CS - 1) / CS;
        if (j < 2) convert_g(j);
#pragma unroll
        for (int u = 0; u < perx; ++u)
          if (j * perx + u < XM) convert_x(j * perx + u);
      }
    });
    __syncthreads();
  }

  // ---- partial[slot][tap][o][c]: D rows = 4 kg + r (couts), column = l16 (input channel)
  const int OP = a.nob * 64, CP = a.ncb * 64;
  const int slot = sp.slot;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = ob * 64 + ot * 32 + b * 16 + 4 * kg + r;
        const int c = cbk * 64 + ct * 16 + l16;
        unsafeAtomicAdd(a.partial + (((size_t)slot * NT + t) * OP + o) * CP + c, acc[t][b][r]);
      }
  if (cbk == 0) {   // lanes 16 k .. 16 k + 15 staged channel (lane >> 4) + 4 m of this wave's 8
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v = dbl[m];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
      const int co = ob * 64 + wave * 8 + (lane >> 4) + 4 * m;
      if ((lane & 15) == 0 && co < OP) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + co, v);
    }
  }
}

template <int KS, int VX, int SHIFT>
static int launch_split3(const WgradLaunch& l, hipStream_t st) {
  if constexpr (VX != 0) {
    // DVSR_WGRAD_S3V=0: the round-4 schedule of the same kernel (A/B switch, read once per process); samples whose byte
    // offsets do not fit 32 bits keep it too.  l.kys (conv2d_wgrad_prepare): one kernel row per workgroup, two workgroups per CU
    static const bool s3v = [] { const char* v = getenv("DVSR_WGRAD_S3V"); return !(v && v[0] == '0'); }();
    const bool fits = (unsigned long long)l.k.Cin * l.k.H * l.k.W < (1ull << 30) && (unsigned long long)l.k.Cout * l.k.Ho * l.k.Wo < (1ull << 30);
    if (l.kys) {
      DVSR_REQUIRE(s3v && fits, DVSR_ERR_UNSUPPORTED, "conv2d_wgrad_split3: the row split needs the vector-staging kernel");
      constexpr size_t lds = (size_t)(3 * S3_GP + 3 * 64 * (2 * S3_XROW + 8)) * 2;
      auto kv = conv2d_wgrad_split3v_kernel<KS, VX, SHIFT, true>;
      static PerDeviceOnce attr_once_k;
      set_dyn_lds_once(attr_once_k, (const void*)kv, lds);
      hipLaunchKernelGGL(kv, l.grid, dim3(256), lds, st, l.k);
      return check_launch("conv2d_wgrad_split3v_kernel<kys>");
    }
    // the eight-wave form (two waves per SIMD, 16x16x32 MFMAs) for the float4-staged launches that are not row-split: 7 - 14 %
    // per launch over the four-wave form (profiles/r05_wgrad_s3w.txt); the float2 forms spill there and stay on four waves.
    // DVSR_WGRAD_S3W=0: A/B switch, read once per process
    static const bool s3w = [] { const char* v = getenv("DVSR_WGRAD_S3W"); return !(v && v[0] == '0'); }();
    if (s3v && fits && s3w && VX == 4) {
      auto kw = conv2d_wgrad_split3w_kernel<KS, VX, SHIFT>;
      static PerDeviceOnce attr_once_w;
      set_dyn_lds_once(attr_once_w, (const void*)kw, S3_LDS_BYTES);
      hipLaunchKernelGGL(kw, l.grid, dim3(512), S3_LDS_BYTES, st, l.k);
      return check_launch("conv2d_wgrad_split3w_kernel");
    }
    if (s3v && fits) {
      auto kv = conv2d_wgrad_split3v_kernel<KS, VX, SHIFT, false>;
      static PerDeviceOnce attr_once_v;
      set_dyn_lds_once(attr_once_v, (const void*)kv, S3_LDS_BYTES);
      hipLaunchKernelGGL(kv, l.grid, dim3(256), S3_LDS_BYTES, st, l.k);
      return check_launch("conv2d_wgrad_split3v_kernel");
    }
  }
  DVSR_REQUIRE(!l.kys, DVSR_ERR_UNSUPPORTED, "conv2d_wgrad_split3: the row split needs the vector-staging kernel");
  auto kern = conv2d_wgrad_split3_kernel<KS, VX, SHIFT>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, S3_LDS_BYTES);
  hipLaunchKernelGGL(kern, l.grid, dim3(256), S3_LDS_BYTES, st, l.k);
  return check_launch("conv2d_wgrad_split3_kernel");
}

int conv2d_wgrad_split3_launch(const WgradLaunch& l, hipStream_t st) {
  // l.k.vx (conv2d_wgrad_prepare): 4 / 2 when gy rows are float4-loadable and the x rows float4 / float2-loadable
  const int pad = l.k.pad;
  if (l.ks == 2) {   // (pad 0 or 1: the forward of the space-to-depth form / nothing else; other pads take the scalar staging)
    if (l.k.vx == 4 && pad == 0) return launch_split3<2, 4, 0>(l, st);
    if (l.k.vx == 2 && pad == 0) return launch_split3<2, 2, 0>(l, st);
    return launch_split3<2, 0, 0>(l, st);
  }
  if (l.k.vx == 4 && pad == 1) return launch_split3<3, 4, 3>(l, st);
  if (l.k.vx == 4 && pad == 0) return launch_split3<3, 4, 0>(l, st);
  if (l.k.vx == 2 && pad == 1) return launch_split3<3, 2, 1>(l, st);
  if (l.k.vx == 2 && pad == 0) return launch_split3<3, 2, 0>(l, st);
  return launch_split3<3, 0, 0>(l, st);
}

int conv2d_wgrad_bf16_launch(const WgradLaunch& l, hipStream_t st) {
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)conv2d_wgrad_bf16_kernel, WB_LDS_BYTES);
  hipLaunchKernelGGL(conv2d_wgrad_bf16_kernel, l.grid, dim3(256), WB_LDS_BYTES, st, l.k);
  return check_launch("conv2d_wgrad_bf16_kernel");
}

}  // namespace dvsr
