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

int conv2d_wgrad_bf16_launch(const WgradLaunch& l, hipStream_t st) {
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)conv2d_wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WB_LDS_BYTES);
  }
  hipLaunchKernelGGL(conv2d_wgrad_bf16_kernel, l.grid, dim3(256), WB_LDS_BYTES, st, l.k);
  return check_launch("conv2d_wgrad_bf16_kernel");
}

}  // namespace dvsr
