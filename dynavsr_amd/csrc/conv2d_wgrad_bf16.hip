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
// loads per lane, issued for tile i+1 before the MFMAs of tile i), converts and writes them (3 ds_write_b16 per x
// element).  Accumulation and the flush stay fp32; the bias gradient is summed in fp32 from the loaded values.
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

__global__ __launch_bounds__(256, 1) void conv2d_wgrad_bf16_kernel(WgradK a) {
  constexpr int IW = 34, PLANE = 4 * IW, XM = 3;
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
  int xiy[XM], xix[XM];
#pragma unroll
  for (int m = 0; m < XM; ++m) {
    const int e = lane + 64 * m;
    xiy[m] = e / IW;
    xix[m] = e - xiy[m] * IW;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbacc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) dbacc[j] = 0.f;

  float rg[16], rx[16][XM];
  bool g_ok, x_ok[XM];
  auto issue_loads = [&](int tile) {
    const int tx_ = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_ = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int oy0 = ty_ * 2, ox0 = tx_ * 32;
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
    const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
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
  auto write_lds = [&]() {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int o = wave * 16 + j;
      const float gv = (g_ok && ob * 64 + o < a.Cout) ? rg[j] : 0.f;
      dbacc[j] += gv;
      s_g[o * WB_GROW + lane] = (__bf16)gv;
      const bool cok = cbk * 64 + o < a.Cin;
#pragma unroll
      for (int m = 0; m < XM; ++m) {
        if (lane + 64 * m < PLANE) {
          const __bf16 v = (__bf16)((x_ok[m] && cok) ? rx[j][m] : 0.f);
          __bf16* row = s_x + o * WB_XCH + xiy[m] * 32;
          const int ix = xix[m];
          if (ix < 32) row[ix] = v;                                   // copy 0: columns 0..31
          if (ix >= 1 && ix < 33) row[WB_XCOPY + ix - 1] = v;         // copy 1: shifted by one
          if (ix >= 2) row[2 * WB_XCOPY + ix - 2] = v;                // copy 2: shifted by two
        }
      }
    }
  };

  int tile = split;
  if (tile < a.ntiles) issue_loads(tile);
  for (; tile < a.ntiles; tile += a.nsplit) {
    write_lds();
    if (tile + a.nsplit < a.ntiles) issue_loads(tile + a.nsplit);  // in flight under the MFMAs below
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int py = kb >> 1, px0 = (kb & 1) * 16 + 8 * hi;
      const wbf16x8 A = *reinterpret_cast<const wbf16x8*>(s_g + (ot * 32 + lo) * WB_GROW + 16 * kb + 8 * hi);
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
  const int slot = split % a.nslot;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * 64 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int c = cbk * 64 + ct * 32 + lo;
      unsafeAtomicAdd(a.partial + (((size_t)slot * 9 + t) * OP + o) * CP + c, acc[t][r]);
    }
  if (cbk == 0) {  // bias gradient: channel wave*16 + j, summed over the 64 pixel lanes
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v = dbacc[j];
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
      if (lane == 0) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + ob * 64 + wave * 16 + j, v);
    }
  }
}

int conv2d_wgrad_bf16_launch(const WgradLaunch& l, hipStream_t st) {
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute((const void*)conv2d_wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WB_LDS_BYTES);
    done = true;
  }
  hipLaunchKernelGGL(conv2d_wgrad_bf16_kernel, l.grid, dim3(256), WB_LDS_BYTES, st, l.k);
  return check_launch("conv2d_wgrad_bf16_kernel");
}

}  // namespace dvsr
