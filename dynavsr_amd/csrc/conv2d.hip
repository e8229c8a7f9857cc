// Dense 1x1 / 3x3 convolution as an LDS-staged implicit GEMM on v_mfma_f32_32x32x2_f32.
//
// Replaces every nn.Conv2d of EDVR_arch.py / arch_util.py together with the torch.cat in front
// of it (two-pointer K loop) and the bias / activation / residual / PixelShuffle(2) behind it.
//
// Tiling (wave = 64 lanes): a workgroup of 4 waves owns an 8x32-pixel output tile x 64 output
// channels.  GEMM view per workgroup: D[cout 64][pixel 256] += Wt[cout][k] * X[k][pixel] with
// k = (cin, tap).  MFMA operand roles are chosen so that D rows = cout and D columns = pixels:
// a half-wave then stores 32 consecutive pixels of one channel = one 128-byte line.
//   A (lane l): W[cout = l&31][cin = 2kk + (l>>5)]  <- s_w[(c*KK + tap)*65 + cout]   (conflict-free)
//   B (lane l): X[cin = 2kk + (l>>5)][px = l&31]    <- s_in[c][row*S + ty][px*S + tx] (conflict-free)
// Each wave: rows {2w, 2w+1} of the tile x 2 cout halves = 4 accumulators of 16 VGPRs.
// K loop: chunks of CC input channels; each chunk stages CC x (8S+ks-S) x (32S+ks-S) inputs
// (zero padded) and 64 x CC x ks^2 weights (transposed from OIHW on the fly) into LDS.
#include "common.h"
#include "kernels.h"

namespace dvsr {

struct ConvK {
  const float* x0; const float* x1; const float* w; const float* bias; const float* res; float* y;
  int N, c0, c1, H, W, Cout, Ho, Wo, pad, act, ps, x1_bdiv;
  long long x0_bs, x1_bs;
  int tiles_x, tiles_y, ntiles, ncb;
  int wt;        // 0: w is [Cout][Ctot][KK]; 1: transposed+flipped view (dgrad), see below
  int w_ctot, w_coff;  // wt=1: element (ci, co, t) = w[(ci*w_ctot + w_coff + co)*KK + t]
  int in_ps;     // 1: x0 is stored pixel-shuffled [N][c0/4][2H][2W] (gradient of a PixelShuffle(2) output)
  int in_dil;    // 2: x0 is a zero-dilated view of a [N][c0][Hs][Ws] tensor (dgrad of a stride-2 conv)
  int Hs, Ws;    // source dims for in_dil
  int accum;     // 1: y += result instead of y = result
};

template <int KS, int S, int CC>
struct ConvShape {
  static constexpr int TH = 8, TW = 32, KK = KS * KS;
  static constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
  static constexpr int PLANE = IH * IW;
  static constexpr int WROW = 65;
  static constexpr int IN_FLOATS = CC * PLANE;
  static constexpr int W_FLOATS = CC * KK * WROW;
  static constexpr size_t LDS_BYTES = (size_t)(IN_FLOATS + W_FLOATS) * sizeof(float);
};

// MT = 32-cout halves computed per workgroup: 2, or 1 when the layer has <= 32 output channels (SpyNet's 7x7 convs
// to 32 / 16 / 2 channels and their data gradients: the second half would multiply zero weights).
template <int KS, int S, int CC, int MT = 2>
__global__ __launch_bounds__(256, 2) void conv2d_mfma_kernel(ConvK a) {
  using Sh = ConvShape<KS, S, CC>;
  constexpr int KK = Sh::KK, IH = Sh::IH, IW = Sh::IW, PLANE = Sh::PLANE, WROW = Sh::WROW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;
  float* s_w = smem + Sh::IN_FLOATS;

  // XCD-aware block order: consecutive ids go to different XCDs (id % 8), so the ncb cout
  // blocks of one pixel tile are given ids that differ by 8 -> same XCD, shared L2 input tile.
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

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int cbase = 0; cbase < Ctot; cbase += CC) {
    __syncthreads();
    // ---- stage the input halo tile (zero padding outside the image / beyond Ctot)
    for (int idx = tid; idx < Sh::IN_FLOATS; idx += 256) {
      const int c = idx / PLANE;
      const int r = idx - c * PLANE;
      const int iy = r / IW, ix = r - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix, ci = cbase + c;
      float v = 0.f;
      if (ci < Ctot && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
        if (a.in_ps) {
          // channel ci = 4*cq + 2*dy + dx of the un-shuffled view lives at [cq][2y+dy][2x+dx]
          const float* src = x0n + (size_t)(ci >> 2) * (4 * HW);
          v = src[(size_t)(2 * gy + ((ci >> 1) & 1)) * (2 * a.W) + 2 * gx + (ci & 1)];
        } else if (a.in_dil) {
          if (!((gy | gx) & 1) && (gy >> 1) < a.Hs && (gx >> 1) < a.Ws)
            v = x0n[(size_t)ci * a.Hs * a.Ws + (size_t)(gy >> 1) * a.Ws + (gx >> 1)];
        } else {
          const float* src = ci < a.c0 ? x0n + (size_t)ci * HW : x1n + (size_t)(ci - a.c0) * HW;
          v = src[(size_t)gy * a.W + gx];
        }
      }
      s_in[idx] = v;
    }
    // ---- stage weights: s_w[(c*KK + tap)*65 + o]
    for (int idx = tid; idx < 64 * CC * KK; idx += 256) {
      int o, c, tap;
      float v = 0.f;
      if (a.wt == 0) {
        o = idx / (CC * KK);
        const int rem = idx - o * (CC * KK);
        c = rem / KK;
        tap = rem - c * KK;
        const int co = cb * 64 + o, ci = cbase + c;
        if (co < a.Cout && ci < Ctot) v = a.w[((size_t)co * Ctot + ci) * KK + tap];
      } else {
        // dgrad view: this conv's input channel ci is the original conv's output channel,
        // this conv's output channel co the original's input channel, taps mirrored.
        c = idx / (64 * KK);
        const int rem = idx - c * (64 * KK);
        o = rem / KK;
        const int t = rem - o * KK;
        tap = KK - 1 - t;
        const int co = cb * 64 + o, ci = cbase + c;
        if (co < a.Cout && ci < Ctot) v = a.w[((size_t)ci * a.w_ctot + a.w_coff + co) * KK + t];
      }
      s_w[(c * KK + tap) * WROW + o] = v;
    }
    __syncthreads();
    // ---- MFMA over (tap, channel pair)
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      const int ty = tap / KS, tx = tap - ty * KS;
      const float* pin = s_in + ((2 * wave) * S + ty) * IW + lo * S + tx;
#pragma unroll
      for (int kk = 0; kk < CC / 2; ++kk) {
        const int c = 2 * kk + hi;
        const float a0 = s_w[(c * KK + tap) * WROW + lo];
        const float b0 = pin[c * PLANE];
        const float b1 = pin[c * PLANE + S * IW];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        if constexpr (MT == 2) {
          const float a1 = s_w[(c * KK + tap) * WROW + 32 + lo];
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: bias, activation, residual, (pixel-shuffled) store
  const TileOut t{a.y, a.bias, a.res, a.act, a.ps, a.accum, a.Cout, a.Ho, a.Wo};
  store_mfma_tile<MT, 2>(acc, t, n, cb * 64, oy0, 8, ox0, oy0 + 2 * wave, lo, hi);
}

template <int KS, int S, int CC, int MT = 2>
static int launch_conv(const ConvK& k, hipStream_t st) {
  using Sh = ConvShape<KS, S, CC>;
  auto kern = conv2d_mfma_kernel<KS, S, CC, MT>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, Sh::LDS_BYTES);
  const int grid = ceil_div(k.ntiles, 8) * 8 * k.ncb;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Sh::LDS_BYTES, st, k);
  return check_launch("conv2d_mfma_kernel");
}

int conv2d_run(const dvsr_conv2d_desc& d, const ConvExtra& ex, hipStream_t st) {
  const int transposed_w = ex.wt;
  DVSR_REQUIRE(d.x0 && d.w && d.y, DVSR_ERR_INVALID, "conv2d: null x0/w/y");
  DVSR_REQUIRE(d.N > 0 && d.c0 > 0 && d.c1 >= 0 && d.H > 0 && d.W > 0 && d.Cout > 0,
               DVSR_ERR_INVALID, "conv2d: non-positive dimension");
  DVSR_REQUIRE(d.c1 == 0 || d.x1, DVSR_ERR_INVALID, "conv2d: c1 > 0 but x1 is null");
  DVSR_REQUIRE(d.ks == 1 || d.ks == 3 || d.ks == 7 || d.ks == 9, DVSR_ERR_UNSUPPORTED,
               "conv2d: ks=%d (supported: 1, 3, and the 7 / 9 of TOFlow)", d.ks);
  DVSR_REQUIRE(d.stride == 1 || (d.stride == 2 && d.ks == 3), DVSR_ERR_UNSUPPORTED,
               "conv2d: stride=%d with ks=%d unsupported", d.stride, d.ks);
  DVSR_REQUIRE(d.pad == d.ks / 2, DVSR_ERR_UNSUPPORTED, "conv2d: pad=%d must be ks/2", d.pad);
  DVSR_REQUIRE(d.act >= 0 && d.act <= 2, DVSR_ERR_INVALID, "conv2d: act=%d", d.act);
  DVSR_REQUIRE(d.pixel_shuffle == 0 || (d.pixel_shuffle == 2 && d.Cout % 4 == 0 && !d.res),
               DVSR_ERR_INVALID, "conv2d: pixel_shuffle needs Cout%%4==0 and no residual");
  ConvK k;
  k.x0 = d.x0; k.x1 = d.x1; k.w = d.w; k.bias = d.bias; k.res = d.res; k.y = d.y;
  k.N = d.N; k.c0 = d.c0; k.c1 = d.c1; k.H = d.H; k.W = d.W; k.Cout = d.Cout;
  k.pad = d.pad; k.act = d.act; k.ps = d.pixel_shuffle; k.x1_bdiv = d.x1_bdiv > 0 ? d.x1_bdiv : 1;
  k.x0_bs = d.x0_bstride > 0 ? d.x0_bstride : (long long)d.c0 * d.H * d.W;
  k.x1_bs = d.x1_bstride > 0 ? d.x1_bstride : (long long)d.c1 * d.H * d.W;
  k.Ho = (d.H + 2 * d.pad - d.ks) / d.stride + 1;
  k.Wo = (d.W + 2 * d.pad - d.ks) / d.stride + 1;
  k.tiles_x = ceil_div(k.Wo, 32);
  k.tiles_y = ceil_div(k.Ho, 8);
  k.ntiles = k.tiles_x * k.tiles_y * d.N;
  k.ncb = ceil_div(d.Cout, 64);
  k.wt = transposed_w;
  k.w_ctot = ex.w_ctot; k.w_coff = ex.w_coff; k.in_ps = ex.in_ps; k.in_dil = ex.in_dil;
  k.Hs = ex.Hs; k.Ws = ex.Ws; k.accum = ex.accum;
  DVSR_REQUIRE(!(ex.in_ps || ex.in_dil) || d.c1 == 0, DVSR_ERR_INVALID,
               "conv2d: in_ps/in_dil views take a single input");
  if (k.in_ps) k.x0_bs = (long long)d.c0 * d.H * d.W;
  if (k.in_dil) k.x0_bs = (long long)d.c0 * ex.Hs * ex.Ws;
  if (d.ks == 3 && d.stride == 1) return launch_conv<3, 1, 8>(k, st);
  if (d.ks == 3 && d.stride == 2) return launch_conv<3, 2, 8>(k, st);
  // SpyNet's 7x7 and TOFlow's 9x9 (TOF_arch.py:32-42, 107-108): chunks sized so that the 64 x CC x ks^2 weight
  // image fits LDS next to the halo tile (119 KB / 94 KB: one workgroup per CU)
  if (d.ks == 7) return d.Cout <= 32 ? launch_conv<7, 1, 8, 1>(k, st) : launch_conv<7, 1, 8, 2>(k, st);
  if (d.ks == 9) return d.Cout <= 32 ? launch_conv<9, 1, 4, 1>(k, st) : launch_conv<9, 1, 4, 2>(k, st);
  return launch_conv<1, 1, 32>(k, st);
}

}  // namespace dvsr

extern "C" int dvsr_conv2d_forward(const dvsr_conv2d_desc* d, dvsr_stream_t stream) {
  DVSR_REQUIRE(d, DVSR_ERR_INVALID, "conv2d: null descriptor");
  return dvsr::conv2d_run(*d, dvsr::ConvExtra(), (hipStream_t)stream);
}
