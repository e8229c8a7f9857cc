// 3x3 convolution with very few output channels (EDVR's conv_last: 64 -> 3 on the 720x1280 HR grid).
//
// On the MFMA tile this layer computes a 32-wide cout block to keep 3 of it (9.9 TFLOP/s useful,
// 322 us in the r01 profile).  It is really an HBM-bound streaming op (read 64 planes once, write 3):
// one thread = one output pixel x all COUT channels on the VALU, the 8-channel halo tile staged in LDS
// exactly like conv2d.hip, weights read with wave-uniform (scalar) loads.  Bias, activation and the
// residual add (the bilinear base frame, EDVR_arch.py:311-312) are fused.
#include "common.h"
#include "kernels.h"

namespace dvsr {

struct SmallK {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int N, C, H, W, Cout, act, tiles_x, tiles_y, ntiles;
  int wdiv; long long w_gs; int b_gs;   // per-sample weight sets (common.h: wset_ptr)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One thread = TWO horizontally adjacent output pixels (one packed fp32 pair: v_pk_fma_f32 does 2 FMAs per
// lane per instruction) x COUT channels, tile = 8 rows x 64 columns, 8 input channels per chunk.  A window
// row for both pixels is one ds_read_b128.  The weights (w[o][ci][tap] at ci*WPC + tap*COUT + o) are staged
// in LDS once and read as broadcast float4s: scalar loads share lgkmcnt with the LDS reads and cost a full
// SMEM round trip several times per channel (125 -> 105 us).  Measured alternatives that were slower:
// 16-row tiles (130 us), four pixels per thread with 4-channel chunks (133 us).
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_cout_kernel(SmallK a) {
  constexpr int CC = 8, TH = 8, TW = 64, IH = TH + 2, IW = TW + 2, PLANE = IH * IW, E = (PLANE + 255) / 256;
  constexpr int WPC = ((9 * COUT + 3) / 4) * 4;
  __shared__ __attribute__((aligned(16))) float s_in[2][CC * PLANE];
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [ceil(C/8)*8][WPC]
  // an XCD (blockIdx.x & 7: workgroups are dealt round-robin) owns a band of tile rows, so the two halo rows a
  // tile shares with its vertical neighbours are L2 hits
  const int tpx = (a.ntiles + 7) >> 3;
  const int tile = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= tpx || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * TW;
  const int tid = threadIdx.x;
  const int py = tid >> 5, px2 = (tid & 31) * 2;
  const size_t HW = (size_t)a.H * a.W;
  const float* xn = a.x + (size_t)n * a.C * HW;
  const float* wn = wset_ptr(a.w, a.w_gs, n, a.wdiv);
  const float* bn = wset_ptr(a.bias, a.b_gs, n, a.wdiv);

  unsigned eoff[E];
  bool evalid[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = tid + 256 * e;
    const int iy = idx / IW, ix = idx - iy * IW;
    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
    evalid[e] = idx < PLANE && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    eoff[e] = evalid[e] ? (unsigned)(gy * a.W + gx) * 4u : 0u;
  }
  f32x2 acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = f32x2{0.f, 0.f};
  float rin[CC][E];
  const int nch = ceil_div(a.C, CC);
  for (int i = tid; i < nch * CC * WPC; i += 256) {
    const int ci = i / WPC, r = i - ci * WPC;
    const int tap = r / COUT, o = r - tap * COUT;
    s_w[i] = (ci < a.C && tap < 9) ? wn[((size_t)o * a.C + ci) * 9 + tap] : 0.f;
  }
  auto prefetch = [&](int k) {
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const int ci = k * CC + c;
      const char* src = reinterpret_cast<const char*>(xn + (size_t)(ci < a.C ? ci : 0) * HW);  // scalar base
#pragma unroll
      for (int e = 0; e < E; ++e) rin[c][e] = *reinterpret_cast<const float*>(src + eoff[e]);
    }
  };
  prefetch(0);
  for (int k = 0; k < nch; ++k) {
    float* s = s_in[k & 1];
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int idx = tid + 256 * e;
        if (e + 1 < E || idx < PLANE) s[c * PLANE + idx] = (evalid[e] && k * CC + c < a.C) ? rin[c][e] : 0.f;
      }
    __syncthreads();  // (also orders the one-time weight staging before its first use)
    if (k + 1 < nch) prefetch(k + 1);
    const float* p0 = s + py * IW + px2;
#pragma unroll
    for (int c = 0; c < CC; ++c) {  // channels >= C hold zeros in both LDS images: no branch needed
      f32x4 wq[WPC / 4];
#pragma unroll
      for (int q = 0; q < WPC / 4; ++q)
        wq[q] = *reinterpret_cast<const f32x4*>(s_w + (size_t)(k * CC + c) * WPC + 4 * q);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const f32x2 lo = *reinterpret_cast<const f32x2*>(p0 + c * PLANE + ty * IW);
        const f32x2 hi = *reinterpret_cast<const f32x2*>(p0 + c * PLANE + ty * IW + 2);
        const float v[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            const int wi = (ty * 3 + tx) * COUT + o;
            const float wv = wq[wi >> 2][wi & 3];
            acc[o] = __builtin_elementwise_fma(f32x2{wv, wv}, f32x2{v[tx], v[tx + 1]}, acc[o]);
          }
      }
    }
  }
  const int oy = oy0 + py, ox = ox0 + px2;
  if (oy >= a.H || ox >= a.W) return;
  const bool two = ox + 1 < a.W;
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    const float bv = bn ? bn[o] : 0.f;
    f32x2 v = {apply_act(acc[o][0] + bv, a.act), apply_act(acc[o][1] + bv, a.act)};
    const size_t idx = ((size_t)n * a.Cout + o) * HW + (size_t)oy * a.W + ox;
    if (two && (idx & 1) == 0) {  // whole, 8-byte aligned pair
      if (a.res) v += *reinterpret_cast<const f32x2*>(a.res + idx);
      *reinterpret_cast<f32x2*>(a.y + idx) = v;
    } else {
      a.y[idx] = v[0] + (a.res ? a.res[idx] : 0.f);
      if (two) a.y[idx + 1] = v[1] + (a.res ? a.res[idx + 1] : 0.f);
    }
  }
}

// x [N][C][H][W] (dense), w [Cout][C][3][3], Cout <= 4, stride 1, pad 1.
int conv3x3_small_cout_run(const float* x, const float* w, const float* bias, const float* res, float* y, int N,
                           int C, int H, int W, int Cout, int act, hipStream_t st, int wdiv, long long w_gs, int b_gs) {
  DVSR_REQUIRE(x && w && y && Cout >= 1 && Cout <= 4, DVSR_ERR_INVALID, "conv3x3_small_cout: bad argument");
  SmallK k;
  k.x = x; k.w = w; k.bias = bias; k.res = res; k.y = y; k.N = N; k.C = C; k.H = H; k.W = W; k.Cout = Cout; k.act = act;
  k.wdiv = wdiv > 0 ? wdiv : 1; k.w_gs = w_gs; k.b_gs = b_gs;
  k.tiles_x = ceil_div(W, 64); k.tiles_y = ceil_div(H, 8); k.ntiles = k.tiles_x * k.tiles_y * N;
  auto wbytes = [&](int cout) { return (size_t)ceil_div(C, 8) * 8 * (((9 * cout + 3) / 4) * 4) * sizeof(float); };
  switch (Cout) {  // exact channel count: no wasted accumulators
    case 1: hipLaunchKernelGGL(conv3x3_small_cout_kernel<1>, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), wbytes(1), st, k); break;
    case 2: hipLaunchKernelGGL(conv3x3_small_cout_kernel<2>, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), wbytes(2), st, k); break;
    case 3: hipLaunchKernelGGL(conv3x3_small_cout_kernel<3>, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), wbytes(3), st, k); break;
    default: hipLaunchKernelGGL(conv3x3_small_cout_kernel<4>, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), wbytes(4), st, k); break;
  }
  return check_launch("conv3x3_small_cout_kernel");
}

}  // namespace dvsr
