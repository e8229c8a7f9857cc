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
};

template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_cout_kernel(SmallK a) {
  constexpr int CC = 8, TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, PLANE = IH * IW, E = (PLANE + 255) / 256;
  __shared__ float s_in[2][CC * PLANE];
  const int tile = blockIdx.x;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * TW;
  const int tid = threadIdx.x;
  const int py = tid >> 5, px = tid & 31;
  const size_t HW = (size_t)a.H * a.W;
  const float* xn = a.x + (size_t)n * a.C * HW;

  int eoff[E];
  bool evalid[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = tid + 256 * e;
    const int iy = idx / IW, ix = idx - iy * IW;
    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
    evalid[e] = idx < PLANE && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    eoff[e] = evalid[e] ? gy * a.W + gx : 0;
  }
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
  float rin[CC][E];
  const int nch = ceil_div(a.C, CC);
  auto prefetch = [&](int k) {
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const int ci = k * CC + c;
      const float* src = xn + (size_t)(ci < a.C ? ci : 0) * HW;
#pragma unroll
      for (int e = 0; e < E; ++e) rin[c][e] = src[eoff[e]];
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
        if (idx < PLANE) s[c * PLANE + idx] = (evalid[e] && k * CC + c < a.C) ? rin[c][e] : 0.f;
      }
    __syncthreads();
    if (k + 1 < nch) prefetch(k + 1);
    const float* p0 = s + py * IW + px;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const int ci = k * CC + c;
      if (ci >= a.C) break;  // wave-uniform
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = p0[c * PLANE + (t / 3) * IW + (t % 3)];
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
          // channel index clamped (never read past the [Cout][C][3][3] tensor); the surplus
          // accumulators are simply not stored.  A branch here costs 3.5x (r01_c profile).
          const int oc = o < a.Cout ? o : a.Cout - 1;
          acc[o] = fmaf(a.w[((size_t)oc * a.C + ci) * 9 + t], v, acc[o]);
        }
      }
    }
  }
  const int oy = oy0 + py, ox = ox0 + px;
  if (oy < a.H && ox < a.W) {
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      if (o >= a.Cout) break;
      float v = acc[o];
      if (a.bias) v += a.bias[o];
      v = apply_act(v, a.act);
      const size_t idx = ((size_t)n * a.Cout + o) * HW + (size_t)oy * a.W + ox;
      if (a.res) v += a.res[idx];
      a.y[idx] = v;
    }
  }
}

// x [N][C][H][W] (dense), w [Cout][C][3][3], Cout <= 4, stride 1, pad 1.
int conv3x3_small_cout_run(const float* x, const float* w, const float* bias, const float* res, float* y, int N,
                           int C, int H, int W, int Cout, int act, hipStream_t st) {
  DVSR_REQUIRE(x && w && y && Cout >= 1 && Cout <= 4, DVSR_ERR_INVALID, "conv3x3_small_cout: bad argument");
  SmallK k;
  k.x = x; k.w = w; k.bias = bias; k.res = res; k.y = y; k.N = N; k.C = C; k.H = H; k.W = W; k.Cout = Cout; k.act = act;
  k.tiles_x = ceil_div(W, 32); k.tiles_y = ceil_div(H, 8); k.ntiles = k.tiles_x * k.tiles_y * N;
  hipLaunchKernelGGL(conv3x3_small_cout_kernel<4>, dim3(k.ntiles), dim3(256), 0, st, k);
  return check_launch("conv3x3_small_cout_kernel");
}

}  // namespace dvsr
