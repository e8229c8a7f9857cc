// Streaming helpers of the MFDN / SFDN estimators (LRimg_estimator.py:38-117): explicit padding as
// the reference does it (nn.ReflectionPad2d(1) :46,78 / nn.ReplicationPad3d(1) :77) plus the layout
// changes that let every convolution of the estimator run on the dense stride-1 MFMA kernels:
//   PAD_REFLECT      out[n][c][y][x]              = in[n][c][refl(y-1)][refl(x-1)]          (H+2 x W+2)
//   PAD_REFLECT_S2D  out[n][4c+2dy+dx][Y][X]      = reflect-padded in at (2Y+dy, 2X+dx): the 4x4 stride-2
//                    convolutions (:83-85) become 2x2 stride-1 convolutions over 4C channels
//   PAD_REPL_T3      out[b*T+t][3c+dt][y][x]      = in[b*T+clamp(t+dt-1)][c][clamp(y-1)][clamp(x-1)]: the
//                    3x3x3 Conv3d over ReplicationPad3d(1) (:76,88) becomes a 3x3 conv over 3C channels
//                    whose weight tensor [Cout][C][3][3][3] is already laid out as [Cout][3C][3][3]
// and their adjoints (gradient folds), the per-frame mean subtraction / re-addition (:93,116) with
// the B,C,T,H,W <-> (B*T),C,H,W transposes folded in, and the 4x4 -> 2x2-over-4C weight re-layout.
// All of these are HBM-bound elementwise kernels (<1 FLOP/B).
#include "common.h"
#include "kernels.h"

namespace dvsr {

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ int clamp0(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// one thread per output element
__global__ void pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int mode,
                               int C, int H, int W, int T) {
  const int Hp = H + 2, Wp = W + 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (mode == PAD_REFLECT) {
      const int q = (int)(i % Wp);
      size_t t = i / Wp;
      const int r = (int)(t % Hp);
      const size_t plane = t / Hp;  // n*C + c
      y[i] = x[(plane * H + reflect1(r - 1, H)) * W + reflect1(q - 1, W)];
    } else if (mode == PAD_REFLECT_S2D) {
      const int Wh = Wp / 2, Hh = Hp / 2;
      const int X = (int)(i % Wh);
      size_t t = i / Wh;
      const int Y = (int)(t % Hh); t /= Hh;
      const int c4 = (int)(t % (4 * C));
      const size_t n = t / (4 * C);
      const int c = c4 >> 2, dy = (c4 >> 1) & 1, dx = c4 & 1;
      y[i] = x[((n * C + c) * H + reflect1(2 * Y + dy - 1, H)) * W + reflect1(2 * X + dx - 1, W)];
    } else {  // PAD_REPL_T3
      const int q = (int)(i % Wp);
      size_t t = i / Wp;
      const int r = (int)(t % Hp); t /= Hp;
      const int c3 = (int)(t % (3 * C));
      const size_t n = t / (3 * C);
      const int c = c3 / 3, dt = c3 - 3 * c;
      const int b = (int)(n / T), tt = (int)(n - (size_t)b * T);
      const size_t src = (size_t)b * T + clamp0(tt + dt - 1, T);
      y[i] = x[((src * C + c) * H + clamp0(r - 1, H)) * W + clamp0(q - 1, W)];
    }
  }
}

// adjoint: one thread per INPUT element, gathering every padded position that maps to it
__global__ void pad_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, size_t total, int mode,
                               int C, int H, int W, int T, int accumulate) {
  const int Hp = H + 2, Wp = W + 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    size_t t = i / W;
    const int yy = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const size_t n = t / C;
    // padded rows / columns that read this element
    int rows[3], cols[3], nr = 0, nc = 0;
    rows[nr++] = yy + 1;
    cols[nc++] = xx + 1;
    if (mode == PAD_REPL_T3) {
      if (yy == 0) rows[nr++] = 0;
      if (yy == H - 1) rows[nr++] = Hp - 1;
      if (xx == 0) cols[nc++] = 0;
      if (xx == W - 1) cols[nc++] = Wp - 1;
    } else {
      if (yy == 1) rows[nr++] = 0;
      if (yy == H - 2) rows[nr++] = Hp - 1;
      if (xx == 1) cols[nc++] = 0;
      if (xx == W - 2) cols[nc++] = Wp - 1;
    }
    float s = 0.f;
    if (mode == PAD_REFLECT) {
      const float* p = gy + (n * C + c) * (size_t)Hp * Wp;
      for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nc; ++b) s += p[(size_t)rows[a] * Wp + cols[b]];
    } else if (mode == PAD_REFLECT_S2D) {
      const int Wh = Wp / 2, Hh = Hp / 2;
      for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nc; ++b) {
          const int r = rows[a], q = cols[b];
          s += gy[((n * 4 * C + c * 4 + (r & 1) * 2 + (q & 1)) * Hh + (r >> 1)) * (size_t)Wh + (q >> 1)];
        }
    } else {
      const int b_ = (int)(n / T), tt = (int)(n - (size_t)b_ * T);
      // (t, dt) pairs with clamp(t + dt - 1) == tt
      int ts[5], ds[5], np = 0;
      for (int dt = 0; dt < 3; ++dt) {
        const int t0 = tt - dt + 1;
        if (t0 >= 0 && t0 < T) { ts[np] = t0; ds[np] = dt; ++np; }
      }
      if (tt == 0) { ts[np] = 0; ds[np] = 0; ++np; }          // t + dt - 1 = -1 clamps to 0
      if (tt == T - 1) { ts[np] = T - 1; ds[np] = 2; ++np; }  // = T clamps to T - 1
      for (int k = 0; k < np; ++k) {
        const float* p = gy + (((size_t)b_ * T + ts[k]) * 3 * C + c * 3 + ds[k]) * (size_t)Hp * Wp;
        for (int a = 0; a < nr; ++a)
          for (int b = 0; b < nc; ++b) s += p[(size_t)rows[a] * Wp + cols[b]];
      }
    }
    gx[i] = accumulate ? gx[i] + s : s;
  }
}

static int grid_for(size_t total) {
  const size_t b = (total + 255) / 256;
  return (int)(b < 65535u * 16u ? (b ? b : 1) : 65535u * 16u);
}

size_t pad_out_numel(int mode, size_t N, int C, int H, int W) {
  if (mode == PAD_REFLECT) return N * C * (size_t)(H + 2) * (W + 2);
  if (mode == PAD_REFLECT_S2D) return N * 4 * C * (size_t)((H + 2) / 2) * ((W + 2) / 2);
  return N * 3 * C * (size_t)(H + 2) * (W + 2);
}

int pad_fwd(const float* x, float* y, int mode, int N, int C, int H, int W, int T, hipStream_t st) {
  DVSR_REQUIRE(x && y, DVSR_ERR_INVALID, "pad_fwd: null pointer");
  DVSR_REQUIRE(mode >= PAD_REFLECT && mode <= PAD_REPL_T3, DVSR_ERR_INVALID, "pad_fwd: mode %d", mode);
  DVSR_REQUIRE(mode == PAD_REPL_T3 || (H >= 2 && W >= 2), DVSR_ERR_INVALID, "pad_fwd: reflect needs H, W >= 2");
  DVSR_REQUIRE(mode != PAD_REFLECT_S2D || (H % 2 == 0 && W % 2 == 0), DVSR_ERR_INVALID,
               "pad_fwd: space-to-depth needs even H, W (got %dx%d)", H, W);
  DVSR_REQUIRE(mode != PAD_REPL_T3 || (T > 0 && N % T == 0), DVSR_ERR_INVALID, "pad_fwd: N=%d not a multiple of T=%d", N, T);
  const size_t total = pad_out_numel(mode, (size_t)N, C, H, W);
  hipLaunchKernelGGL(pad_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, y, total, mode, C, H, W, T);
  return check_launch("pad_fwd_kernel");
}

int pad_bwd(const float* gy, float* gx, int mode, int N, int C, int H, int W, int T, int accumulate,
            hipStream_t st) {
  DVSR_REQUIRE(gy && gx, DVSR_ERR_INVALID, "pad_bwd: null pointer");
  DVSR_REQUIRE(mode >= PAD_REFLECT && mode <= PAD_REPL_T3, DVSR_ERR_INVALID, "pad_bwd: mode %d", mode);
  const size_t total = (size_t)N * C * H * W;
  hipLaunchKernelGGL(pad_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, gy, gx, total, mode, C, H, W, T,
                     accumulate);
  return check_launch("pad_bwd_kernel");
}

// ---- per-frame mean (LRimg_estimator.py:93: x.mean(-1).mean(-2)) ------------------------------------
// x: [B][C][T][H][W]  ->  xm: [(B*T)][C][H][W] = x - mean,  mean: [B][C][T].  One workgroup per plane.
__global__ __launch_bounds__(256) void meansub_kernel(const float* __restrict__ x, float* __restrict__ xm,
                                                      float* __restrict__ mean, int C, int T, int H, int W) {
  const int plane = blockIdx.x;  // (b*C + c)*T + t
  const int t = plane % T, c = (plane / T) % C, b = plane / (T * C);
  const size_t HW = (size_t)H * W;
  const float* src = x + (size_t)plane * HW;
  // mean over W, then over H, like the reference (row means are averaged)
  __shared__ float s_part[256];
  float acc = 0.f;
  for (int row = threadIdx.x >> 6; row < H; row += 4) {  // one wave per row
    float s = 0.f;
    for (int col = threadIdx.x & 63; col < W; col += 64) s += src[(size_t)row * W + col];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) acc += s / (float)W;
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = (s_part[0] + s_part[64] + s_part[128] + s_part[192]) / (float)H;
    s_part[0] = m;
    mean[plane] = m;
  }
  __syncthreads();
  const float m = s_part[0];
  float* dst = xm + (((size_t)b * T + t) * C + c) * HW;
  for (size_t i = threadIdx.x; i < HW; i += 256) dst[i] = src[i] - m;
}

// out[b][c][t][p] = y[(b*T+t)][c][p] + mean[b][c][t]
__global__ void addmean_kernel(const float* __restrict__ y, const float* __restrict__ mean, float* __restrict__ out,
                               size_t total, int C, int T, size_t HW) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    size_t r = i / HW;  // (b*C + c)*T + t
    const int t = (int)(r % T);
    const size_t bc = r / T;
    const int c = (int)(bc % C);
    const size_t b = bc / C;
    out[i] = y[((b * T + t) * C + c) * HW + p] + mean[r];
  }
}

// gy[(b*T+t)][c][p] = gout[b][c][t][p]
__global__ void addmean_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gy, size_t total, int C, int T,
                                   size_t HW) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    size_t r = i / HW;  // (b*T + t)*C + c
    const int c = (int)(r % C);
    const size_t bt = r / C;
    const int t = (int)(bt % T);
    const size_t b = bt / T;
    gy[i] = gout[((b * C + c) * T + t) * HW + p];
  }
}

int meansub_fwd(const float* x, float* xm, float* mean, int B, int C, int T, int H, int W, hipStream_t st) {
  DVSR_REQUIRE(x && xm && mean, DVSR_ERR_INVALID, "meansub_fwd: null pointer");
  hipLaunchKernelGGL(meansub_kernel, dim3(B * C * T), dim3(256), 0, st, x, xm, mean, C, T, H, W);
  return check_launch("meansub_kernel");
}

int addmean_fwd(const float* y, const float* mean, float* out, int B, int C, int T, size_t HW, hipStream_t st) {
  DVSR_REQUIRE(y && mean && out, DVSR_ERR_INVALID, "addmean_fwd: null pointer");
  const size_t total = (size_t)B * C * T * HW;
  hipLaunchKernelGGL(addmean_kernel, dim3(grid_for(total)), dim3(256), 0, st, y, mean, out, total, C, T, HW);
  return check_launch("addmean_kernel");
}

int addmean_bwd(const float* gout, float* gy, int B, int C, int T, size_t HW, hipStream_t st) {
  DVSR_REQUIRE(gout && gy, DVSR_ERR_INVALID, "addmean_bwd: null pointer");
  const size_t total = (size_t)B * C * T * HW;
  hipLaunchKernelGGL(addmean_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, gout, gy, total, C, T, HW);
  return check_launch("addmean_bwd_kernel");
}

// ---- 4x4 stride-2 weights <-> 2x2 weights over the space-to-depth input -------------------------------
// w2[o][4c + 2dy + dx][ky][kx] = w[o][c][2ky + dy][2kx + dx]   (inverse = 1: the same map, other direction)
__global__ void w4_s2d_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t total, int C,
                              int inverse) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i indexes w2: [o][c4][ky][kx]
    const int kx = (int)(i & 1), ky = (int)((i >> 1) & 1);
    const size_t r = i >> 2;
    const int c4 = (int)(r % (4 * C));
    const size_t o = r / (4 * C);
    const int c = c4 >> 2, dy = (c4 >> 1) & 1, dx = c4 & 1;
    const size_t j = ((o * C + c) * 4 + 2 * ky + dy) * 4 + 2 * kx + dx;
    if (inverse) dst[j] = src[i];
    else dst[i] = src[j];
  }
}

int w4_to_s2d(const float* w, float* w2, int Cout, int C, int inverse, hipStream_t st) {
  DVSR_REQUIRE(w && w2, DVSR_ERR_INVALID, "w4_to_s2d: null pointer");
  const size_t total = (size_t)Cout * C * 16;
  hipLaunchKernelGGL(w4_s2d_kernel, dim3(grid_for(total)), dim3(256), 0, st, w, w2, total, C, inverse);
  return check_launch("w4_s2d_kernel");
}

}  // namespace dvsr
