// Ops of the TOFlow backbone (codes/models/archs/TOF_arch.py:25-140) that are not convolutions:
//   flow_warp            arch_util.py:55-79   bilinear grid_sample (zeros padding, align_corners=False as torch >= 1.3
//                                             defaults) at pixel + flow, the grid normalised with (size - 1)
//   avg_pool2            TOF_arch.py:67-74    F.avg_pool2d(kernel 2, stride 2) -- the image pyramids of SpyNet
//   resize_bilinear_ac   TOF_arch.py:84-85    F.interpolate(size=..., bilinear, align_corners=True) * 2 of the flow
//   batchnorm            TOF_arch.py:33-42    nn.BatchNorm2d (+ the ReLU behind it), training and eval mode
//   channel_affine       TOF_arch.py:13-22    normalize / denormalize, and channel-slice copies (torch.cat)
// All are HBM-bound streaming kernels: fp32 NCHW, one pass over every distinct input and output; reductions
// (batch-norm statistics) are fixed-order two-stage sums in double.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "kernels.h"

namespace dvsr {

static inline int sgrid(size_t n) {
  size_t g = (n + 255) / 256;
  const size_t cap = 256 * 16;
  return (int)(g < cap ? (g ? g : 1) : cap);
}

// ---- flow_warp ------------------------------------------------------------------------------------
// Sampling position of output pixel (y, x): the reference adds the flow to the pixel grid, scales to [-1, 1] with
// (size - 1) and hands it to grid_sample, whose align_corners=False un-normalisation is ((g + 1) * size - 1) / 2 --
// i.e. ix = (x + fx) * W / (W - 1) - 0.5, evaluated in the reference's operation order.
struct WarpPos {
  float ix, iy;
  int x0, y0;
  float wx1, wy1;  // weight of the +1 neighbour
};
__device__ __forceinline__ WarpPos warp_pos(float fx, float fy, int x, int y, int H, int W) {
  const float gx = 2.0f * ((float)x + fx) / (float)(W - 1 > 1 ? W - 1 : 1) - 1.0f;
  const float gy = 2.0f * ((float)y + fy) / (float)(H - 1 > 1 ? H - 1 : 1) - 1.0f;
  WarpPos p;
  p.ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
  p.iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
  const float fx0 = floorf(p.ix), fy0 = floorf(p.iy);
  p.x0 = (int)fx0; p.y0 = (int)fy0;
  p.wx1 = p.ix - fx0; p.wy1 = p.iy - fy0;
  return p;
}

// x [N][C][H][W], flow [N][2][H][W] (channel 0 = dx, 1 = dy), out: batch stride out_bs (a channel slice of a wider
// tensor when the caller builds torch.cat([ref, warped, flow]) in place)
__global__ void flow_warp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ flow, float* __restrict__ out,
                                     int N, int C, int H, int W, long long out_bs) {
  const size_t HW = (size_t)H * W, total = (size_t)N * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW);
    const size_t r = i - (size_t)n * HW;
    const int y = (int)(r / W), xx = (int)(r - (size_t)y * W);
    const float* f = flow + (size_t)n * 2 * HW + r;
    const WarpPos p = warp_pos(f[0], f[HW], xx, y, H, W);
    const bool vx0 = p.x0 >= 0 && p.x0 < W, vx1 = p.x0 + 1 >= 0 && p.x0 + 1 < W;
    const bool vy0 = p.y0 >= 0 && p.y0 < H, vy1 = p.y0 + 1 >= 0 && p.y0 + 1 < H;
    const float wnw = (1.f - p.wx1) * (1.f - p.wy1), wne = p.wx1 * (1.f - p.wy1), wsw = (1.f - p.wx1) * p.wy1,
                wse = p.wx1 * p.wy1;
    const float* xn = x + (size_t)n * C * HW;
    float* on = out + (size_t)n * out_bs + r;
    for (int c = 0; c < C; ++c) {
      const float* pl = xn + (size_t)c * HW;
      float v = 0.f;
      if (vy0 && vx0) v += pl[(size_t)p.y0 * W + p.x0] * wnw;
      if (vy0 && vx1) v += pl[(size_t)p.y0 * W + p.x0 + 1] * wne;
      if (vy1 && vx0) v += pl[(size_t)(p.y0 + 1) * W + p.x0] * wsw;
      if (vy1 && vx1) v += pl[(size_t)(p.y0 + 1) * W + p.x0 + 1] * wse;
      on[(size_t)c * HW] = v;
    }
  }
}

// gx (zeroed by the caller) += scatter of gout; gflow [N][2][H][W] written.  gout has batch stride g_bs.
__global__ void flow_warp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                     const float* __restrict__ gout, float* __restrict__ gx, float* __restrict__ gflow,
                                     int N, int C, int H, int W, long long g_bs) {
  const size_t HW = (size_t)H * W, total = (size_t)N * HW;
  const float sx = (float)W / (float)(W - 1 > 1 ? W - 1 : 1), sy = (float)H / (float)(H - 1 > 1 ? H - 1 : 1);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW);
    const size_t r = i - (size_t)n * HW;
    const int y = (int)(r / W), xx = (int)(r - (size_t)y * W);
    const float* f = flow + (size_t)n * 2 * HW + r;
    const WarpPos p = warp_pos(f[0], f[HW], xx, y, H, W);
    const bool vx0 = p.x0 >= 0 && p.x0 < W, vx1 = p.x0 + 1 >= 0 && p.x0 + 1 < W;
    const bool vy0 = p.y0 >= 0 && p.y0 < H, vy1 = p.y0 + 1 >= 0 && p.y0 + 1 < H;
    const float ax = 1.f - p.wx1, bx = p.wx1, ay = 1.f - p.wy1, by = p.wy1;
    const float* xn = x + (size_t)n * C * HW;
    const float* gn = gout + (size_t)n * g_bs + r;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
      const float g = gn[(size_t)c * HW];
      const float* pl = xn + (size_t)c * HW;
      float* gp = gx ? gx + ((size_t)n * C + c) * HW : nullptr;
      const float nw = (vy0 && vx0) ? pl[(size_t)p.y0 * W + p.x0] : 0.f;
      const float ne = (vy0 && vx1) ? pl[(size_t)p.y0 * W + p.x0 + 1] : 0.f;
      const float sw = (vy1 && vx0) ? pl[(size_t)(p.y0 + 1) * W + p.x0] : 0.f;
      const float se = (vy1 && vx1) ? pl[(size_t)(p.y0 + 1) * W + p.x0 + 1] : 0.f;
      gix += g * ((ne - nw) * ay + (se - sw) * by);
      giy += g * ((sw - nw) * ax + (se - ne) * bx);
      if (gp) {
        if (vy0 && vx0) unsafeAtomicAdd(gp + (size_t)p.y0 * W + p.x0, g * ax * ay);
        if (vy0 && vx1) unsafeAtomicAdd(gp + (size_t)p.y0 * W + p.x0 + 1, g * bx * ay);
        if (vy1 && vx0) unsafeAtomicAdd(gp + (size_t)(p.y0 + 1) * W + p.x0, g * ax * by);
        if (vy1 && vx1) unsafeAtomicAdd(gp + (size_t)(p.y0 + 1) * W + p.x0 + 1, g * bx * by);
      }
    }
    if (gflow) {
      float* gf = gflow + (size_t)n * 2 * HW + r;
      gf[0] = gix * sx;
      gf[HW] = giy * sy;
    }
  }
}

// ---- avg_pool2d(kernel 2, stride 2) ----------------------------------------------------------------
__global__ void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t planes, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = planes * Ho * Wo;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oy = (int)(t % Ho);
    const size_t pl = t / Ho;
    const float* p = x + pl * H * W + (size_t)(2 * oy) * W + 2 * ox;
    y[i] = (p[0] + p[1] + p[W] + p[W + 1]) * 0.25f;
  }
}
__global__ void avgpool2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, size_t planes, int H, int W,
                                    int accumulate) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = planes * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    const size_t t = i / W;
    const int y = (int)(t % H);
    const size_t pl = t / H;
    const float v = (y / 2 < Ho && xx / 2 < Wo) ? gy[(pl * Ho + y / 2) * Wo + xx / 2] * 0.25f : 0.f;
    gx[i] = accumulate ? gx[i] + v : v;
  }
}

// ---- F.interpolate(size=(Ho, Wo), mode='bilinear', align_corners=True) * mul ---------------------------
__device__ __forceinline__ void ac_src(int o, int in, int out, int& i0, int& i1, float& l) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float s = scale * (float)o;
  i0 = (int)s;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  l = s - (float)i0;
}
__global__ void resize_ac_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t planes, int H, int W,
                                     int Ho, int Wo, float mul, long long y_ps) {
  const size_t total = planes * Ho * Wo;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oy = (int)(t % Ho);
    const size_t pl = t / Ho;
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(oy, H, Ho, y0, y1, ly);
    ac_src(ox, W, Wo, x0, x1, lx);
    const float* p = x + pl * H * W;
    const float v = (1.f - ly) * ((1.f - lx) * p[(size_t)y0 * W + x0] + lx * p[(size_t)y0 * W + x1]) +
                    ly * ((1.f - lx) * p[(size_t)y1 * W + x0] + lx * p[(size_t)y1 * W + x1]);
    y[pl * y_ps + (size_t)oy * Wo + ox] = v * mul;
  }
}
// gx zeroed by the caller; scatter with atomics (the flow fields are 2 channels: tiny)
__global__ void resize_ac_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, size_t planes, int H, int W,
                                     int Ho, int Wo, float mul, long long gy_ps) {
  const size_t total = planes * Ho * Wo;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oy = (int)(t % Ho);
    const size_t pl = t / Ho;
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(oy, H, Ho, y0, y1, ly);
    ac_src(ox, W, Wo, x0, x1, lx);
    const float g = gy[pl * gy_ps + (size_t)oy * Wo + ox] * mul;
    float* p = gx + pl * H * W;
    unsafeAtomicAdd(p + (size_t)y0 * W + x0, g * (1.f - ly) * (1.f - lx));
    unsafeAtomicAdd(p + (size_t)y0 * W + x1, g * (1.f - ly) * lx);
    unsafeAtomicAdd(p + (size_t)y1 * W + x0, g * ly * (1.f - lx));
    unsafeAtomicAdd(p + (size_t)y1 * W + x1, g * ly * lx);
  }
}

// ---- F.interpolate(x, scale_factor=S, mode='bicubic', align_corners=True) -------------------------------
// The drivers up-sample the (S)LR clip to the output size before TOFlow (test_dynavsr.py:188-193, 245-250).  ATen's
// upsample_bicubic2d: source = dst * (in-1)/(out-1), 4x4 taps with the cubic-convolution kernel A = -0.75, indices
// clamped to the image.
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
template <bool BWD>
__global__ void bicubic_ac_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ gx,
                                  const float* __restrict__ gy, size_t planes, int H, int W, int Ho, int Wo) {
  const size_t total = planes * Ho * Wo;
  const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oy = (int)(t % Ho);
    const size_t pl = t / Ho;
    const float ry = sh * (float)oy, rx = sw * (float)ox;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float cy[4], cx[4];
    cubic_coeffs(ry - (float)iy, cy);
    cubic_coeffs(rx - (float)ix, cx);
    float acc = 0.f;
    const float g = BWD ? gy[i] : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), H - 1);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(ix - 1 + b, 0), W - 1);
        const size_t o = pl * H * W + (size_t)yy * W + xx;
        if (BWD) unsafeAtomicAdd(gx + o, g * cy[a] * cx[b]);
        else acc += x[o] * cy[a] * cx[b];
      }
    }
    if (!BWD) y[i] = acc;
  }
}

// ---- out[n][c][:] = x[n][c][:] * scale[c] + shift[c]  (batch strides: channel slices of wider tensors) ------
__global__ void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                      const float* __restrict__ shift, float* __restrict__ out, int N, int C, size_t HW,
                                      long long x_bs, long long out_bs, int accumulate) {
  const size_t total = (size_t)N * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    const size_t t = i / HW;
    const int c = (int)(t % C);
    const size_t n = t / C;
    float v = x[n * x_bs + (size_t)c * HW + p];
    if (scale) v *= scale[c];
    if (shift) v += shift[c];
    float* o = out + n * out_bs + (size_t)c * HW + p;
    *o = accumulate ? *o + v : v;
  }
}

// ---- BatchNorm2d (+ ReLU) ---------------------------------------------------------------------------
constexpr int BN_SLICES = 64;  // partial sums per channel
// part[(c * BN_SLICES + s) * 2 + {0, 1}] = sum / sum of squares of x over slice s of (n, hw)
__global__ void bn_stats_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int N, int C, size_t HW) {
  const int c = blockIdx.x, s = blockIdx.y;
  const size_t M = (size_t)N * HW;
  double a = 0.0, b = 0.0;
  for (size_t i = (size_t)s * blockDim.x + threadIdx.x; i < M; i += (size_t)BN_SLICES * blockDim.x) {
    const size_t n = i / HW, p = i - n * HW;
    const double v = x[(n * C + c) * HW + p];
    a += v; b += v * v;
  }
  __shared__ double ra[256], rb[256];
  ra[threadIdx.x] = a; rb[threadIdx.x] = b;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) { ra[threadIdx.x] += ra[threadIdx.x + k]; rb[threadIdx.x] += rb[threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((size_t)c * BN_SLICES + s) * 2] = ra[0]; part[((size_t)c * BN_SLICES + s) * 2 + 1] = rb[0]; }
}
// mean / rstd of the batch (biased variance for the normalisation, unbiased for the running estimate, as torch)
__global__ void bn_stats_final_kernel(const double* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd,
                                      float* __restrict__ running_mean, float* __restrict__ running_var, int C, double M,
                                      float momentum, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < BN_SLICES; ++s) { a += part[((size_t)c * BN_SLICES + s) * 2]; b += part[((size_t)c * BN_SLICES + s) * 2 + 1]; }
  const double m = a / M;
  double var = b / M - m * m;
  var = var > 0.0 ? var : 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(M > 1.0 ? var * M / (M - 1.0) : var);
}
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float* __restrict__ mean, float* __restrict__ rstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  rstd[c] = 1.f / sqrtf(running_var[c] + eps);
}
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                int N, int C, size_t HW, int relu) {
  const size_t total = (size_t)N * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    float v = (x[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    if (relu) v = v > 0.f ? v : 0.f;
    y[i] = v;
  }
}
// backward, stage 1: part[(c*S + s)*2 + {0,1}] = sum g', sum g' * xhat   with g' = gy * [y > 0] when relu
__global__ void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ y,
                                      const float* __restrict__ mean, const float* __restrict__ rstd, double* __restrict__ part,
                                      int N, int C, size_t HW, int relu) {
  const int c = blockIdx.x, s = blockIdx.y;
  const size_t M = (size_t)N * HW;
  const float m = mean[c], r = rstd[c];
  double a = 0.0, b = 0.0;
  for (size_t i = (size_t)s * blockDim.x + threadIdx.x; i < M; i += (size_t)BN_SLICES * blockDim.x) {
    const size_t n = i / HW, p = i - n * HW;
    const size_t o = (n * C + c) * HW + p;
    float g = gy[o];
    if (relu && !(y[o] > 0.f)) g = 0.f;
    a += (double)g;
    b += (double)g * (double)((x[o] - m) * r);
  }
  __shared__ double ra[256], rb[256];
  ra[threadIdx.x] = a; rb[threadIdx.x] = b;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) { ra[threadIdx.x] += ra[threadIdx.x + k]; rb[threadIdx.x] += rb[threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((size_t)c * BN_SLICES + s) * 2] = ra[0]; part[((size_t)c * BN_SLICES + s) * 2 + 1] = rb[0]; }
}
__global__ void bn_bwd_final_kernel(const double* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    float* __restrict__ sums, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < BN_SLICES; ++s) { a += part[((size_t)c * BN_SLICES + s) * 2]; b += part[((size_t)c * BN_SLICES + s) * 2 + 1]; }
  if (dbeta) dbeta[c] = (float)a;
  if (dgamma) dgamma[c] = (float)b;
  sums[2 * c] = (float)a; sums[2 * c + 1] = (float)b;
}
// stage 2: training: gx = gamma*rstd * (g' - sum_g/M - xhat * sum_gx/M);  eval: gx = gamma*rstd * g'
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ sums, float* __restrict__ gx,
                                    int N, int C, size_t HW, int relu, int training) {
  const size_t total = (size_t)N * C * HW;
  const float invM = 1.f / (float)((size_t)N * HW);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    float g = gy[i];
    if (relu && !(y[i] > 0.f)) g = 0.f;
    const float k = gamma[c] * rstd[c];
    if (training) {
      const float xh = (x[i] - mean[c]) * rstd[c];
      g = g - sums[2 * c] * invM - xh * sums[2 * c + 1] * invM;
    }
    gx[i] = k * g;
  }
}

}  // namespace dvsr

using namespace dvsr;

#define TOF_LAUNCH(kern, n, st, ...) hipLaunchKernelGGL(kern, dim3(sgrid(n)), dim3(256), 0, st, __VA_ARGS__)

extern "C" int dvsr_flow_warp_forward(const float* x, const float* flow, float* out, int N, int C, int H, int W,
                                      long long out_bstride, dvsr_stream_t stream) {
  DVSR_REQUIRE(x && flow && out && N > 0 && C > 0 && H > 0 && W > 0, DVSR_ERR_INVALID, "flow_warp_forward: bad argument");
  const long long obs = out_bstride > 0 ? out_bstride : (long long)C * H * W;
  TOF_LAUNCH(flow_warp_fwd_kernel, (size_t)N * H * W, (hipStream_t)stream, x, flow, out, N, C, H, W, obs);
  return check_launch("flow_warp_fwd_kernel");
}

extern "C" int dvsr_flow_warp_backward(const float* x, const float* flow, const float* grad_out, float* grad_x,
                                       float* grad_flow, int N, int C, int H, int W, long long gout_bstride,
                                       dvsr_stream_t stream) {
  DVSR_REQUIRE(x && flow && grad_out && (grad_x || grad_flow) && N > 0 && C > 0 && H > 0 && W > 0, DVSR_ERR_INVALID,
               "flow_warp_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (grad_x)
    DVSR_REQUIRE(hipMemsetAsync(grad_x, 0, (size_t)N * C * H * W * sizeof(float), st) == hipSuccess, DVSR_ERR_HIP,
                 "flow_warp_backward: memset failed");
  const long long gbs = gout_bstride > 0 ? gout_bstride : (long long)C * H * W;
  TOF_LAUNCH(flow_warp_bwd_kernel, (size_t)N * H * W, st, x, flow, grad_out, grad_x, grad_flow, N, C, H, W, gbs);
  return check_launch("flow_warp_bwd_kernel");
}

extern "C" int dvsr_avgpool2_forward(const float* x, float* y, long long planes, int H, int W, dvsr_stream_t stream) {
  DVSR_REQUIRE(x && y && planes > 0 && H >= 2 && W >= 2, DVSR_ERR_INVALID, "avgpool2_forward: bad argument");
  TOF_LAUNCH(avgpool2_fwd_kernel, (size_t)planes * (H / 2) * (W / 2), (hipStream_t)stream, x, y, (size_t)planes, H, W);
  return check_launch("avgpool2_fwd_kernel");
}

extern "C" int dvsr_avgpool2_backward(const float* grad_y, float* grad_x, long long planes, int H, int W, int accumulate,
                                      dvsr_stream_t stream) {
  DVSR_REQUIRE(grad_y && grad_x && planes > 0 && H >= 2 && W >= 2, DVSR_ERR_INVALID, "avgpool2_backward: bad argument");
  TOF_LAUNCH(avgpool2_bwd_kernel, (size_t)planes * H * W, (hipStream_t)stream, grad_y, grad_x, (size_t)planes, H, W, accumulate);
  return check_launch("avgpool2_bwd_kernel");
}

extern "C" int dvsr_resize_bilinear_ac_forward(const float* x, float* y, long long planes, int H, int W, int Ho, int Wo,
                                               float mul, long long y_plane_stride, dvsr_stream_t stream) {
  DVSR_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, DVSR_ERR_INVALID,
               "resize_bilinear_ac_forward: bad argument");
  const long long ps = y_plane_stride > 0 ? y_plane_stride : (long long)Ho * Wo;
  TOF_LAUNCH(resize_ac_fwd_kernel, (size_t)planes * Ho * Wo, (hipStream_t)stream, x, y, (size_t)planes, H, W, Ho, Wo, mul, ps);
  return check_launch("resize_ac_fwd_kernel");
}

extern "C" int dvsr_resize_bilinear_ac_backward(const float* grad_y, float* grad_x, long long planes, int H, int W, int Ho,
                                                int Wo, float mul, long long gy_plane_stride, dvsr_stream_t stream) {
  DVSR_REQUIRE(grad_y && grad_x && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, DVSR_ERR_INVALID,
               "resize_bilinear_ac_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  DVSR_REQUIRE(hipMemsetAsync(grad_x, 0, (size_t)planes * H * W * sizeof(float), st) == hipSuccess, DVSR_ERR_HIP,
               "resize_bilinear_ac_backward: memset failed");
  const long long ps = gy_plane_stride > 0 ? gy_plane_stride : (long long)Ho * Wo;
  TOF_LAUNCH(resize_ac_bwd_kernel, (size_t)planes * Ho * Wo, st, grad_y, grad_x, (size_t)planes, H, W, Ho, Wo, mul, ps);
  return check_launch("resize_ac_bwd_kernel");
}

extern "C" int dvsr_upsample_bicubic_ac_forward(const float* x, float* y, long long planes, int H, int W, int scale,
                                                dvsr_stream_t stream) {
  DVSR_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && scale >= 1, DVSR_ERR_INVALID, "upsample_bicubic_ac_forward: bad argument");
  TOF_LAUNCH(bicubic_ac_kernel<false>, (size_t)planes * H * W * scale * scale, (hipStream_t)stream, x, y, nullptr, nullptr,
             (size_t)planes, H, W, H * scale, W * scale);
  return check_launch("bicubic_ac_kernel");
}

extern "C" int dvsr_upsample_bicubic_ac_backward(const float* grad_y, float* grad_x, long long planes, int H, int W,
                                                 int scale, dvsr_stream_t stream) {
  DVSR_REQUIRE(grad_y && grad_x && planes > 0 && H > 0 && W > 0 && scale >= 1, DVSR_ERR_INVALID,
               "upsample_bicubic_ac_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  DVSR_REQUIRE(hipMemsetAsync(grad_x, 0, (size_t)planes * H * W * sizeof(float), st) == hipSuccess, DVSR_ERR_HIP,
               "upsample_bicubic_ac_backward: memset failed");
  TOF_LAUNCH(bicubic_ac_kernel<true>, (size_t)planes * H * W * scale * scale, st, nullptr, nullptr, grad_x, grad_y,
             (size_t)planes, H, W, H * scale, W * scale);
  return check_launch("bicubic_ac_kernel(bwd)");
}

extern "C" int dvsr_channel_affine(const float* x, const float* scale, const float* shift, float* out, int N, int C,
                                   long long HW, long long x_bstride, long long out_bstride, int accumulate,
                                   dvsr_stream_t stream) {
  DVSR_REQUIRE(x && out && N > 0 && C > 0 && HW > 0, DVSR_ERR_INVALID, "channel_affine: bad argument");
  const long long xb = x_bstride > 0 ? x_bstride : (long long)C * HW, ob = out_bstride > 0 ? out_bstride : (long long)C * HW;
  TOF_LAUNCH(channel_affine_kernel, (size_t)N * C * HW, (hipStream_t)stream, x, scale, shift, out, N, C, (size_t)HW, xb, ob,
             accumulate);
  return check_launch("channel_affine_kernel");
}

extern "C" size_t dvsr_batchnorm_workspace_bytes(int C) { return (size_t)C * BN_SLICES * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float); }

// y = relu?( (x - mean) * rstd * gamma + beta ).  training: batch statistics (written to save_mean / save_rstd for the
// backward; running estimates updated in place with `momentum`, unbiased variance, as nn.BatchNorm2d); eval: the
// running estimates.  save_mean / save_rstd: [C] each.
extern "C" int dvsr_batchnorm_forward(const float* x, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float* y, float* save_mean, float* save_rstd, int N, int C,
                                      long long HW, int training, float momentum, float eps, int relu, void* workspace,
                                      size_t workspace_bytes, dvsr_stream_t stream) {
  DVSR_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && N > 0 && C > 0 && HW > 0, DVSR_ERR_INVALID,
               "batchnorm_forward: bad argument");
  DVSR_REQUIRE(training || (running_mean && running_var), DVSR_ERR_INVALID, "batchnorm_forward: eval mode needs running stats");
  hipStream_t st = (hipStream_t)stream;
  if (training) {
    DVSR_REQUIRE(workspace && workspace_bytes >= dvsr_batchnorm_workspace_bytes(C), DVSR_ERR_WORKSPACE,
                 "batchnorm_forward: workspace too small");
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, BN_SLICES), dim3(256), 0, st, x, (double*)workspace, N, C, (size_t)HW);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, (const double*)workspace, save_mean,
                       save_rstd, running_mean, running_var, C, (double)N * (double)HW, momentum, eps);
  } else {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, running_mean, running_var, save_mean,
                       save_rstd, C, eps);
  }
  TOF_LAUNCH(bn_apply_kernel, (size_t)N * C * HW, st, x, save_mean, save_rstd, gamma, beta, y, N, C, (size_t)HW, relu);
  return check_launch("batchnorm_forward");
}

// grad_y: gradient w.r.t. the (post-ReLU when relu) output y; y needed only when relu.
extern "C" int dvsr_batchnorm_backward(const float* x, const float* grad_y, const float* y, const float* gamma,
                                       const float* save_mean, const float* save_rstd, float* grad_x, float* grad_gamma,
                                       float* grad_beta, int N, int C, long long HW, int training, int relu,
                                       void* workspace, size_t workspace_bytes, dvsr_stream_t stream) {
  DVSR_REQUIRE(x && grad_y && gamma && save_mean && save_rstd && grad_x && (!relu || y) && N > 0 && C > 0 && HW > 0,
               DVSR_ERR_INVALID, "batchnorm_backward: bad argument");
  DVSR_REQUIRE(workspace && workspace_bytes >= dvsr_batchnorm_workspace_bytes(C), DVSR_ERR_WORKSPACE,
               "batchnorm_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace;
  float* sums = (float*)(part + (size_t)C * BN_SLICES * 2);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(C, BN_SLICES), dim3(256), 0, st, x, grad_y, y, save_mean, save_rstd, part, N,
                     C, (size_t)HW, relu);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, (const double*)part, grad_gamma, grad_beta,
                     sums, C);
  TOF_LAUNCH(bn_bwd_apply_kernel, (size_t)N * C * HW, st, x, grad_y, y, save_mean, save_rstd, gamma, (const float*)sums, grad_x,
             N, C, (size_t)HW, relu, training);
  return check_launch("batchnorm_backward");
}
