// Per-frame image metrics on the device.  The reference's test-time-adaptation loop
// (codes/test_dynavsr.py:285-292, and the validation loops of train_dynavsr.py) copies every super-resolved
// frame to the host as fp32, converts it with util.tensor2img (codes/utils/util.py:112-142: clamp to
// [min,max], rescale, x255, round-half-even, uint8 HWC) and evaluates util.calculate_psnr (:262-269) and
// util.calculate_ssim (:271-313: 11x11 Gaussian window sigma 1.5 via cv2.filter2D in float64, "valid" region,
// mean over pixels and channels) with numpy / cv2.  Here the same arithmetic runs where the frame already is:
//   frame_quant_kernel   both frames -> uint8 (planar for the SSIM pass, optionally HWC for the PNG writer),
//                        sum of squared differences as an exact 64-bit integer
//   ssim_tile_kernel     separable 11-tap Gaussian of x, y, x^2, y^2, xy in float64 over 32x16-pixel tiles
//                        (the 2-D window of the reference is the outer product of the same 11 taps), SSIM map,
//                        one partial sum per workgroup
//   metrics_finalize     fixed-order sum of the partials -> {mse, mean ssim} (deterministic, no fp atomics)
// Byte / integer work bound by HBM: 2 x 4 B read per sample, 2(+1) B written, the uint8 planes re-read once.
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "common.h"
#include "kernels.h"

namespace dvsr {

// tensor2img: clamp, rescale to [0,1], x 255, round half to even -- all in fp32 like torch / numpy do it
__device__ __forceinline__ int quant_u8(float v, float lo, float hi) {
  const float t = (fminf(fmaxf(v, lo), hi) - lo) / (hi - lo);
  return (int)rintf(t * 255.0f);
}

// One thread = 4 consecutive samples of every channel plane (VEC) or one sample (ragged sizes / unaligned
// views): 16-byte loads, one dword store per uint8 plane, 4*C contiguous bytes of the HWC image.  The squared
// differences are summed as integers per workgroup (no atomics: ~14 k same-address 64-bit atomics cost 150 us).
template <bool VEC>
__global__ __launch_bounds__(256) void frame_quant_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          int C, long long HW, float lo, float hi,
                                                          unsigned char* __restrict__ qx, unsigned char* __restrict__ qy,
                                                          unsigned char* __restrict__ x_hwc,
                                                          unsigned long long* __restrict__ sse_partials) {
  __shared__ unsigned long long s_red[4];
  unsigned long long acc = 0;
  constexpr int V = VEC ? 4 : 1;
  const long long n = HW / V;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
    unsigned char hwc[16];
    for (int c = 0; c < C; ++c) {
      float xv[V], yv[V];
      if (VEC) {
        const float4 a4 = *reinterpret_cast<const float4*>(x + c * HW + 4 * p);
        const float4 b4 = *reinterpret_cast<const float4*>(y + c * HW + 4 * p);
        xv[0] = a4.x; xv[V > 1 ? 1 : 0] = a4.y; xv[V > 2 ? 2 : 0] = a4.z; xv[V > 3 ? 3 : 0] = a4.w;
        yv[0] = b4.x; yv[V > 1 ? 1 : 0] = b4.y; yv[V > 2 ? 2 : 0] = b4.z; yv[V > 3 ? 3 : 0] = b4.w;
      } else {
        xv[0] = x[c * HW + p]; yv[0] = y[c * HW + p];
      }
      unsigned pa = 0, pb = 0;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int a = quant_u8(xv[i], lo, hi), b = quant_u8(yv[i], lo, hi);
        pa |= (unsigned)a << (8 * i); pb |= (unsigned)b << (8 * i);
        hwc[i * 4 + c] = (unsigned char)a;  // C <= 4
        const int d = a - b;
        acc += (unsigned long long)(d * d);
      }
      if (VEC) {
        *reinterpret_cast<unsigned*>(qx + c * HW + 4 * p) = pa;
        *reinterpret_cast<unsigned*>(qy + c * HW + 4 * p) = pb;
      } else {
        qx[c * HW + p] = (unsigned char)pa; qy[c * HW + p] = (unsigned char)pb;
      }
    }
    if (x_hwc) {
      if (VEC && C == 3) {  // 4 RGB pixels = 12 contiguous, 4-byte aligned bytes: three dword stores
        unsigned char b[12];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 3; ++c) b[i * 3 + c] = hwc[i * 4 + c];
        unsigned* dst = reinterpret_cast<unsigned*>(x_hwc + 12 * p);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          dst[q] = (unsigned)b[4 * q] | ((unsigned)b[4 * q + 1] << 8) | ((unsigned)b[4 * q + 2] << 16) | ((unsigned)b[4 * q + 3] << 24);
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i)
          for (int c = 0; c < C; ++c) x_hwc[(V * p + i) * C + c] = hwc[i * 4 + c];
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sse_partials[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

struct GaussTaps { double w[11]; };

constexpr int ST_W = 32, ST_H = 16, ST_IW = ST_W + 10, ST_IH = ST_H + 10;

__global__ __launch_bounds__(256) void ssim_tile_kernel(const unsigned char* __restrict__ qx,
                                                        const unsigned char* __restrict__ qy, int H, int W,
                                                        GaussTaps g, double* __restrict__ partials) {
  __shared__ unsigned char s_x[ST_IH][ST_IW + 2], s_y[ST_IH][ST_IW + 2];
  __shared__ double s_h[5][ST_IH][ST_W];  // horizontal pass of x, y, xx, yy, xy
  __shared__ double s_red[4];
  const int c = blockIdx.z;
  const int ox0 = blockIdx.x * ST_W, oy0 = blockIdx.y * ST_H;  // first valid-output pixel of the tile = input origin
  const unsigned char* px = qx + (size_t)c * H * W;
  const unsigned char* py = qy + (size_t)c * H * W;
  {  // all 2 x 5 byte loads of a thread are issued before the first LDS write (one latency, not five)
    constexpr int NL = (ST_IH * ST_IW + 255) / 256;
    unsigned char vx[NL], vy[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int r = i / ST_IW, col = i - r * ST_IW;
      const int gy = oy0 + r, gx = ox0 + col;
      const bool ok = i < ST_IH * ST_IW && gy < H && gx < W;
      const size_t off = ok ? (size_t)gy * W + gx : 0;
      vx[k] = px[off]; vy[k] = py[off];
      if (!ok) { vx[k] = 0; vy[k] = 0; }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int r = i / ST_IW, col = i - r * ST_IW;
      if (i < ST_IH * ST_IW) { s_x[r][col] = vx[k]; s_y[r][col] = vy[k]; }
    }
  }
  __syncthreads();
  // horizontal pass: one thread = 4 consecutive columns of a row (14 bytes of each frame converted once)
  for (int i = threadIdx.x; i < ST_IH * (ST_W / 4); i += 256) {
    const int r = i / (ST_W / 4), col = (i - r * (ST_W / 4)) * 4;
    double a[14], b[14];
#pragma unroll
    for (int t = 0; t < 14; ++t) { a[t] = (double)s_x[r][col + t]; b[t] = (double)s_y[r][col + t]; }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
      for (int t = 0; t < 11; ++t) {
        const double w = g.w[t], u = a[o + t], v = b[o + t];
        sx += w * u; sy += w * v; sxx += w * (u * u); syy += w * (v * v); sxy += w * (u * v);
      }
      s_h[0][r][col + o] = sx; s_h[1][r][col + o] = sy; s_h[2][r][col + o] = sxx; s_h[3][r][col + o] = syy;
      s_h[4][r][col + o] = sxy;
    }
  }
  __syncthreads();
  const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
  double acc = 0;
  {  // vertical pass: one thread = 2 consecutive rows of a column (12 rows of the 5 maps read once)
    const int col = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 2;
    double m[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
#pragma unroll
    for (int t = 0; t < 12; ++t) {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const double v = s_h[q][r0 + t][col];
        if (t < 11) m[0][q] += g.w[t] * v;
        if (t > 0) m[1][q] += g.w[t - 1] * v;
      }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (oy0 + r0 + o >= H - 10 || ox0 + col >= W - 10) continue;  // outside the "valid" region
      const double mu1 = m[o][0], mu2 = m[o][1];
      const double mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const double s1 = m[o][2] - mu1_sq, s2 = m[o][3] - mu2_sq, s12 = m[o][4] - mu12;
      acc += ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(256) void metrics_finalize_kernel(const double* __restrict__ partials, int n,
                                                               const unsigned long long* __restrict__ sse_partials,
                                                               int n_sse, double n_samples, double n_valid,
                                                               double* __restrict__ out) {
  __shared__ double s[256];
  __shared__ unsigned long long si[256];
  double acc = 0;
  unsigned long long iacc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];  // fixed order per thread
  for (int i = threadIdx.x; i < n_sse; i += 256) iacc += sse_partials[i];
  s[threadIdx.x] = acc;
  si[threadIdx.x] = iacc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { s[threadIdx.x] += s[threadIdx.x + off]; si[threadIdx.x] += si[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (double)si[0] / n_samples;
    out[1] = n_valid > 0 ? s[0] / n_valid : nan("");  // numpy: mean of an empty map
  }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace dvsr

using namespace dvsr;

constexpr int QUANT_BLOCKS = 1024;

extern "C" size_t dvsr_frame_metrics_workspace_bytes(int C, int H, int W) {
  if (C <= 0 || H <= 0 || W <= 0) return 0;
  const size_t plane = align256((size_t)C * H * W);
  const size_t tiles = (size_t)ceil_div(W > 10 ? W - 10 : 1, ST_W) * ceil_div(H > 10 ? H - 10 : 1, ST_H) * C;
  return 2 * plane + align256(QUANT_BLOCKS * sizeof(unsigned long long)) + align256(tiles * sizeof(double));
}

extern "C" int dvsr_frame_metrics(const float* sr, const float* gt, int C, int H, int W, float lo, float hi,
                                  unsigned char* sr_hwc_u8, double* out, void* workspace, size_t workspace_bytes,
                                  dvsr_stream_t stream) {
  DVSR_REQUIRE(sr && gt && out && workspace, DVSR_ERR_INVALID, "frame_metrics: null sr/gt/out/workspace");
  DVSR_REQUIRE(C >= 1 && C <= 4 && H >= 1 && W >= 1 && hi > lo, DVSR_ERR_INVALID,
               "frame_metrics: C=%d H=%d W=%d range [%g, %g]", C, H, W, (double)lo, (double)hi);
  DVSR_REQUIRE(workspace_bytes >= dvsr_frame_metrics_workspace_bytes(C, H, W), DVSR_ERR_WORKSPACE,
               "frame_metrics: workspace %zu < %zu bytes", workspace_bytes, dvsr_frame_metrics_workspace_bytes(C, H, W));
  DVSR_REQUIRE(((uintptr_t)workspace & 15) == 0, DVSR_ERR_INVALID, "frame_metrics: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const size_t plane = align256((size_t)C * H * W);
  unsigned char* qx = (unsigned char*)workspace;
  unsigned char* qy = qx + plane;
  unsigned long long* sse_partials = (unsigned long long*)(qy + plane);
  double* partials = (double*)((unsigned char*)sse_partials + align256(QUANT_BLOCKS * sizeof(unsigned long long)));
  const long long HW = (long long)H * W;
  const bool vec = HW % 4 == 0 && (((uintptr_t)sr | (uintptr_t)gt) & 15) == 0;
  const int qblocks = (int)std::min<long long>(((vec ? HW / 4 : HW) + 255) / 256, QUANT_BLOCKS);
  if (vec)
    hipLaunchKernelGGL(frame_quant_kernel<true>, dim3(qblocks), dim3(256), 0, st, sr, gt, C, HW, lo, hi, qx, qy,
                       sr_hwc_u8, sse_partials);
  else
    hipLaunchKernelGGL(frame_quant_kernel<false>, dim3(qblocks), dim3(256), 0, st, sr, gt, C, HW, lo, hi, qx, qy,
                       sr_hwc_u8, sse_partials);
  int rc = check_launch("frame_quant_kernel");
  if (rc) return rc;
  int ntiles = 0;
  if (H > 10 && W > 10) {
    GaussTaps g;  // cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised to sum 1
    double sum = 0;
    for (int i = 0; i < 11; ++i) { g.w[i] = std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g.w[i]; }
    for (int i = 0; i < 11; ++i) g.w[i] /= sum;
    const dim3 grid(ceil_div(W - 10, ST_W), ceil_div(H - 10, ST_H), C);
    ntiles = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(ssim_tile_kernel, grid, dim3(256), 0, st, qx, qy, H, W, g, partials);
    rc = check_launch("ssim_tile_kernel");
    if (rc) return rc;
  }
  const double n_valid = (H > 10 && W > 10) ? (double)(H - 10) * (W - 10) * C : 0.0;
  hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(256), 0, st, partials, ntiles, sse_partials, qblocks,
                     (double)C * H * W, n_valid, out);
  return check_launch("metrics_finalize_kernel");
}
