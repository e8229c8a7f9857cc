// Ops of the DUF backbone (codes/models/archs/DUF_arch.py:32-180) beyond convolutions and BatchNorm:
//   temporal_gather3   the (3,3,3) Conv3d of the dense blocks (:45-58) = a 3x3 conv2d over 3C channels after gathering
//                      frames t-1, t, t+1 (zero padding in time, or none when the block reduces T): the weight
//                      [Cout][C][3][3][3] IS the [Cout][3C][3][3] tensor that needs (channel index c*3 + kt);
//   dynamic_filter     F.softmax over the 25 generated filter taps (:166), DynamicUpsamplingFilter_3C (:86-110: the
//                      5x5 local patch of the centre frame times the per-pixel filters), the image residual Rx with
//                      its adapt_official channel order (:17-29,168-170) and F.pixel_shuffle (:175), in ONE kernel:
//                      the reference materialises the [B,75,H,W] patch tensor, two permuted copies and the
//                      [B,25,R,H,W] softmax.
// fp32 NCHW, frames as the batch axis ([B*T][C][H][W]); HBM-bound streaming kernels.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "kernels.h"

namespace dvsr {

static inline int dgrid(size_t n) {
  size_t g = (n + 255) / 256;
  const size_t cap = 256 * 16;
  return (int)(g < cap ? (g ? g : 1) : cap);
}

// y[(b*To + t)][c*3 + kt][p] = x[(b*T + t + kt - pad)][c][p]  (0 outside 0 <= frame < T);  To = T + 2*pad - 2
__global__ void temporal_gather3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int C,
                                            size_t HW, int pad) {
  const int To = T + 2 * pad - 2;
  const size_t total = (size_t)B * To * C * 3 * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    size_t r = i / HW;
    const int kt = (int)(r % 3); r /= 3;
    const int c = (int)(r % C); r /= C;
    const int t = (int)(r % To);
    const int b = (int)(r / To);
    const int ts = t + kt - pad;
    y[i] = (ts >= 0 && ts < T) ? x[(((size_t)b * T + ts) * C + c) * HW + p] : 0.f;
  }
}
// gx[(b*T + ts)][c][p] = sum_kt gy[(b*To + ts - kt + pad)][c*3 + kt][p]
__global__ void temporal_gather3_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int B, int T, int C,
                                            size_t HW, int pad) {
  const int To = T + 2 * pad - 2;
  const size_t total = (size_t)B * T * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    size_t r = i / HW;
    const int c = (int)(r % C); r /= C;
    const int ts = (int)(r % T);
    const int b = (int)(r / T);
    float s = 0.f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int t = ts - kt + pad;
      if (t >= 0 && t < To) s += gy[((((size_t)b * To + t) * C + c) * 3 + kt) * HW + p];
    }
    gx[i] = s;
  }
}

// Channel of the residual tensor that pairs with output channel (c, r): the reference reorders Rx in place when the
// weights come from the official TensorFlow model (adapt_official, DUF_arch.py:17-29): Rx'[c*R + r] = Rx[3r + c].
__device__ __forceinline__ int rx_channel(int c, int r, int R, int adapt) { return adapt ? 3 * r + c : c * R + r; }

// One thread = one (b, r, y, x): softmax over the 25 logits fx[b][f*R + r][y][x], the 5x5 patch of the three colour
// planes of the centre frame (zero padded), + residual, stored at the pixel-shuffled position.
__global__ void dynamic_filter_fwd_kernel(const float* __restrict__ xc, const float* __restrict__ fx,
                                          const float* __restrict__ rx, float* __restrict__ out, int B, int H, int W,
                                          int S, int flags) {
  const int adapt = flags & 1, raw = flags & 2;
  const int R = S * S;
  const size_t HW = (size_t)H * W, total = (size_t)B * R * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    const int r = (int)((i / HW) % R);
    const int b = (int)(i / (HW * R));
    const int y = (int)(p / W), x = (int)(p % W);
    const float* f = fx + ((size_t)b * 25 * R + r) * HW + p;
    float l[25], m = -3.4e38f;
#pragma unroll
    for (int k = 0; k < 25; ++k) { l[k] = f[(size_t)k * R * HW]; m = fmaxf(m, l[k]); }
    float inv = 1.f;
    if (!raw) {   // raw: the 25 values ARE the filter taps (DynamicUpsamplingFilter_3C as a module applies them as given)
      float den = 0.f;
#pragma unroll
      for (int k = 0; k < 25; ++k) { l[k] = expf(l[k] - m); den += l[k]; }
      inv = 1.f / den;
    }
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 25; ++k) {
      const int yy = y + k / 5 - 2, xx = x + k % 5 - 2;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        const float w = l[k] * inv;
        const float* q = xc + (size_t)b * 3 * HW + (size_t)yy * W + xx;
        acc[0] += w * q[0]; acc[1] += w * q[HW]; acc[2] += w * q[2 * HW];
      }
    }
    const int oy = S * y + r / S, ox = S * x + r % S;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = acc[c] + rx[((size_t)b * 3 * R + rx_channel(c, r, R, adapt)) * HW + p];
      out[(((size_t)b * 3 + c) * (S * H) + oy) * (size_t)(S * W) + ox] = v;
    }
  }
}

// gfx (logits) and grx written in full; gxc (optional, zeroed by the caller) accumulated with atomics.
__global__ void dynamic_filter_bwd_kernel(const float* __restrict__ xc, const float* __restrict__ fx,
                                          const float* __restrict__ gout, float* __restrict__ gfx, float* __restrict__ grx,
                                          float* __restrict__ gxc, int B, int H, int W, int S, int flags) {
  const int adapt = flags & 1, raw = flags & 2;
  const int R = S * S;
  const size_t HW = (size_t)H * W, total = (size_t)B * R * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    const int r = (int)((i / HW) % R);
    const int b = (int)(i / (HW * R));
    const int y = (int)(p / W), x = (int)(p % W);
    const float* f = fx + ((size_t)b * 25 * R + r) * HW + p;
    float l[25], m = -3.4e38f;
#pragma unroll
    for (int k = 0; k < 25; ++k) { l[k] = f[(size_t)k * R * HW]; m = fmaxf(m, l[k]); }
    float inv = 1.f;
    if (!raw) {
      float den = 0.f;
#pragma unroll
      for (int k = 0; k < 25; ++k) { l[k] = expf(l[k] - m); den += l[k]; }
      inv = 1.f / den;
    }
    const int oy = S * y + r / S, ox = S * x + r % S;
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      g[c] = gout[(((size_t)b * 3 + c) * (S * H) + oy) * (size_t)(S * W) + ox];
      grx[((size_t)b * 3 * R + rx_channel(c, r, R, adapt)) * HW + p] = g[c];
    }
    float dp[25], dot = 0.f;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
      const int yy = y + k / 5 - 2, xx = x + k % 5 - 2;
      float d = 0.f;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        const size_t o = (size_t)b * 3 * HW + (size_t)yy * W + xx;
        d = g[0] * xc[o] + g[1] * xc[o + HW] + g[2] * xc[o + 2 * HW];
        if (gxc) {
          const float w = l[k] * inv;
          unsafeAtomicAdd(gxc + o, w * g[0]);
          unsafeAtomicAdd(gxc + o + HW, w * g[1]);
          unsafeAtomicAdd(gxc + o + 2 * HW, w * g[2]);
        }
      }
      dp[k] = d;
      dot += l[k] * inv * d;
    }
    float* gf = gfx + ((size_t)b * 25 * R + r) * HW + p;
#pragma unroll
    for (int k = 0; k < 25; ++k) gf[(size_t)k * R * HW] = raw ? dp[k] : l[k] * inv * (dp[k] - dot);
  }
}

}  // namespace dvsr

using namespace dvsr;

extern "C" int dvsr_temporal_gather3_forward(const float* x, float* y, int B, int T, int C, long long HW, int pad_t,
                                             dvsr_stream_t stream) {
  DVSR_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && (pad_t == 0 || pad_t == 1) && T + 2 * pad_t - 2 > 0, DVSR_ERR_INVALID,
               "temporal_gather3_forward: bad argument (T=%d pad_t=%d)", T, pad_t);
  const size_t n = (size_t)B * (T + 2 * pad_t - 2) * C * 3 * HW;
  hipLaunchKernelGGL(temporal_gather3_fwd_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, x, y, B, T, C,
                     (size_t)HW, pad_t);
  return check_launch("temporal_gather3_fwd_kernel");
}

extern "C" int dvsr_temporal_gather3_backward(const float* grad_y, float* grad_x, int B, int T, int C, long long HW,
                                              int pad_t, dvsr_stream_t stream) {
  DVSR_REQUIRE(grad_y && grad_x && B > 0 && C > 0 && HW > 0 && (pad_t == 0 || pad_t == 1) && T + 2 * pad_t - 2 > 0,
               DVSR_ERR_INVALID, "temporal_gather3_backward: bad argument (T=%d pad_t=%d)", T, pad_t);
  const size_t n = (size_t)B * T * C * HW;
  hipLaunchKernelGGL(temporal_gather3_bwd_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, grad_y, grad_x, B, T, C,
                     (size_t)HW, pad_t);
  return check_launch("temporal_gather3_bwd_kernel");
}

extern "C" int dvsr_dynamic_filter_forward(const float* x_center, const float* filter_logits, const float* residual,
                                           float* out, int B, int H, int W, int scale, int adapt_official,
                                           dvsr_stream_t stream) {
  DVSR_REQUIRE(x_center && filter_logits && residual && out && B > 0 && H > 0 && W > 0 && scale >= 1 && scale <= 4,
               DVSR_ERR_INVALID, "dynamic_filter_forward: bad argument");
  const size_t n = (size_t)B * scale * scale * H * W;
  hipLaunchKernelGGL(dynamic_filter_fwd_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)stream, x_center, filter_logits,
                     residual, out, B, H, W, scale, adapt_official);
  return check_launch("dynamic_filter_fwd_kernel");
}

extern "C" int dvsr_dynamic_filter_backward(const float* x_center, const float* filter_logits, const float* grad_out,
                                            float* grad_logits, float* grad_residual, float* grad_x_center, int B, int H,
                                            int W, int scale, int adapt_official, dvsr_stream_t stream) {
  DVSR_REQUIRE(x_center && filter_logits && grad_out && grad_logits && grad_residual && B > 0 && H > 0 && W > 0 &&
                   scale >= 1 && scale <= 4, DVSR_ERR_INVALID, "dynamic_filter_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (grad_x_center)
    DVSR_REQUIRE(hipMemsetAsync(grad_x_center, 0, (size_t)B * 3 * H * W * sizeof(float), st) == hipSuccess, DVSR_ERR_HIP,
                 "dynamic_filter_backward: memset failed");
  const size_t n = (size_t)B * scale * scale * H * W;
  hipLaunchKernelGGL(dynamic_filter_bwd_kernel, dim3(dgrid(n)), dim3(256), 0, st, x_center, filter_logits, grad_out, grad_logits,
                     grad_residual, grad_x_center, B, H, W, scale, adapt_official);
  return check_launch("dynamic_filter_bwd_kernel");
}
