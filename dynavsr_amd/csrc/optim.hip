// Inner-loop optimiser steps over lists of parameter tensors (test_dynavsr.py:223-231 builds
// torch.optim.Adam(lr_alpha, betas=(0.9, 0.99)) or torch.optim.SGD(lr_alpha) over the ~158 tensors of
// netG + netE and steps it once per inner iteration).  One launch per 48 tensors (blockIdx.y = tensor)
// instead of a dozen foreach launches plus the per-step tensor grouping of the framework optimiser.
// Arithmetic follows torch.optim's single-tensor formulas: Adam without amsgrad / maximize,
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g g;  p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// with L2 weight decay folded into g first; SGD without momentum: p -= lr (g + wd p).
#include <cmath>

#include "common.h"
#include "kernels.h"

namespace dvsr {

constexpr int OPT_BATCH = 48;
struct OptEntry { float* p; const float* g; float* m; float* v; long long n; };
struct OptTable { int count; OptEntry e[OPT_BATCH]; };

__global__ void adam_step_kernel(OptTable t, float lr_over_bc1, float beta1, float beta2, float inv_sqrt_bc2, float eps,
                                 float weight_decay) {
  const OptEntry& e = t.e[blockIdx.y];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < e.n; i += (long long)gridDim.x * blockDim.x) {
    float g = e.g[i];
    const float p = e.p[i];
    if (weight_decay != 0.f) g += weight_decay * p;
    const float m = beta1 * e.m[i] + (1.f - beta1) * g;
    const float v = beta2 * e.v[i] + (1.f - beta2) * g * g;
    e.m[i] = m;
    e.v[i] = v;
    e.p[i] = p - lr_over_bc1 * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
  }
}

__global__ void sgd_step_kernel(OptTable t, float lr, float weight_decay) {
  const OptEntry& e = t.e[blockIdx.y];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < e.n; i += (long long)gridDim.x * blockDim.x) {
    const float p = e.p[i];
    e.p[i] = p - lr * (e.g[i] + weight_decay * p);
  }
}

// dst[i][c][:] = src[i][:] for c < copies: the per-frame deep copies of the networks (test_dynavsr.py:208) for a batch
// of frames, every tensor's copies stacked along a leading axis (dynavsr_amd/adapt.py: FrameBatch).  e.g = source.
__global__ void replicate_kernel(OptTable t) {
  const OptEntry& e = t.e[blockIdx.y];
  float* const dst = e.p + (long long)blockIdx.z * e.n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < e.n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = e.g[i];
}

}  // namespace dvsr

using namespace dvsr;

extern "C" int dvsr_replicate_tensors(const float* const* src, float* const* dst, const long long* numel, int n_tensors,
                                      int copies, dvsr_stream_t stream) {
  DVSR_REQUIRE(src && dst && numel && n_tensors >= 0 && copies >= 1, DVSR_ERR_INVALID, "replicate_tensors: bad argument");
  OptTable t;
  t.count = 0;
  auto flush = [&]() -> int {
    if (!t.count) return DVSR_OK;
    hipLaunchKernelGGL(replicate_kernel, dim3(16, t.count, copies), dim3(256), 0, (hipStream_t)stream, t);
    t.count = 0;
    return check_launch("replicate_kernel");
  };
  for (int i = 0; i < n_tensors; ++i) {
    if (numel[i] <= 0) continue;
    DVSR_REQUIRE(src[i] && dst[i], DVSR_ERR_INVALID, "replicate_tensors: null tensor %d", i);
    t.e[t.count++] = OptEntry{dst[i], src[i], nullptr, nullptr, numel[i]};
    if (t.count == OPT_BATCH) { int rc = flush(); if (rc) return rc; }
  }
  return flush();
}

extern "C" int dvsr_adam_step(float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const long long* numel, int n_tensors, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, dvsr_stream_t stream) {
  DVSR_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && n_tensors >= 0 && step >= 1, DVSR_ERR_INVALID,
               "adam_step: bad argument (step counts from 1)");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const float lr_over_bc1 = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  OptTable t;
  t.count = 0;
  auto flush = [&]() -> int {
    if (!t.count) return DVSR_OK;
    hipLaunchKernelGGL(adam_step_kernel, dim3(32, t.count), dim3(256), 0, (hipStream_t)stream, t, lr_over_bc1, beta1, beta2,
                       inv_sqrt_bc2, eps, weight_decay);
    t.count = 0;
    return check_launch("adam_step_kernel");
  };
  for (int i = 0; i < n_tensors; ++i) {
    if (numel[i] <= 0 || !grads[i]) continue;  // parameter without a gradient this step: untouched, like torch.optim
    DVSR_REQUIRE(params[i] && exp_avg[i] && exp_avg_sq[i], DVSR_ERR_INVALID, "adam_step: null tensor %d", i);
    t.e[t.count++] = OptEntry{params[i], grads[i], exp_avg[i], exp_avg_sq[i], numel[i]};
    if (t.count == OPT_BATCH) { int rc = flush(); if (rc) return rc; }
  }
  return flush();
}

extern "C" int dvsr_sgd_step(float* const* params, const float* const* grads, const long long* numel, int n_tensors,
                             float lr, float weight_decay, dvsr_stream_t stream) {
  DVSR_REQUIRE(params && grads && numel && n_tensors >= 0, DVSR_ERR_INVALID, "sgd_step: bad argument");
  OptTable t;
  t.count = 0;
  auto flush = [&]() -> int {
    if (!t.count) return DVSR_OK;
    hipLaunchKernelGGL(sgd_step_kernel, dim3(32, t.count), dim3(256), 0, (hipStream_t)stream, t, lr, weight_decay);
    t.count = 0;
    return check_launch("sgd_step_kernel");
  };
  for (int i = 0; i < n_tensors; ++i) {
    if (numel[i] <= 0 || !grads[i]) continue;
    DVSR_REQUIRE(params[i], DVSR_ERR_INVALID, "sgd_step: null tensor %d", i);
    t.e[t.count++] = OptEntry{params[i], grads[i], nullptr, nullptr, numel[i]};
    if (t.count == OPT_BATCH) { int rc = flush(); if (rc) return rc; }
  }
  return flush();
}
