// Probe (GPU box): accuracy of fp32 GEMM emulated with split-bf16 MFMAs vs the exact fp32 MFMA, and the
// instruction rates.  C[32x32] = A[32xK] * B[Kx32], K = 576 (one 64-channel 3x3 conv output), data ~ N(0,1)
// and a second set with wide dynamic range.  build+run: hipcc --offload-arch=gfx950 -O3 tools/bf16_split_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; float r = x - (float)h; m = (__bf16)r; r -= (float)m; l = (__bf16)r;
}
// mode 0: fp32 MFMA; 1: bf16; 2: 2-way split, 3 products; 3: 3-way split, 6 products
__global__ void gemm(const float* A, const float* B, float* C, int K, int mode) {
  const int lane = threadIdx.x, lo = lane & 31, hi = lane >> 5;
  f32x16 acc = {0};
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[lo * K + k + hi], B[(k + hi) * 32 + lo], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      bf16x8 a[3], b[3];
      for (int i = 0; i < 8; ++i) {
        const float av = A[lo * K + k0 + 8 * hi + i], bv = B[(k0 + 8 * hi + i) * 32 + lo];
        __bf16 h, m, l;
        split3(av, h, m, l); a[0][i] = h; a[1][i] = m; a[2][i] = l;
        split3(bv, h, m, l); b[0][i] = h; b[1][i] = m; b[2][i] = l;
      }
      if (mode == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
      else {
        // small terms first
        if (mode == 3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
      }
    }
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lo] = acc[r];
}
__global__ void rate(float* out, int iters, int mode) {
  f32x16 acc[4] = {};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it)
    for (int j = 0; j < 4; ++j) {
      if (mode == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.f, 2.f, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
  float s = 0; for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 123.456f) out[0] = s;
}
int main() {
  const int K = 576;
  for (int set = 0; set < 2; ++set) {
    std::vector<float> A(32 * K), B(K * 32), C(1024);
    srand(1 + set);
    auto rnd = [&]() { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.f; };
    for (auto& v : A) v = rnd() * (set ? expf(4.f * rnd()) : 1.f);
    for (auto& v : B) v = rnd() * (set ? expf(4.f * rnd()) : 1.f);
    std::vector<double> R(1024, 0.0), Rabs(1024, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < K; ++k) {
      R[i * 32 + j] += (double)A[i * K + k] * B[k * 32 + j]; Rabs[i * 32 + j] += fabs((double)A[i * K + k] * B[k * 32 + j]); }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"fp32 MFMA 32x32x2", "bf16 x1", "bf16 2-way split (3 products)", "bf16 3-way split (6 products)"};
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
      hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
      double num = 0, den = 0, worst = 0;
      for (int i = 0; i < 1024; ++i) { num += (C[i] - R[i]) * (C[i] - R[i]); den += R[i] * R[i]; worst = fmax(worst, fabs(C[i] - R[i]) / Rabs[i]); }
      printf("set %d  %-32s rel-L2 %.3e   max |err| / sum|terms| %.3e\n", set, names[mode], sqrt(num / den), worst);
    }
  }
  float* d; hipMalloc(&d, 4);
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 4000;
    hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, d, 10, mode);
    hipEventRecord(s); hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, d, iters, mode); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double inst = 1024.0 * 4 * iters * 4;
    printf("%s: %.1f G MFMA/s  -> %.1f TFLOP/s\n", mode ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32", inst / ms / 1e6,
           inst * (mode ? 32768.0 : 4096.0) / ms / 1e9);
  }
  return 0;
}
