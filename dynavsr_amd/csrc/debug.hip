// Measurement aid: register-only v_mfma_f32_32x32x2_f32 loop with the conv kernel's launch shape
// (256-thread workgroups, NACC independent accumulators per wave).  Gives the MFMA ceiling that the
// conv kernels can be compared against on the box they run on.
#include "common.h"
#include "kernels.h"

namespace dvsr {
template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace dvsr

// Returns the number of MFMA instructions each wave executes (iters * 8 * nacc); grid x 4 waves.
extern "C" long long dvsr_debug_mfma_peak(float* out, int blocks, int iters, int nacc, int lds_bytes,
                                          dvsr_stream_t stream) {
  using namespace dvsr;
  hipStream_t st = (hipStream_t)stream;
  if (nacc == 2) hipLaunchKernelGGL(mfma_peak_kernel<2>, dim3(blocks), dim3(256), lds_bytes, st, out, iters, 1.f, 1.f);
  else if (nacc == 4) hipLaunchKernelGGL(mfma_peak_kernel<4>, dim3(blocks), dim3(256), lds_bytes, st, out, iters, 1.f, 1.f);
  else hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(256), lds_bytes, st, out, iters, 1.f, 1.f);
  if (check_launch("mfma_peak_kernel")) return -1;
  return (long long)iters * 8 * (nacc == 2 || nacc == 4 ? nacc : 1);
}
