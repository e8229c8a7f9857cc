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

// How many other instructions fit in the shadow of a 64-cycle v_mfma_f32_32x32x2_f32 of the SAME wave: a stream
// of MFMAs on four accumulators with NV independent v_fma_f32 (KIND 0) or ds_read_b32 (KIND 1) behind each one,
// the accumulators in VGPRs (ACC 0) or AGPRs (ACC 1).  Every thread reports its cycle count.
template <int NV, int KIND, int ACC>
__global__ __launch_bounds__(256, 1) void mfma_shadow_kernel(long long* cycles, float* out, int iters, float a0) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a0; lds[threadIdx.x + 512] = a0; lds[threadIdx.x + 768] = a0;
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = a0;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = a0 + i;
  const unsigned la = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds) + (threadIdx.x & 63) * 4;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (ACC) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k & 15]) : "v"(b));
        else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[k & 15]) : "v"(la), "i"((k & 15) * 256));
      }
    }
    if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
}  // namespace dvsr

// cycles[0] = shader cycles wave 0 of workgroup 0 spent on iters x 16 MFMAs, each followed by nv instructions of
// `kind` (0 v_fma_f32, 1 ds_read_b32); acc = 1 keeps the accumulators in AGPRs.  nv in {0, 4, 8, 12, 16}.
extern "C" int dvsr_debug_mfma_shadow(long long* cycles, float* out, int blocks, int iters, int nv, int kind, int acc,
                                      dvsr_stream_t stream) {
  using namespace dvsr;
  hipStream_t st = (hipStream_t)stream;
#define DVSR_SH(NV, K, A)                                                                                      \
  if (nv == NV && kind == K && acc == A) {                                                                      \
    hipLaunchKernelGGL((mfma_shadow_kernel<NV, K, A>), dim3(blocks), dim3(256), 0, st, cycles, out, iters, 1.f); \
    return check_launch("mfma_shadow_kernel");                                                                  \
  }
#define DVSR_SH4(NV) DVSR_SH(NV, 0, 0) DVSR_SH(NV, 0, 1) DVSR_SH(NV, 1, 0) DVSR_SH(NV, 1, 1)
  DVSR_SH4(0) DVSR_SH4(4) DVSR_SH4(8) DVSR_SH4(12) DVSR_SH4(16)
#undef DVSR_SH4
#undef DVSR_SH
  DVSR_REQUIRE(false, DVSR_ERR_INVALID, "debug_mfma_shadow: nv in {0,4,8,12,16}, kind / acc in {0,1}");
}

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
