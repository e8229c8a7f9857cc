// Micro-benchmark (MI355X): does running the two waves of a SIMD in OPPOSITE block orders pay when a workgroup barrier
// closes every iteration?  512-thread workgroups (waves w, w + 4 share a SIMD), one per CU; every wave runs `iters` of
//   M: NM bf16 MFMAs on four accumulators,  V: NV VALU instructions in NC independent dependency chains,
//   L: NL ds_read_b128 (waited for at the top of the next M),  then s_barrier;
// order 0: every wave M L V | barrier;  order 1: waves 4-7 run V M L | barrier (anti-phase);  order 2: as 0 without barrier;
// order 3: as 1 without barrier.  Prints cycles per iteration of wave 0.
//   hipcc --offload-arch=gfx950 -O3 tools/wave_phase.hip -o /tmp/wave_phase && /tmp/wave_phase
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NC, int NL>
__global__ __launch_bounds__(512, 2) void k(long long* cyc, float* out, int iters, int order) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = 1.f + i * 1e-6f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool vfirst = (order & 1) && wave >= 4, bar = order < 2;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a4 = {1.f + lane * 1e-6f, 2.f, 3.f, 4.f}, b4 = {1.f, 1.f, 1.f, 1.f}, l4[4];
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 1.f + i + lane;
  const unsigned la = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds) + lane * 16;
  auto M = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < NM; ++u) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a4), "v"(b4));
  };
  auto V = [&]() {
#pragma unroll
    for (int u = 0; u < NV; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u % NC]) : "v"(b4[0]));
  };
  auto L = [&]() {
#pragma unroll
    for (int u = 0; u < NL; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l4[u & 3]) : "v"(la), "i"((u & 15) * 1024));
  };
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (vfirst) { V(); M(); L(); } else { M(); L(); V(); }
    if (bar) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15] + l4[i][0];
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0 && wave == 0) cyc[0] = t1 - t0;
}

template <int NM, int NV, int NC, int NL>
void run(const char* name, long long* dcyc, float* dout) {
  const int N = 2000;
  printf("%-46s", name);
  for (int order = 0; order < 4; ++order) {
    long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k<NM, NV, NC, NL>), dim3(256), dim3(512), 0, 0, dcyc, dout, N, order);
      hipDeviceSynchronize();
    }
    hipMemcpy(&h, dcyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("  %s %7.0f", order == 0 ? "sym+bar" : order == 1 ? "anti+bar" : order == 2 ? "sym" : "anti", (double)h / N);
  }
  printf("   (cycles per iteration)\n");
}

int main() {
  long long* dcyc; float* dout;
  hipMalloc(&dcyc, 16); hipMalloc(&dout, 256 * 512 * 4);
  run<12, 0, 8, 0>("12 MFMA", dcyc, dout);
  run<0, 60, 8, 0>("60 VALU (8 chains)", dcyc, dout);
  run<12, 60, 8, 0>("12 MFMA + 60 VALU (8 chains)", dcyc, dout);
  run<12, 60, 2, 0>("12 MFMA + 60 VALU (2 chains)", dcyc, dout);
  run<12, 60, 8, 14>("12 MFMA + 60 VALU (8 chains) + 14 b128", dcyc, dout);
  run<12, 60, 2, 14>("12 MFMA + 60 VALU (2 chains) + 14 b128", dcyc, dout);
  run<12, 120, 4, 14>("12 MFMA + 120 VALU (4 chains) + 14 b128", dcyc, dout);
  run<6, 30, 4, 7>("6 MFMA + 30 VALU (4 chains) + 7 b128", dcyc, dout);
  return 0;
}
