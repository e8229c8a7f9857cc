// Micro-benchmark (MI355X): what the two waves of a SIMD share.  512-thread workgroups (waves w and w + 4 sit on one SIMD),
// one workgroup per CU.  Waves 0-3 run `ma` iterations of an MFMA stream, waves 4-7 `mb` iterations of a second stream
// (kind: 0 nothing, 1 v_fma_f32 chains, 2 bf16 MFMA, 3 ds_read_b128, 4 v_cvt_pk + shifts (the split's instruction mix),
// 5 ds_write_b32).  Prints wave-0 and wave-4 cycles alone and together.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int KA, int KB>
__global__ __launch_bounds__(512, 2) void k(long long* cyc, float* out, int ia, int ib) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 1.f + i * 1e-6f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kind = wave < 4 ? KA : KB, iters = wave < 4 ? ia : ib;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a4 = {1.f + lane * 1e-6f, 2.f, 3.f, 4.f}, b4 = {1.f, 1.f, 1.f, 1.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.f + i + lane;
  const unsigned la = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds) + lane * 16;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (kind == 1) {
#pragma unroll
      for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u & 7]) : "v"(b4[0]));
    } else if (kind == 2) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a4), "v"(b4));
    } else if (kind == 3) {
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a4) : "v"(la), "i"((u & 15) * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (kind == 4) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        unsigned h;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(v[u & 7]), "v"(v[(u + 1) & 7]));
        unsigned lo, hi;
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(lo) : "v"(h));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi) : "v"(h));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[u & 7]) : "v"(lo));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[(u + 1) & 7]) : "v"(hi));
      }
    } else if (kind == 5) {
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(la), "v"(v[u & 7]), "i"((u & 15) * 1024) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  for (int i = 0; i < 8; ++i) s += v[i];
  s += a4[0] + a4[3];
  if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}

template <int KA, int KB>
void run(const char* name, int ia, int ib, long long* dcyc, float* dout, double per_a, double per_b) {
  long long h[2];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(512), 0, 0, dcyc, dout, ia, ib);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s wave0 %9lld cyc (%7.1f per unit)   wave4 %9lld cyc (%7.1f per unit)\n", name, h[0], ia ? h[0] / (ia * per_a) : 0.0, h[1],
         ib ? h[1] / (ib * per_b) : 0.0);
}

int main() {
  long long* dcyc; float* dout;
  hipMalloc(&dcyc, 16); hipMalloc(&dout, 256 * 512 * 4);
  const int N = 2000;
  run<2, 0>("bf16 MFMA alone (unit = 1 MFMA)", N, 0, dcyc, dout, 16, 1);
  run<0, 1>("v_fma alone (unit = 1 VALU)", 0, N, dcyc, dout, 1, 64);
  run<2, 1>("bf16 MFMA | v_fma on the partner wave", N, N, dcyc, dout, 16, 64);
  run<2, 2>("bf16 MFMA | bf16 MFMA", N, N, dcyc, dout, 16, 16);
  run<0, 3>("ds_read_b128 alone (unit = 1 read)", 0, N, dcyc, dout, 1, 16);
  run<2, 3>("bf16 MFMA | ds_read_b128", N, N, dcyc, dout, 16, 16);
  run<0, 4>("cvt/shift/sub mix alone (unit = 5 VALU)", 0, N, dcyc, dout, 1, 16);
  run<2, 4>("bf16 MFMA | cvt/shift/sub mix", N, N, dcyc, dout, 16, 16);
  run<0, 5>("ds_write_b32 alone (unit = 1 write)", 0, N, dcyc, dout, 1, 16);
  run<2, 5>("bf16 MFMA | ds_write_b32", N, N, dcyc, dout, 16, 16);
  run<1, 1>("v_fma | v_fma", N, N, dcyc, dout, 64, 64);
  run<1, 3>("v_fma | ds_read_b128", N, N, dcyc, dout, 64, 16);
  run<4, 3>("cvt mix | ds_read_b128", N, N, dcyc, dout, 16, 16);
  return 0;
}
