// Degradation.apply on the device: codes/data/random_kernel_generator.py:84-130 blurs and
// down-samples a clip with ONE 2-D kernel per frame (the centre-of-mass-shifted anisotropic Gaussian, 21 taps
// padded to ~27 by kernel_shift :51-76): ReflectionPad2d(K // 2), then conv2d(groups = 3, stride = scale) with
// the same K x K weights for every channel.  vsrbase.py:184-186 follows it with the 8-bit quantisation
// mul(255).clamp(0, 255).round().div(255) before the second application (LR -> SLR); that step is an option of
// the same launch.  In DDP meta-training this runs twice per sample inside the dataloader workers on the CPU.
//
// One workgroup = a 16 x 16 output tile of one (frame, channel) plane: the ((16-1) s + K)^2 input patch is
// gathered into LDS through the reflection index map (no padded copy in HBM), the K x K weights next to it;
// each thread then walks the taps in row-major order with fp32 FMAs.  HBM-bound by construction: every input
// sample is read ~once (plus the tile halo), every output written once.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dvsr {

constexpr int DG_T = 16;        // output tile edge
constexpr int DG_MAXK = 33;     // largest (shifted) kernel edge
constexpr int DG_MAXS = 4;      // largest stride

__device__ __forceinline__ int reflect_index(int i, int n) {  // ReflectionPad2d: -1 -> 1, n -> n - 2 (pad < n)
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

__global__ __launch_bounds__(256) void degrade_kernel(const float* __restrict__ x, const float* __restrict__ kern,
                                                      float* __restrict__ y, int N, int C, int H, int W, int K, int S,
                                                      int Ho, int Wo, int kern_per_frame, int kern_frames,
                                                      int kern_offset, int quantise) {
  extern __shared__ float smem[];
  const int P = (DG_T - 1) * S + K;  // input patch edge
  float* s_in = smem;                // [P][P + 1]
  float* s_k = smem + P * (P + 1);   // [K][K]
  const int plane = blockIdx.z, n = plane / C;
  const int oy0 = blockIdx.y * DG_T, ox0 = blockIdx.x * DG_T;
  const int pad = K / 2;
  const float* xp = x + (size_t)plane * H * W;
  // frame n uses kernel (n + kern_offset) mod kern_frames when there is one kernel per frame
  // (random_kernel_generator.py:105-113: offset -1 for DUF's two extra frames), else kernel 0
  const float* kp = kern + (size_t)(kern_per_frame ? ((n + kern_offset) % kern_frames + kern_frames) % kern_frames : 0) * K * K;
  for (int i = threadIdx.x; i < K * K; i += 256) s_k[i] = kp[i];
  const int iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;
  {  // a wave per patch row, two passes of 64 columns (P <= 93); four rows = 8 loads in flight per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cx[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int gx = reflect_index(ix0 + lane + 64 * p, W);
      // rows / columns only needed by outputs beyond the image edge may reflect out of range: clamp (never used)
      cx[p] = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    }
    for (int r0 = wave * 4; r0 < P; r0 += 16) {
      float v[4][2];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int gy = reflect_index(iy0 + r0 + rr, H);
        const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
#pragma unroll
        for (int p = 0; p < 2; ++p) v[rr][p] = xp[(size_t)cy * W + cx[p]];
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (r0 + rr >= P) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
          if (lane + 64 * p < P) s_in[(r0 + rr) * (P + 1) + lane + 64 * p] = v[rr][p];
      }
    }
  }
  __syncthreads();
  const int ty = threadIdx.x / DG_T, tx = threadIdx.x - ty * DG_T;
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy >= Ho || ox >= Wo) return;
  float acc = 0.f;
  const float* row = s_in + (ty * S) * (P + 1) + tx * S;
  for (int ky = 0; ky < K; ++ky) {
    for (int kx = 0; kx < K; ++kx) acc = fmaf(s_k[ky * K + kx], row[kx], acc);
    row += P + 1;
  }
  if (quantise) acc = rintf(fminf(fmaxf(acc * 255.f, 0.f), 255.f)) / 255.f;  // mul(255).clamp(0,255).round().div(255)
  y[((size_t)plane * Ho + oy) * Wo + ox] = acc;
}

// Register-tiled variant for the (stride, taps-per-phase) pairs the datasets produce (21-tap Gaussian: K = 27 at
// scale 4 -> J = 7, K = 25 at scale 2 -> J = 13, ...).  The generic kernel above spends two LDS instructions per
// FMA (input + broadcast weight); here the input patch sits in LDS as S x S PHASE PLANES (space-to-depth by the
// stride: tap (ky, kx) of output (oy, ox) is element (oy + ky / S, ox + kx / S) of plane (ky % S, kx % S)), so
// inside a plane the operation is a dense J x J correlation with unit stride and a thread can own FOUR
// consecutive outputs of a row: per (phase, jy) it reads 4 + J - 1 inputs as aligned 16-byte vectors and the J
// weights of that row as broadcast vectors -- 5 LDS instructions for 4 J FMAs at J = 7.
template <int S, int J>
__global__ __launch_bounds__(128) void degrade_tiled_kernel(const float* __restrict__ x, const float* __restrict__ kern,
                                                            float* __restrict__ y, int N, int C, int H, int W, int K,
                                                            int Ho, int Wo, int kern_per_frame, int kern_frames,
                                                            int kern_offset, int quantise) {
  constexpr int TW = 32, TH = 16;
  constexpr int NV = (4 + J - 1 + 3) / 4;            // float4 input vectors per thread and row
  constexpr int QW = (TW - 4) + 4 * NV, QH = TH + J - 1, JP = (J + 3) / 4 * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_in = smem;                   // [S*S][QH][QW], zero outside the patch
  float* s_k = smem + S * S * QH * QW;  // [S*S][J][JP], zero where the tap does not exist
  const int PH = (TH - 1) * S + K, PW = (TW - 1) * S + K;
  const int plane = blockIdx.z, n = plane / C;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int pad = K / 2;
  const float* xp = x + (size_t)plane * H * W;
  const float* kp = kern + (size_t)(kern_per_frame ? ((n + kern_offset) % kern_frames + kern_frames) % kern_frames : 0) * K * K;
  for (int i = threadIdx.x; i < S * S * J * JP; i += 128) {
    const int jx = i % JP, jy = (i / JP) % J, ph = i / (JP * J);
    const int ky = jy * S + ph / S, kx = jx * S + ph % S;
    s_k[i] = (jx < J && ky < K && kx < K) ? kp[ky * K + kx] : 0.f;
  }
  for (int i = threadIdx.x; i < S * S * QH * QW / 4; i += 128) reinterpret_cast<float4*>(s_in)[i] = float4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;
  // a wave per patch row, 64 consecutive columns per pass (PW <= 157: three passes); sixteen rows = 48 loads are in
  // flight before the first LDS write (a loop with one load per iteration pays one memory latency per iteration)
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cx[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int gx = reflect_index(ix0 + lane + 64 * p, W);
      // rows / columns only needed by outputs beyond the image edge may reflect out of range: clamp (never used)
      cx[p] = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    }
    constexpr int RB = 16;  // rows per batch: 48 loads in flight per lane
    for (int r0 = wave * RB; r0 < PH; r0 += 2 * RB) {
      float v[RB][3];
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int gy = reflect_index(iy0 + r0 + rr, H);
        const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        const float* xr = xp + (size_t)cy * W;
#pragma unroll
        for (int p = 0; p < 3; ++p) v[rr][p] = xr[cx[p]];
      }
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int r = r0 + rr;
        if (r >= PH) break;
        float* dst = s_in + (((r % S) * S) * QH + r / S) * QW;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const int c = lane + 64 * p;
          if (c < PW) dst[(c % S) * QH * QW + c / S] = v[rr][p];
        }
      }
    }
  }
  __syncthreads();
  const int gx4 = (threadIdx.x & 7) * 4, row = threadIdx.x >> 3;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ph = 0; ph < S * S; ++ph) {
    const float* pl = s_in + ((size_t)ph * QH + row) * QW + gx4;
    const float* wk = s_k + ph * J * JP;
#pragma unroll
    for (int jy = 0; jy < J; ++jy) {
      float in[4 * NV], w[JP];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 t = *reinterpret_cast<const float4*>(pl + jy * QW + 4 * v);
        in[4 * v] = t.x; in[4 * v + 1] = t.y; in[4 * v + 2] = t.z; in[4 * v + 3] = t.w;
      }
#pragma unroll
      for (int v = 0; v < JP / 4; ++v) {
        const float4 t = *reinterpret_cast<const float4*>(wk + jy * JP + 4 * v);
        w[4 * v] = t.x; w[4 * v + 1] = t.y; w[4 * v + 2] = t.z; w[4 * v + 3] = t.w;
      }
#pragma unroll
      for (int jx = 0; jx < J; ++jx)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = fmaf(w[jx], in[o + jx], acc[o]);
    }
  }
  const int oy = oy0 + row;
  if (oy >= Ho) return;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int ox = ox0 + gx4 + o;
    if (ox >= Wo) continue;
    float v = acc[o];
    if (quantise) v = rintf(fminf(fmaxf(v * 255.f, 0.f), 255.f)) / 255.f;
    y[((size_t)plane * Ho + oy) * Wo + ox] = v;
  }
}

template <int S, int J>
static int launch_degrade_tiled(const float* img, const float* kernels, float* out, int N, int C, int H, int W, int K,
                                int Ho, int Wo, int n_kernels, int kernel_offset, int quantise, hipStream_t st) {
  constexpr int NV = (4 + J - 1 + 3) / 4, QW = 28 + 4 * NV, QH = 16 + J - 1, JP = (J + 3) / 4 * 4;
  constexpr size_t lds = ((size_t)S * S * QH * QW + (size_t)S * S * J * JP) * sizeof(float);
  static_assert(lds <= 64 * 1024, "phase planes must fit the default dynamic LDS limit");
  const dim3 grid(ceil_div(Wo, 32), ceil_div(Ho, 16), N * C);
  hipLaunchKernelGGL((degrade_tiled_kernel<S, J>), grid, dim3(128), lds, st, img, kernels, out, N, C, H, W, K, Ho, Wo,
                     n_kernels > 1 ? 1 : 0, n_kernels, kernel_offset, quantise);
  return check_launch("degrade_tiled_kernel");
}

}  // namespace dvsr

using namespace dvsr;

extern "C" int dvsr_degrade_apply(const float* img, const float* kernels, float* out, int N, int C, int H, int W,
                                  int K, int scale, int n_kernels, int kernel_offset, int quantise,
                                  dvsr_stream_t stream) {
  DVSR_REQUIRE(img && kernels && out, DVSR_ERR_INVALID, "degrade_apply: null img/kernels/out");
  DVSR_REQUIRE(N >= 1 && C >= 1 && H >= 1 && W >= 1 && n_kernels >= 1, DVSR_ERR_INVALID,
               "degrade_apply: N=%d C=%d H=%d W=%d n_kernels=%d", N, C, H, W, n_kernels);
  DVSR_REQUIRE(K >= 1 && K <= DG_MAXK && scale >= 1 && scale <= DG_MAXS, DVSR_ERR_UNSUPPORTED,
               "degrade_apply: kernel edge %d (<= %d) / scale %d (<= %d)", K, DG_MAXK, scale, DG_MAXS);
  // ReflectionPad2d's own contract: padding must be smaller than the input
  DVSR_REQUIRE(K / 2 < H && K / 2 < W, DVSR_ERR_INVALID, "degrade_apply: reflection pad %d needs H, W > %d (got %dx%d)",
               K / 2, K / 2, H, W);
  const int pad = K / 2;
  const int Ho = (H + 2 * pad - K) / scale + 1, Wo = (W + 2 * pad - K) / scale + 1;
  if (N * C <= 65535 && !getenv("DVSR_DEGRADE_GENERIC")) {
    const int J = ceil_div(K, scale);
    hipStream_t st = (hipStream_t)stream;
#define DVSR_DG_CASE(S_, J_) \
    if (scale == S_ && J == J_) \
      return launch_degrade_tiled<S_, J_>(img, kernels, out, N, C, H, W, K, Ho, Wo, n_kernels, kernel_offset, quantise, st);
    DVSR_DG_CASE(4, 7) DVSR_DG_CASE(2, 13) DVSR_DG_CASE(2, 8) DVSR_DG_CASE(3, 9)
#undef DVSR_DG_CASE
  }
  const int P = (DG_T - 1) * scale + K;
  const size_t lds = ((size_t)P * (P + 1) + (size_t)K * K) * sizeof(float);
  const dim3 grid(ceil_div(Wo, DG_T), ceil_div(Ho, DG_T), N * C);
  DVSR_REQUIRE(grid.z <= 65535, DVSR_ERR_UNSUPPORTED, "degrade_apply: N*C = %d planes", N * C);
  hipLaunchKernelGGL(degrade_kernel, grid, dim3(256), lds, (hipStream_t)stream, img, kernels, out, N, C, H, W, K, scale,
                     Ho, Wo, n_kernels > 1 ? 1 : 0, n_kernels, kernel_offset, quantise);
  return check_launch("degrade_kernel");
}
