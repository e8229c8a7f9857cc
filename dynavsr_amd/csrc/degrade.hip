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
  for (int i = threadIdx.x; i < P * P; i += 256) {
    const int r = i / P, c = i - r * P;
    const int gy = reflect_index(iy0 + r, H), gx = reflect_index(ix0 + c, W);
    // rows / columns only needed by outputs beyond the image edge may reflect out of range: clamp (never used)
    const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    s_in[r * (P + 1) + c] = xp[(size_t)cy * W + cx];
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

}  // namespace dvsr

using namespace dvsr;

extern "C" int dvsr_degrade_apply(const float* img, const float* kernels, float* out, int N, int C, int H, int W,
                                  int K, int scale, int n_kernels, int kernel_offset, int quantise,
                                  dvsr_stream_t stream) {
  DVSR_REQUIRE(img && kernels && out, DVSR_ERR_INVALID, "degrade_apply: null img/kernels/out");
  DVSR_REQUIRE(N >= 1 && C >= 1 && H >= 1 && W >= 1 && n_kernels >= 1, DVSR_ERR_INVALID,
               "degrade_apply: N=%d C=%d H=%d W=%d n_kernels=%d", N, C, H, W, n_kernels);
  DVSR_REQUIRE(K >= 1 && K <= DG_MAXK && (K & 1) && scale >= 1 && scale <= DG_MAXS, DVSR_ERR_UNSUPPORTED,
               "degrade_apply: kernel edge %d (odd, <= %d) / scale %d (<= %d)", K, DG_MAXK, scale, DG_MAXS);
  // ReflectionPad2d's own contract: padding must be smaller than the input
  DVSR_REQUIRE(K / 2 < H && K / 2 < W, DVSR_ERR_INVALID, "degrade_apply: reflection pad %d needs H, W > %d (got %dx%d)",
               K / 2, K / 2, H, W);
  const int pad = K / 2;
  const int Ho = (H + 2 * pad - K) / scale + 1, Wo = (W + 2 * pad - K) / scale + 1;
  const int P = (DG_T - 1) * scale + K;
  const size_t lds = ((size_t)P * (P + 1) + (size_t)K * K) * sizeof(float);
  const dim3 grid(ceil_div(Wo, DG_T), ceil_div(Ho, DG_T), N * C);
  DVSR_REQUIRE(grid.z <= 65535, DVSR_ERR_UNSUPPORTED, "degrade_apply: N*C = %d planes", N * C);
  hipLaunchKernelGGL(degrade_kernel, grid, dim3(256), lds, (hipStream_t)stream, img, kernels, out, N, C, H, W, K, scale,
                     Ho, Wo, n_kernels > 1 ? 1 : 0, n_kernels, kernel_offset, quantise);
  return check_launch("degrade_kernel");
}
