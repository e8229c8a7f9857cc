// Fused modulated deformable convolution (DCNv2) forward for gfx950.
//
// Reference: modulated_deform_conv_cuda_forward (deform_conv_cuda.cpp:486-564) =
// zero-fill + im2col kernel (deform_conv_cuda_kernel.cu:569-632, sampler :466-496) writing a
// [C*9, Ho*Wo] column buffer to HBM + addmm + bias.  Here the column tile never leaves the CU:
// per workgroup (8x32 output pixels x 64 output channels) and per deformable group the sampled,
// mask-weighted values of 3 taps x CPG channels are written to LDS and immediately contracted
// against the matching slice of W on v_mfma_f32_32x32x2_f32 (same operand roles as conv2d.hip:
// D rows = cout, D columns = pixels).  Offsets/masks are read once per (group, tap, pixel) and
// shared by the CPG channels of the group; the x gathers hit L1/L2 (a group's planes are
// CPG*H*W*4 bytes, e.g. 1.8 MB at 180x320).
#include "common.h"
#include "kernels.h"

namespace dvsr {

struct DcnK {
  const float* x; const float* off; const float* msk; const float* w; const float* bias; float* out;
  long long off_bstride, msk_bstride;  // elements between consecutive batch items
  int mask_logit;                      // 1: msk holds pre-sigmoid values (packed conv output)
  int N, C, H, W, Cout, Ho, Wo, stride, pad, dil, dg, act;
  int tiles_x, tiles_y, ntiles, ncb;
};

template <int CPG>
__global__ __launch_bounds__(256, 2) void mdcn_fwd_kernel(DcnK a) {
  constexpr int TP = 3, KK = 9, WROW = 65, NPX = 256;
  __shared__ __attribute__((aligned(16))) float s_col[CPG * TP * NPX];
  __shared__ __attribute__((aligned(16))) float s_w[CPG * KK * WROW];

  const int id = blockIdx.x;
  const int tile = (id / (8 * a.ncb)) * 8 + (id & 7);
  const int cb = (id >> 3) % a.ncb;
  if (tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * 8, ox0 = tx_ * 32;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const size_t HW = (size_t)a.H * a.W, P = (size_t)a.Ho * a.Wo;
  // this thread's pixel for the sampling phases
  const int py = oy0 + (tid >> 5), px = ox0 + (tid & 31);
  const bool pvalid = py < a.Ho && px < a.Wo;
  const size_t pofs = (size_t)py * a.Wo + px;
  const float* offn = a.off + (size_t)n * a.off_bstride;
  const float* mskn = a.msk + (size_t)n * a.msk_bstride;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int g = 0; g < a.dg; ++g) {
    __syncthreads();  // previous group's MFMAs are done with s_w / s_col
    for (int idx = tid; idx < 64 * CPG * KK; idx += 256) {
      const int o = idx / (CPG * KK);
      const int rem = idx - o * (CPG * KK);  // = c*KK + tap, contiguous in OIHW
      const int co = cb * 64 + o;
      float v = 0.f;
      if (co < a.Cout) v = a.w[((size_t)co * a.C + g * CPG) * KK + rem];
      s_w[rem * WROW + o] = v;
    }
    const float* xg = a.x + ((size_t)n * a.C + g * CPG) * HW;
    for (int t0 = 0; t0 < KK; t0 += TP) {
      if (t0) __syncthreads();  // MFMAs of the previous tap triple have consumed s_col
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const int tap = t0 + t;
        const int ki = tap / 3, kj = tap - ki * 3;
        float vals[CPG];
#pragma unroll
        for (int c = 0; c < CPG; ++c) vals[c] = 0.f;
        if (pvalid) {
          const float oh = offn[(size_t)(g * 2 * KK + 2 * tap) * P + pofs];
          const float ow = offn[(size_t)(g * 2 * KK + 2 * tap + 1) * P + pofs];
          float m = mskn[(size_t)(g * KK + tap) * P + pofs];
          if (a.mask_logit) m = sigmoidf_(m);
          const float h_im = (float)(py * a.stride - a.pad + ki * a.dil) + oh;
          const float w_im = (float)(px * a.stride - a.pad + kj * a.dil) + ow;
          DcnTap tp;
          if (make_tap(h_im, w_im, a.H, a.W, tp)) {
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
              const float* pl = xg + (size_t)c * HW;
              const float v1 = tp.v1 ? pl[tp.o1] : 0.f;
              const float v2 = tp.v2 ? pl[tp.o2] : 0.f;
              const float v3 = tp.v3 ? pl[tp.o3] : 0.f;
              const float v4 = tp.v4 ? pl[tp.o4] : 0.f;
              vals[c] = (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * m;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < CPG; ++c) s_col[(c * TP + t) * NPX + tid] = vals[c];
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const int tap = t0 + t;
#pragma unroll
        for (int kk = 0; kk < CPG / 2; ++kk) {
          const int c = 2 * kk + hi;
          const float a0 = s_w[(c * KK + tap) * WROW + lo];
          const float a1 = s_w[(c * KK + tap) * WROW + 32 + lo];
          const float b0 = s_col[(c * TP + t) * NPX + (2 * wave) * 32 + lo];
          const float b1 = s_col[(c * TP + t) * NPX + (2 * wave + 1) * 32 + lo];
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      }
    }
  }

  const int ox = ox0 + lo;
  if (ox >= a.Wo) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int oy = oy0 + 2 * wave + nt;
      if (oy >= a.Ho) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cb * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co >= a.Cout) continue;
        float v = acc[mt][nt][r];
        if (a.bias) v += a.bias[co];
        a.out[((size_t)n * a.Cout + co) * P + (size_t)oy * a.Wo + ox] = apply_act(v, a.act);
      }
    }
}

int mdcn_forward_run(const float* x, const float* off, long long off_bs, const float* msk,
                     long long msk_bs, int mask_logit, const float* w, const float* b, float* out,
                     int N, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad,
                     int dil, int groups, int dg, int act, hipStream_t st) {
  DVSR_REQUIRE(x && off && msk && w && out, DVSR_ERR_INVALID, "mdcn_forward: null pointer");
  DVSR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cout > 0 && dg > 0 && stride > 0 && dil > 0 &&
                   pad >= 0, DVSR_ERR_INVALID, "mdcn_forward: bad dimension");
  DVSR_REQUIRE(C % dg == 0, DVSR_ERR_INVALID, "mdcn_forward: C=%d not divisible by dg=%d", C, dg);
  DVSR_REQUIRE(kh == 3 && kw == 3, DVSR_ERR_UNSUPPORTED, "mdcn_forward: kernel %dx%d (3x3 only)", kh, kw);
  DVSR_REQUIRE(groups == 1, DVSR_ERR_UNSUPPORTED, "mdcn_forward: groups=%d (1 only)", groups);
  DVSR_REQUIRE(act >= 0 && act <= 2, DVSR_ERR_INVALID, "mdcn_forward: act=%d", act);
  DcnK k;
  k.x = x; k.off = off; k.msk = msk; k.w = w; k.bias = b; k.out = out;
  k.mask_logit = mask_logit;
  k.N = N; k.C = C; k.H = H; k.W = W; k.Cout = Cout; k.stride = stride; k.pad = pad; k.dil = dil;
  k.dg = dg; k.act = act;
  k.Ho = (H + 2 * pad - (dil * 2 + 1)) / stride + 1;
  k.Wo = (W + 2 * pad - (dil * 2 + 1)) / stride + 1;
  DVSR_REQUIRE(k.Ho > 0 && k.Wo > 0, DVSR_ERR_INVALID, "mdcn_forward: empty output");
  k.off_bstride = off_bs > 0 ? off_bs : (long long)dg * 18 * k.Ho * k.Wo;
  k.msk_bstride = msk_bs > 0 ? msk_bs : (long long)dg * 9 * k.Ho * k.Wo;
  k.tiles_x = ceil_div(k.Wo, 32);
  k.tiles_y = ceil_div(k.Ho, 8);
  k.ntiles = k.tiles_x * k.tiles_y * N;
  k.ncb = ceil_div(Cout, 64);
  const int grid = ceil_div(k.ntiles, 8) * 8 * k.ncb;
  const int cpg = C / dg;
  if (cpg == 8) hipLaunchKernelGGL(mdcn_fwd_kernel<8>, dim3(grid), dim3(256), 0, st, k);
  else if (cpg == 4) hipLaunchKernelGGL(mdcn_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, k);
  else if (cpg == 16) hipLaunchKernelGGL(mdcn_fwd_kernel<16>, dim3(grid), dim3(256), 0, st, k);
  else DVSR_REQUIRE(false, DVSR_ERR_UNSUPPORTED, "mdcn_forward: C/dg=%d (supported: 4, 8, 16)", cpg);
  return check_launch("mdcn_fwd_kernel");
}

}  // namespace dvsr

extern "C" int dvsr_mdcn_forward(const float* x, const float* offset, const float* mask,
                                 const float* w, const float* b, float* out, int N, int C, int H,
                                 int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                                 int groups, int dg, int act, dvsr_stream_t stream) {
  return dvsr::mdcn_forward_run(x, offset, 0, mask, 0, 0, w, b, out, N, C, H, W, Cout, kh, kw,
                                stride, pad, dil, groups, dg, act, (hipStream_t)stream);
}

extern "C" int dvsr_mdcn_pack_forward(const float* x, const float* om, const float* w,
                                      const float* b, float* out, int N, int C, int H, int W,
                                      int Cout, int kh, int kw, int stride, int pad, int dil,
                                      int groups, int dg, int act, dvsr_stream_t stream) {
  DVSR_REQUIRE(om, DVSR_ERR_INVALID, "mdcn_pack_forward: null om");
  DVSR_REQUIRE(kh == 3 && kw == 3 && stride > 0 && dil > 0, DVSR_ERR_UNSUPPORTED,
               "mdcn_pack_forward: 3x3 only");
  const int Ho = (H + 2 * pad - (dil * 2 + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * 2 + 1)) / stride + 1;
  const long long bs = (long long)dg * 27 * Ho * Wo;
  return dvsr::mdcn_forward_run(x, om, bs, om + (size_t)dg * 18 * Ho * Wo, bs, 1, w, b, out, N, C,
                                H, W, Cout, kh, kw, stride, pad, dil, groups, dg, act,
                                (hipStream_t)stream);
}
