// Backward of the modulated deformable convolution.
//
// Reference: modulated_deform_conv_cuda_backward (deform_conv_cuda.cpp:566-679):
//   dcol = W^T . gout (addmm, :617-620) -> col2im_coord kernel (kernel.cu:694-766: grad offset/mask)
//   -> col2im kernel (kernel.cu:634-692: grad input, atomicAdd) -> im2col again (:569-632)
//   -> grad_weight += gout . col^T, grad_bias += gout . 1 (:653-665).
// Round-1 structure (correctness first): the two contractions run on the MFMA conv kernels
// (dcol = 1x1 "dgrad" conv over the flattened [Cout][C*9] weight; dW/db = 1x1 wgrad over the
// column buffer), the two sampler kernels below are fused per (pixel, tap) over the CPG channels
// of a deformable group: one thread computes the tap geometry once and produces the offset
// gradient pair, the mask gradient and the 4*CPG input-gradient atomics.  Unlike the forward,
// the [C*9, P] column / dcol buffers DO round-trip HBM here (as in the reference); fusing them
// into LDS is the planned next step (DESIGN.md "DCN backward").
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "kernels.h"

namespace dvsr {

struct DcnB {
  const float* x; const float* off; const float* msk;
  long long off_bs, msk_bs;
  int mask_logit;
  int N, C, H, W, Ho, Wo, stride, pad, dil, dg, cpg;
};

// col[n][(g*cpg + c)*9 + tap][p] = mask * bilinear(x[n, g*cpg + c], p + tap + offset)
__global__ void mdcn_im2col_kernel(DcnB a, float* __restrict__ col) {
  const size_t P = (size_t)a.Ho * a.Wo, HW = (size_t)a.H * a.W;
  const size_t total = (size_t)a.N * a.dg * 9 * P;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % P;
    size_t t = i / P;
    const int tap = (int)(t % 9); t /= 9;
    const int g = (int)(t % a.dg);
    const int n = (int)(t / a.dg);
    const int py = (int)(p / a.Wo), px = (int)(p % a.Wo);
    const float* offn = a.off + (size_t)n * a.off_bs;
    const float oh = offn[(size_t)(g * 18 + 2 * tap) * P + p];
    const float ow = offn[(size_t)(g * 18 + 2 * tap + 1) * P + p];
    float m = a.msk[(size_t)n * a.msk_bs + (size_t)(g * 9 + tap) * P + p];
    if (a.mask_logit) m = sigmoidf_(m);
    const float h_im = (float)(py * a.stride - a.pad + (tap / 3) * a.dil) + oh;
    const float w_im = (float)(px * a.stride - a.pad + (tap % 3) * a.dil) + ow;
    DcnTap tp;
    const bool in = make_tap(h_im, w_im, a.H, a.W, tp);
    const float* xg = a.x + ((size_t)n * a.C + g * a.cpg) * HW;
    float* dst = col + ((size_t)n * a.C * 9 + (size_t)(g * a.cpg) * 9 + tap) * P + p;
    for (int c = 0; c < a.cpg; ++c) {
      float v = 0.f;
      if (in) {
        const float* pl = xg + (size_t)c * HW;
        const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
        const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
        v = (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * m;
      }
      dst[(size_t)c * 9 * P] = v;
    }
  }
}

// From dcol[n][(g*cpg + c)*9 + tap][p]: grad offset (dy, dx), grad mask (or mask logit) and the
// input-gradient scatter.  goff/gmsk are written (=), gx is accumulated with hardware fp32 atomics
// (order differs run to run exactly like the reference's atomicAdd, kernel.cu:687).
__global__ void mdcn_col2im_coord_kernel(DcnB a, const float* __restrict__ dcol, float* __restrict__ goff,
                                         long long goff_bs, float* __restrict__ gmsk, long long gmsk_bs,
                                         float* __restrict__ gx) {
  const size_t P = (size_t)a.Ho * a.Wo, HW = (size_t)a.H * a.W;
  const size_t total = (size_t)a.N * a.dg * 9 * P;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % P;
    size_t t = i / P;
    const int tap = (int)(t % 9); t /= 9;
    const int g = (int)(t % a.dg);
    const int n = (int)(t / a.dg);
    const int py = (int)(p / a.Wo), px = (int)(p % a.Wo);
    const float* offn = a.off + (size_t)n * a.off_bs;
    const float oh = offn[(size_t)(g * 18 + 2 * tap) * P + p];
    const float ow = offn[(size_t)(g * 18 + 2 * tap + 1) * P + p];
    const float mraw = a.msk[(size_t)n * a.msk_bs + (size_t)(g * 9 + tap) * P + p];
    const float m = a.mask_logit ? sigmoidf_(mraw) : mraw;
    const float h_im = (float)(py * a.stride - a.pad + (tap / 3) * a.dil) + oh;
    const float w_im = (float)(px * a.stride - a.pad + (tap % 3) * a.dil) + ow;
    DcnTap tp;
    const bool in = make_tap(h_im, w_im, a.H, a.W, tp);
    float gh = 0.f, gw = 0.f, gm = 0.f;
    if (in) {
      const float* xg = a.x + ((size_t)n * a.C + g * a.cpg) * HW;
      float* gxg = gx ? gx + ((size_t)n * a.C + g * a.cpg) * HW : nullptr;
      const float* dc_ = dcol + ((size_t)n * a.C * 9 + (size_t)(g * a.cpg) * 9 + tap) * P + p;
      const float hh = 1.f - tp.lh, hw = 1.f - tp.lw;
      for (int c = 0; c < a.cpg; ++c) {
        const float dc = dc_[(size_t)c * 9 * P];
        const float* pl = xg + (size_t)c * HW;
        const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
        const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
        gm += dc * (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4);           // kernel.cu:752
        gh += (-hw * v1 - tp.lw * v2 + hw * v3 + tp.lw * v4) * dc * m;              // :541-550
        gw += (-hh * v1 + hh * v2 - tp.lh * v3 + tp.lh * v4) * dc * m;              // :552-561
        if (gxg) {
          const float top = dc * m;                                                // :672
          float* gp = gxg + (size_t)c * HW;
          if (tp.v1) unsafeAtomicAdd(gp + tp.o1, tp.w1 * top);
          if (tp.v2) unsafeAtomicAdd(gp + tp.o2, tp.w2 * top);
          if (tp.v3) unsafeAtomicAdd(gp + tp.o3, tp.w3 * top);
          if (tp.v4) unsafeAtomicAdd(gp + tp.o4, tp.w4 * top);
        }
      }
    }
    goff[(size_t)n * goff_bs + (size_t)(g * 18 + 2 * tap) * P + p] = gh;
    goff[(size_t)n * goff_bs + (size_t)(g * 18 + 2 * tap + 1) * P + p] = gw;
    if (a.mask_logit) gm *= m * (1.f - m);
    gmsk[(size_t)n * gmsk_bs + (size_t)(g * 9 + tap) * P + p] = gm;
  }
}

size_t mdcn_backward_workspace_bytes(int N, int C, int H, int W, int Cout, int stride, int pad, int dil) {
  const int Ho = (H + 2 * pad - (dil * 2 + 1)) / stride + 1, Wo = (W + 2 * pad - (dil * 2 + 1)) / stride + 1;
  const size_t col = (size_t)N * C * 9 * Ho * Wo * sizeof(float);
  return col + conv2d_wgrad_workspace_bytes(N, C * 9, Ho, Wo, Cout, 1, 1);
}

// gout: gradient w.r.t. the PRE-activation output.  gx is accumulated into (atomics) -- zero it
// first unless other contributions are already there; goff/gmsk/gw/gb are overwritten.
int mdcn_backward_run(const float* x, const float* off, long long off_bs, const float* msk, long long msk_bs,
                      int mask_logit, const float* w, const float* gout, float* gx, float* goff,
                      long long goff_bs, float* gmsk, long long gmsk_bs, float* gw, float* gb, int N, int C,
                      int H, int W, int Cout, int stride, int pad, int dil, int dg, void* ws,
                      size_t ws_bytes, hipStream_t st) {
  DVSR_REQUIRE(x && off && msk && w && gout && goff && gmsk && ws, DVSR_ERR_INVALID,
               "mdcn_backward: null pointer");
  DVSR_REQUIRE(C % dg == 0, DVSR_ERR_INVALID, "mdcn_backward: C %% dg != 0");
  const size_t need = mdcn_backward_workspace_bytes(N, C, H, W, Cout, stride, pad, dil);
  DVSR_REQUIRE(ws_bytes >= need, DVSR_ERR_WORKSPACE, "mdcn_backward: workspace %zu < %zu", ws_bytes, need);
  DcnB a;
  a.x = x; a.off = off; a.msk = msk; a.mask_logit = mask_logit;
  a.N = N; a.C = C; a.H = H; a.W = W; a.stride = stride; a.pad = pad; a.dil = dil; a.dg = dg; a.cpg = C / dg;
  a.Ho = (H + 2 * pad - (dil * 2 + 1)) / stride + 1;
  a.Wo = (W + 2 * pad - (dil * 2 + 1)) / stride + 1;
  const size_t P = (size_t)a.Ho * a.Wo;
  a.off_bs = off_bs > 0 ? off_bs : (long long)dg * 18 * P;
  a.msk_bs = msk_bs > 0 ? msk_bs : (long long)dg * 9 * P;
  if (goff_bs <= 0) goff_bs = (long long)dg * 18 * P;
  if (gmsk_bs <= 0) gmsk_bs = (long long)dg * 9 * P;
  float* col = (float*)ws;
  void* ws2 = (char*)ws + (size_t)N * C * 9 * P * sizeof(float);
  const size_t ws2_bytes = ws_bytes - (size_t)N * C * 9 * P * sizeof(float);
  // 1) dcol[n][C*9][P] = W^T . gout  as a 1x1 conv with the transposed weight view
  dvsr_conv2d_desc g = {};
  g.x0 = gout; g.w = w; g.y = col; g.N = N; g.c0 = Cout; g.H = a.Ho; g.W = a.Wo; g.Cout = C * 9; g.ks = 1;
  g.stride = 1; g.pad = 0; g.act = ACT_NONE; g.x1_bdiv = 1;
  ConvExtra ex;
  ex.wt = 1; ex.w_ctot = C * 9; ex.w_coff = 0;
  int rc = conv2d_run(g, ex, st);
  if (rc) return rc;
  // 2) offset / mask / input gradients
  const size_t total = (size_t)N * dg * 9 * P;
  const int grid = (int)((total + 255) / 256 < 65535 * 4 ? (total + 255) / 256 : 65535 * 4);
  hipLaunchKernelGGL(mdcn_col2im_coord_kernel, dim3(grid), dim3(256), 0, st, a, col, goff, goff_bs, gmsk,
                     gmsk_bs, gx);
  rc = check_launch("mdcn_col2im_coord_kernel");
  if (rc) return rc;
  // 3) col = im2col(x) (overwrites dcol), dW = gout . col^T, db = gout . 1
  if (gw) {
    hipLaunchKernelGGL(mdcn_im2col_kernel, dim3(grid), dim3(256), 0, st, a, col);
    rc = check_launch("mdcn_im2col_kernel");
    if (rc) return rc;
    rc = conv2d_wgrad_run(col, 0, 1, gout, 0, gw, gb, N, C * 9, a.Ho, a.Wo, Cout, C * 9, 0, 1, 1, ws2,
                          ws2_bytes, st);
  }
  return rc;
}

}  // namespace dvsr

extern "C" size_t dvsr_mdcn_backward_workspace_bytes(int N, int C, int H, int W, int Cout, int kh, int kw,
                                                     int stride, int pad, int dil) {
  (void)kh; (void)kw;
  return dvsr::mdcn_backward_workspace_bytes(N, C, H, W, Cout, stride, pad, dil);
}

extern "C" int dvsr_mdcn_backward(const float* x, const float* offset, const float* mask, const float* w,
                                  const float* grad_out, float* gx, float* goffset, float* gmask, float* gw,
                                  float* gb, int N, int C, int H, int W, int Cout, int kh, int kw, int stride,
                                  int pad, int dil, int groups, int dg, void* workspace, size_t workspace_bytes,
                                  dvsr_stream_t stream) {
  DVSR_REQUIRE(kh == 3 && kw == 3 && groups == 1, DVSR_ERR_UNSUPPORTED,
               "mdcn_backward: 3x3, groups=1 only (got %dx%d, groups=%d)", kh, kw, groups);
  return dvsr::mdcn_backward_run(x, offset, 0, mask, 0, 0, w, grad_out, gx, goffset, 0, gmask, 0, gw, gb, N, C,
                                 H, W, Cout, stride, pad, dil, dg, workspace, workspace_bytes,
                                 (hipStream_t)stream);
}
