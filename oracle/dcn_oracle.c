/*
 * oracle/dcn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of DynaVSR's modulated deformable convolution (DCNv2),
 * the one op of the EDVR hot path that the reference implements only in CUDA:
 *
 *   forward   : codes/models/archs/dcn/src/deform_conv_cuda.cpp:486-564
 *               + modulated_deformable_im2col_gpu_kernel  (deform_conv_cuda_kernel.cu:569-632)
 *               + dmcn_im2col_bilinear                    (deform_conv_cuda_kernel.cu:466-496)
 *   backward  : deform_conv_cuda.cpp:566-679
 *               + modulated_deformable_col2im_coord_gpu_kernel (kernel.cu:694-766)
 *               + modulated_deformable_col2im_gpu_kernel       (kernel.cu:634-692)
 *               + dmcn_get_gradient_weight / dmcn_get_coordinate_weight (kernel.cu:498-567)
 *
 * Parity status: the reference op has NO CPU path (deform_conv.py:109-110 raises for non-CUDA
 * tensors) and its CUDA extension cannot be built in this image (needs nvcc + removed ATen
 * APIs), so the arithmetic below is "parity unpinned" by an execution of the reference itself.
 * It is pinned instead by (i) following the kernel loops statement by statement, (ii) agreeing
 * with an independent pure-torch gather formulation + autograd (tests/test_oracle_dcn.py),
 * (iii) torch.autograd.gradcheck in fp64, (iv) the zero-offset/unit-mask == conv2d identity.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this file.
 *
 * Layouts (all contiguous):
 *   x      [N, C, H, W]
 *   offset [N, dg*2*kh*kw, Ho, Wo]   channel = g*2K + 2*k + {0: dy, 1: dx}
 *   mask   [N, dg*kh*kw,   Ho, Wo]   channel = g*K + k
 *   w      [Cout, C/groups, kh, kw]
 *   out    [N, Cout, Ho, Wo]
 * The file is compiled twice: REAL=float -> *_f32, REAL=double -> *_f64.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#error "compile with -DREAL=float -DSUFFIX=f32 or -DREAL=double -DSUFFIX=f64"
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef REAL real;

typedef struct {
  int N, C, H, W, Cout, kh, kw, stride, pad, dil, groups, dg, Ho, Wo;
} dims_t;

static int fill_dims(dims_t *d, int N, int C, int H, int W, int Cout, int kh, int kw, int stride,
                     int pad, int dil, int groups, int dg) {
  d->N = N; d->C = C; d->H = H; d->W = W; d->Cout = Cout; d->kh = kh; d->kw = kw;
  d->stride = stride; d->pad = pad; d->dil = dil; d->groups = groups; d->dg = dg;
  d->Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1; /* deform_conv_cuda.cpp:512-515 */
  d->Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  if (N <= 0 || C <= 0 || Cout <= 0 || groups <= 0 || dg <= 0) return -1;
  if (C % groups || Cout % groups || C % dg) return -1;
  if (d->Ho <= 0 || d->Wo <= 0) return -1;
  return 0;
}

/* kernel.cu:466-496: bilinear sample, each corner individually zero outside the image */
static real sample_bilinear(const real *plane, int H, int W, real h, real w) {
  int h_lo = (int)floor((double)h), w_lo = (int)floor((double)w);
  int h_hi = h_lo + 1, w_hi = w_lo + 1;
  real lh = h - (real)h_lo, lw = w - (real)w_lo;
  real hh = (real)1 - lh, hw = (real)1 - lw;
  real v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_lo >= 0 && w_lo >= 0) v1 = plane[h_lo * W + w_lo];
  if (h_lo >= 0 && w_hi <= W - 1) v2 = plane[h_lo * W + w_hi];
  if (h_hi <= H - 1 && w_lo >= 0) v3 = plane[h_hi * W + w_lo];
  if (h_hi <= H - 1 && w_hi <= W - 1) v4 = plane[h_hi * W + w_hi];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

/* kernel.cu:498-523: weight with which pixel (h, w) receives the gradient of a sample at
 * (ah, aw); later matches override earlier ones exactly as the if-chain there does. */
static real gradient_weight(real ah, real aw, int h, int w, int H, int W) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
  int h_lo = (int)floor((double)ah), w_lo = (int)floor((double)aw);
  int h_hi = h_lo + 1, w_hi = w_lo + 1;
  real wt = 0;
  if (h == h_lo && w == w_lo) wt = ((real)h + 1 - ah) * ((real)w + 1 - aw);
  if (h == h_lo && w == w_hi) wt = ((real)h + 1 - ah) * (aw + 1 - (real)w);
  if (h == h_hi && w == w_lo) wt = (ah + 1 - (real)h) * ((real)w + 1 - aw);
  if (h == h_hi && w == w_hi) wt = (ah + 1 - (real)h) * (aw + 1 - (real)w);
  return wt;
}

/* kernel.cu:525-567: d(sample)/d(coordinate); dir 0 = d/dh, dir 1 = d/dw */
static real coordinate_weight(real ah, real aw, int H, int W, const real *plane, int dir) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
  int h_lo = (int)floor((double)ah), w_lo = (int)floor((double)aw);
  int h_hi = h_lo + 1, w_hi = w_lo + 1;
  real wt = 0;
  if (dir == 0) {
    if (h_lo >= 0 && w_lo >= 0) wt += -1 * ((real)w_lo + 1 - aw) * plane[h_lo * W + w_lo];
    if (h_lo >= 0 && w_hi <= W - 1) wt += -1 * (aw - (real)w_lo) * plane[h_lo * W + w_hi];
    if (h_hi <= H - 1 && w_lo >= 0) wt += ((real)w_lo + 1 - aw) * plane[h_hi * W + w_lo];
    if (h_hi <= H - 1 && w_hi <= W - 1) wt += (aw - (real)w_lo) * plane[h_hi * W + w_hi];
  } else {
    if (h_lo >= 0 && w_lo >= 0) wt += -1 * ((real)h_lo + 1 - ah) * plane[h_lo * W + w_lo];
    if (h_lo >= 0 && w_hi <= W - 1) wt += ((real)h_lo + 1 - ah) * plane[h_lo * W + w_hi];
    if (h_hi <= H - 1 && w_lo >= 0) wt += -1 * (ah - (real)h_lo) * plane[h_hi * W + w_lo];
    if (h_hi <= H - 1 && w_hi <= W - 1) wt += (ah - (real)h_lo) * plane[h_hi * W + w_hi];
  }
  return wt;
}

/* kernel.cu:569-632 for one batch element: col[(c*K + k), p] = mask * sample */
static void im2col_one(const dims_t *d, const real *x, const real *offset, const real *mask,
                       real *col) {
  const int K = d->kh * d->kw, P = d->Ho * d->Wo, cpg = d->C / d->dg;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < d->C; ++c) {
    const int g = c / cpg;
    const real *plane = x + (size_t)c * d->H * d->W;
    const real *off_g = offset + (size_t)g * 2 * K * P;
    const real *msk_g = mask + (size_t)g * K * P;
    for (int i = 0; i < d->kh; ++i)
      for (int j = 0; j < d->kw; ++j) {
        const int k = i * d->kw + j;
        real *dst = col + ((size_t)c * K + k) * P;
        for (int ho = 0; ho < d->Ho; ++ho)
          for (int wo = 0; wo < d->Wo; ++wo) {
            const int p = ho * d->Wo + wo;
            const real oh = off_g[(size_t)(2 * k) * P + p];
            const real ow = off_g[(size_t)(2 * k + 1) * P + p];
            const real m = msk_g[(size_t)k * P + p];
            const real h_im = (real)(ho * d->stride - d->pad + i * d->dil) + oh;
            const real w_im = (real)(wo * d->stride - d->pad + j * d->dil) + ow;
            real val = 0;
            if (h_im > -1 && w_im > -1 && h_im < d->H && w_im < d->W) /* kernel.cu:617 */
              val = sample_bilinear(plane, d->H, d->W, h_im, w_im);
            dst[p] = val * m;
          }
      }
  }
}

int FN(dcn_oracle_forward)(const real *x, const real *offset, const real *mask, const real *w,
                           const real *b /* nullable */, real *out, int N, int C, int H, int W,
                           int Cout, int kh, int kw, int stride, int pad, int dil, int groups,
                           int dg) {
  dims_t d;
  if (fill_dims(&d, N, C, H, W, Cout, kh, kw, stride, pad, dil, groups, dg)) return -1;
  const int K = kh * kw, P = d.Ho * d.Wo;
  const int cg = C / groups, og = Cout / groups;
  real *col = (real *)malloc((size_t)C * K * P * sizeof(real));
  if (!col) return -2;
  for (int n = 0; n < N; ++n) {
    im2col_one(&d, x + (size_t)n * C * H * W, offset + (size_t)n * dg * 2 * K * P,
               mask + (size_t)n * dg * K * P, col);
    /* cpp:545-550  out[n, g] = W[g] (og x cg*K) . col[g] (cg*K x P), then + bias (cpp:561-563) */
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o) {
      const int g = o / og;
      real *dst = out + ((size_t)n * Cout + o) * P;
      for (int p = 0; p < P; ++p) dst[p] = 0;
      for (int r = 0; r < cg * K; ++r) {
        const real wv = w[(size_t)o * cg * K + r];
        const real *src = col + ((size_t)g * cg * K + r) * P;
        for (int p = 0; p < P; ++p) dst[p] += wv * src[p];
      }
      if (b)
        for (int p = 0; p < P; ++p) dst[p] += b[o];
    }
  }
  free(col);
  return 0;
}

/* All five gradients are ACCUMULATED into (callers pass zero-filled buffers, deform_conv.py:128-132),
 * except grad_offset / grad_mask which the reference kernel assigns (kernel.cu:759-764). */
int FN(dcn_oracle_backward)(const real *x, const real *offset, const real *mask, const real *w,
                            const real *gout, real *gx, real *goffset, real *gmask, real *gw,
                            real *gb /* nullable */, int N, int C, int H, int W, int Cout, int kh,
                            int kw, int stride, int pad, int dil, int groups, int dg) {
  dims_t d;
  if (fill_dims(&d, N, C, H, W, Cout, kh, kw, stride, pad, dil, groups, dg)) return -1;
  const int K = kh * kw, P = d.Ho * d.Wo;
  const int cg = C / groups, og = Cout / groups, cpg = C / dg;
  real *col = (real *)malloc((size_t)C * K * P * sizeof(real));
  if (!col) return -2;
  for (int n = 0; n < N; ++n) {
    const real *xn = x + (size_t)n * C * H * W;
    const real *offn = offset + (size_t)n * dg * 2 * K * P;
    const real *mskn = mask + (size_t)n * dg * K * P;
    const real *gon = gout + (size_t)n * Cout * P;
    /* cpp:616-621  dcol = W^T . gout */
#pragma omp parallel for schedule(static)
    for (int r = 0; r < C * K; ++r) {
      const int c = r / K, g = c / cg, rl = (c - g * cg) * K + (r % K);
      real *dst = col + (size_t)r * P;
      for (int p = 0; p < P; ++p) dst[p] = 0;
      for (int ol = 0; ol < og; ++ol) {
        const int o = g * og + ol;
        const real wv = w[(size_t)o * cg * K + rl];
        const real *src = gon + (size_t)o * P;
        for (int p = 0; p < P; ++p) dst[p] += wv * src[p];
      }
    }
    /* kernel.cu:694-766  grad wrt offset and mask */
#pragma omp parallel for schedule(static)
    for (int oc = 0; oc < dg * 2 * K; ++oc) {
      const int g = oc / (2 * K), offc = oc - g * 2 * K, k = offc / 2, dir = offc % 2;
      const int i = k / kw, j = k % kw;
      for (int ho = 0; ho < d.Ho; ++ho)
        for (int wo = 0; wo < d.Wo; ++wo) {
          const int p = ho * d.Wo + wo;
          const real oh = offn[((size_t)g * 2 * K + 2 * k) * P + p];
          const real ow = offn[((size_t)g * 2 * K + 2 * k + 1) * P + p];
          const real m = mskn[((size_t)g * K + k) * P + p];
          real val = 0, mval = 0;
          for (int cl = 0; cl < cpg; ++cl) {
            const int c = g * cpg + cl;
            const real *plane = xn + (size_t)c * H * W;
            const real dc = col[((size_t)c * K + k) * P + p];
            real inv_h = (real)(ho * stride - pad + i * dil) + oh;
            real inv_w = (real)(wo * stride - pad + j * dil) + ow;
            if (inv_h <= -1 || inv_w <= -1 || inv_h >= H || inv_w >= W) {
              inv_h = inv_w = -2;
            } else {
              mval += dc * sample_bilinear(plane, H, W, inv_h, inv_w);
            }
            val += coordinate_weight(inv_h, inv_w, H, W, plane, dir) * dc * m;
          }
          goffset[((size_t)n * dg * 2 * K + oc) * P + p] = val;
          if (dir == 0) gmask[((size_t)n * dg * K + g * K + k) * P + p] = mval;
        }
    }
    /* kernel.cu:634-692  grad wrt input (sequential, hence deterministic, adds) */
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
      const int g = c / cpg;
      real *gplane = gx + ((size_t)n * C + c) * H * W;
      for (int k = 0; k < K; ++k) {
        const int i = k / kw, j = k % kw;
        for (int ho = 0; ho < d.Ho; ++ho)
          for (int wo = 0; wo < d.Wo; ++wo) {
            const int p = ho * d.Wo + wo;
            const real oh = offn[((size_t)g * 2 * K + 2 * k) * P + p];
            const real ow = offn[((size_t)g * 2 * K + 2 * k + 1) * P + p];
            const real m = mskn[((size_t)g * K + k) * P + p];
            const real ih = (real)(ho * stride - pad + i * dil) + oh;
            const real iw = (real)(wo * stride - pad + j * dil) + ow;
            const real top = col[((size_t)c * K + k) * P + p] * m;
            const int ch = (int)ih, cw = (int)iw; /* truncation toward zero, kernel.cu:673-674 */
            for (int dy = -2; dy <= 2; ++dy)
              for (int dx = -2; dx <= 2; ++dx) {
                const int yy = ch + dy, xx = cw + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && fabs((double)(ih - (real)yy)) < 1 &&
                    fabs((double)(iw - (real)xx)) < 1)
                  gplane[yy * W + xx] += gradient_weight(ih, iw, yy, xx, H, W) * top;
              }
          }
      }
    }
    /* cpp:643-665  recompute col, grad_weight += gout . col^T, grad_bias += gout . 1 */
    im2col_one(&d, xn, offn, mskn, col);
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o) {
      const int g = o / og;
      const real *src = gon + (size_t)o * P;
      for (int r = 0; r < cg * K; ++r) {
        const real *cr = col + ((size_t)g * cg * K + r) * P;
        real acc = 0;
        for (int p = 0; p < P; ++p) acc += src[p] * cr[p];
        gw[(size_t)o * cg * K + r] += acc;
      }
      if (gb) {
        real acc = 0;
        for (int p = 0; p < P; ++p) acc += src[p];
        gb[o] += acc;
      }
    }
  }
  free(col);
  return 0;
}
