// HBM-bound streaming kernels of the EDVR hot path: bilinear up-sampling, the 3x3/s2 pooling
// pair and the TSA gate / blend (EDVR_arch.py:107-120,166-202,311).  One thread per output
// element (or per pixel for the channel reductions), consecutive lanes = consecutive pixels, so
// every load/store is a coalesced 256-byte wave access; grids are capped and grid-strided.
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dvsr {

static inline int stream_grid(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  const size_t cap = 256 * 16;  // 256 CUs x 16 resident blocks
  return (int)(g < cap ? (g ? g : 1) : cap);
}

// ---- F.interpolate(x, scale_factor=S, mode='bilinear', align_corners=False) * mul ------------
// Source index rule of ATen's area_pixel_compute_source_index: src = max(0, (dst+0.5)/S - 0.5),
// i0 = floor(src), i1 = min(i0+1, in-1), lambda = src - i0.
__device__ __forceinline__ void src_index(int dst, float inv_scale, int in_size, int& i0, int& i1,
                                          float& l1) {
  float s = ((float)dst + 0.5f) * inv_scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ void upsample_bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                             size_t planes, int H, int W, int S, float mul) {
  const int Ho = H * S, Wo = W * S;
  const size_t total = planes * Ho * Wo;
  const float inv = 1.f / (float)S;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oy = (int)(t % Ho);
    const size_t p = t / Ho;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(oy, inv, H, y0, y1, ly);
    src_index(ox, inv, W, x0, x1, lx);
    const float* pl = x + p * (size_t)H * W;
    const float v = (1.f - ly) * ((1.f - lx) * pl[y0 * W + x0] + lx * pl[y0 * W + x1]) +
                    ly * ((1.f - lx) * pl[y1 * W + x0] + lx * pl[y1 * W + x1]);
    y[i] = v * mul;
  }
}

// Same op, 4 consecutive outputs per thread (one 16-byte store per lane; requires S*W % 4 == 0 and a
// 16-byte aligned y).  The scalar kernel above reached 1.3-1.8 TB/s in the r01 profile, dominated by
// 64-bit index arithmetic and 4-byte stores.
__global__ void upsample_bilinear_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                              unsigned planes, int H, int W, int S, float mul) {
  const int Ho = H * S, Wo = W * S, Wq = Wo >> 2;
  const unsigned total = planes * (unsigned)Ho * (unsigned)Wq;
  const float inv = 1.f / (float)S;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int oq = (int)(i % (unsigned)Wq);
    const unsigned t = i / (unsigned)Wq;
    const int oy = (int)(t % (unsigned)Ho);
    const unsigned p = t / (unsigned)Ho;
    int y0, y1;
    float ly;
    src_index(oy, inv, H, y0, y1, ly);
    const float* r0 = x + (size_t)p * H * W + (size_t)y0 * W;
    const float* r1 = x + (size_t)p * H * W + (size_t)y1 * W;
    f32x4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      float lx;
      src_index(4 * oq + j, inv, W, x0, x1, lx);
      out[j] = ((1.f - ly) * ((1.f - lx) * r0[x0] + lx * r0[x1]) + ly * ((1.f - lx) * r1[x0] + lx * r1[x1])) * mul;
    }
    *reinterpret_cast<f32x4*>(y + ((size_t)p * Ho + oy) * Wo + 4 * oq) = out;
  }
}

// S = 2 (every PCD / TSA up-sampling): one thread = 4 input columns of one input row -> the 2 x 8 output
// block they generate.  Three input rows x (one 16-byte load + the two neighbour columns) in, four
// 16-byte stores out; no integer division beyond thread -> (row, column group), no gathers.  Same
// arithmetic as src_index() above: even outputs blend (i-1, i) with lambda 0.75 (index 0: exactly in[0]),
// odd outputs blend (i, min(i+1, n-1)) with lambda 0.25.
__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned planes, int H,
                                      int W, float mul) {
  const int Wq = W >> 2, Wo = 2 * W;
  const unsigned total = planes * (unsigned)H * (unsigned)Wq;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int xq = (int)(i % (unsigned)Wq);
    const unsigned t = i / (unsigned)Wq;
    const int iy = (int)(t % (unsigned)H);
    const unsigned p = t / (unsigned)H;
    const float* pl = x + (size_t)p * H * W;
    const int x0 = 4 * xq;
    const int rows[3] = {iy > 0 ? iy - 1 : 0, iy, iy < H - 1 ? iy + 1 : H - 1};
    float h[3][8];  // horizontally interpolated rows: output columns 2*x0 .. 2*x0 + 7
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float* row = pl + (size_t)rows[r] * W;
      const f32x4 c = *reinterpret_cast<const f32x4*>(row + x0);
      const float in[6] = {row[x0 > 0 ? x0 - 1 : 0], c[0], c[1], c[2], c[3], row[x0 + 4 < W ? x0 + 4 : W - 1]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // even output 2(x0+j): (in[j-1], in[j]) with lambda 0.75, except output 0 = in[0]
        h[r][2 * j] = (x0 + j > 0) ? 0.25f * in[j] + 0.75f * in[j + 1] : in[j + 1];
        // odd output 2(x0+j)+1: (in[j], in[min(j+1)]) with lambda 0.25
        h[r][2 * j + 1] = 0.75f * in[j + 1] + 0.25f * in[j + 2];
      }
    }
    float* o0 = y + ((size_t)p * 2 * H + 2 * iy) * Wo + 2 * x0;
    f32x4 a0, a1, b0, b1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float ev = iy > 0 ? 0.25f * h[0][j] + 0.75f * h[1][j] : h[1][j];
      const float od = 0.75f * h[1][j] + 0.25f * h[2][j];
      if (j < 4) { a0[j] = ev * mul; b0[j] = od * mul; } else { a1[j - 4] = ev * mul; b1[j - 4] = od * mul; }
    }
    *reinterpret_cast<f32x4*>(o0) = a0;
    *reinterpret_cast<f32x4*>(o0 + 4) = a1;
    *reinterpret_cast<f32x4*>(o0 + Wo) = b0;
    *reinterpret_cast<f32x4*>(o0 + Wo + 4) = b1;
  }
}

// The same arithmetic with one thread = TWO input columns of one input row -> a 2 x 4 output block: the lanes of a wave store
// 64 consecutive 16-byte groups of an output row (one whole 1 KB segment per store instruction).  With four input columns per
// thread every store instruction wrote alternate 16-byte halves of its 32-byte lane stride: two instructions to complete
// each 64-byte line, 3.2 TB/s on the 92 MB launches of the PCD pyramid (round 6).
__global__ void upsample2x_fwd2_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned planes, int H,
                                       int W, float mul) {
  const int Wh = W >> 1, Wo = 2 * W;
  const unsigned total = planes * (unsigned)H * (unsigned)Wh;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int xh = (int)(i % (unsigned)Wh);
    const unsigned t = i / (unsigned)Wh;
    const int iy = (int)(t % (unsigned)H);
    const unsigned p = t / (unsigned)H;
    const float* pl = x + (size_t)p * H * W;
    const int x0 = 2 * xh;
    const int rows[3] = {iy > 0 ? iy - 1 : 0, iy, iy < H - 1 ? iy + 1 : H - 1};
    float h[3][4];  // horizontally interpolated rows: output columns 2*x0 .. 2*x0 + 3
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float* row = pl + (size_t)rows[r] * W;
      typedef float f32x2_ __attribute__((ext_vector_type(2)));
      const f32x2_ c = *reinterpret_cast<const f32x2_*>(row + x0);
      const float in[4] = {row[x0 > 0 ? x0 - 1 : 0], c[0], c[1], row[x0 + 2 < W ? x0 + 2 : W - 1]};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        h[r][2 * j] = (x0 + j > 0) ? 0.25f * in[j] + 0.75f * in[j + 1] : in[j + 1];
        h[r][2 * j + 1] = 0.75f * in[j + 1] + 0.25f * in[j + 2];
      }
    }
    float* o0 = y + ((size_t)p * 2 * H + 2 * iy) * Wo + 2 * x0;
    f32x4 a0, b0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ev = iy > 0 ? 0.25f * h[0][j] + 0.75f * h[1][j] : h[1][j];
      const float od = 0.75f * h[1][j] + 0.25f * h[2][j];
      a0[j] = ev * mul; b0[j] = od * mul;
    }
    *reinterpret_cast<f32x4*>(o0) = a0;
    *reinterpret_cast<f32x4*>(o0 + Wo) = b0;
  }
}

// Backward of the above: each input pixel gathers from the <= (S+1)^2 outputs that read it
// (deterministic, no atomics).  gx = sum_o w(o -> i) * gy[o] * mul.
__global__ void upsample_bilinear_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                             size_t planes, int H, int W, int S, float mul,
                                             int accumulate) {
  const int Ho = H * S, Wo = W * S;
  const size_t total = planes * H * W;
  const float inv = 1.f / (float)S;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    const size_t t = i / W;
    const int iy = (int)(t % H);
    const size_t p = t / H;
    const float* g = gy + p * (size_t)Ho * Wo;
    float acc = 0.f;
    // outputs whose source interval can touch row iy: oy in [(iy-1)*S, (iy+1)*S + S)
    const int oy_lo = max(0, (iy - 1) * S), oy_hi = min(Ho, (iy + 2) * S);
    const int ox_lo = max(0, (ix - 1) * S), ox_hi = min(Wo, (ix + 2) * S);
    for (int oy = oy_lo; oy < oy_hi; ++oy) {
      int y0, y1;
      float ly;
      src_index(oy, inv, H, y0, y1, ly);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox < ox_hi; ++ox) {
        int x0, x1;
        float lx;
        src_index(ox, inv, W, x0, x1, lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx != 0.f) acc += wy * wx * g[(size_t)oy * Wo + ox];
      }
    }
    acc *= mul;
    gx[i] = accumulate ? gx[i] + acc : acc;
  }
}

// ---- MaxPool2d(3,2,1) and AvgPool2d(3,2,1) of the same input in one pass ---------------------
// (EDVR_arch.py:149-150,184-185,189-190; max pads with -inf, avg divides by 9 always =
// count_include_pad=True).  Writes the two results to separate tensors: the following conv
// consumes them as its two concatenated inputs.
__global__ void pool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ ymax,
                                   float* __restrict__ yavg, size_t planes, int H, int W, int Ho,
                                   int Wo) {
  const size_t total = planes * Ho * Wo;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oy = (int)(t % Ho);
    const size_t p = t / Ho;
    const float* pl = x + p * (size_t)H * W;
    float mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = 2 * oy + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = 2 * ox + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const float v = pl[yy * W + xx];
        mx = fmaxf(mx, v);
        sum += v;
      }
    }
    ymax[i] = mx;
    yavg[i] = sum * (1.f / 9.f);
  }
}

// Backward: gx[i] = sum over the <= 4 windows containing i of (gavg/9 + gmax * [i is the
// window's arg-max]).  The arg-max is the FIRST maximal element in row-major window order, which
// is what ATen's max_pool2d_with_indices records.
__global__ void pool3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gmax,
                                   const float* __restrict__ gavg, float* __restrict__ gx,
                                   size_t planes, int H, int W, int Ho, int Wo, int accumulate) {
  const size_t total = planes * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    const size_t t = i / W;
    const int iy = (int)(t % H);
    const size_t p = t / H;
    const float* pl = x + p * (size_t)H * W;
    float acc = 0.f;
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy >= Ho) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox >= Wo) continue;
        const size_t o = (p * Ho + oy) * (size_t)Wo + ox;
        acc += gavg[o] * (1.f / 9.f);
        // locate the window's first maximum
        float mx = -INFINITY;
        int ay = -1, ax = -1;
        for (int dy = -1; dy <= 1; ++dy) {
          const int yy = 2 * oy + dy;
          if ((unsigned)yy >= (unsigned)H) continue;
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = 2 * ox + dx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const float v = pl[yy * W + xx];
            if (v > mx) { mx = v; ay = yy; ax = xx; }
          }
        }
        if (ay == iy && ax == ix) acc += gmax[o];
      }
    }
    gx[i] = accumulate ? gx[i] + acc : acc;
  }
}

// ---- TSA temporal gate (EDVR_arch.py:169-176) ------------------------------------------------
// cor[b,n,p] = sigmoid(sum_c emb[b,n,c,p] * emb_ref[b,c,p]);  gated[b,n,c,p] = aligned * cor.
// lane = pixel, so the channel reduction is a private register loop (no cross-lane traffic).
__global__ void tsa_gate_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ emb_ref,
                                    const float* __restrict__ aligned, float* __restrict__ cor,
                                    float* __restrict__ gated, int B, int N, int C, size_t HW) {
  const size_t total = (size_t)B * N * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    const size_t bn = i / HW;
    const size_t b = bn / N;
    const float* e = emb + bn * C * HW + p;
    const float* r = emb_ref + b * C * HW + p;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += e[c * HW] * r[c * HW];
    const float s = sigmoidf_(acc);
    cor[i] = s;
    const float* al = aligned + bn * C * HW + p;
    float* gt = gated + bn * C * HW + p;
    for (int c = 0; c < C; ++c) gt[c * HW] = al[c * HW] * s;
  }
}

// V consecutive pixels per thread (8- / 16-byte loads and stores): same arithmetic, fewer and wider
// memory instructions.  Requires HW % V == 0 and V*4-byte aligned planes.
template <int V>
__global__ void tsa_gate_fwd_vec_kernel(const float* __restrict__ emb, const float* __restrict__ emb_ref,
                                        const float* __restrict__ aligned, float* __restrict__ cor,
                                        float* __restrict__ gated, int B, int N, int C, size_t HW) {
  typedef float vec __attribute__((ext_vector_type(V)));
  const size_t HWv = HW / V, total = (size_t)B * N * HWv;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = (i % HWv) * V;
    const size_t bn = i / HWv;
    const size_t b = bn / N;
    const float* e = emb + bn * C * HW + p;
    const float* r = emb_ref + b * C * HW + p;
    vec acc = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c)
      acc += *reinterpret_cast<const vec*>(e + c * HW) * *reinterpret_cast<const vec*>(r + c * HW);
    vec s;
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = sigmoidf_(acc[j]);
    *reinterpret_cast<vec*>(cor + bn * HW + p) = s;
    const float* al = aligned + bn * C * HW + p;
    float* gt = gated + bn * C * HW + p;
#pragma unroll 8
    for (int c = 0; c < C; ++c) *reinterpret_cast<vec*>(gt + c * HW) = *reinterpret_cast<const vec*>(al + c * HW) * s;
  }
}

// Backward: g_aligned = g_gated * cor (+= into ga);  g_cor = sum_c g_gated * aligned;
// g_dot = g_cor * cor * (1 - cor);  g_emb[b,n,c,p] = g_dot * emb_ref;  g_emb_ref accumulates
// over n -> the thread loops the N frames of its pixel itself (deterministic).
__global__ void tsa_gate_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ emb_ref,
                                    const float* __restrict__ aligned, const float* __restrict__ cor,
                                    const float* __restrict__ g_gated, float* __restrict__ g_emb,
                                    float* __restrict__ g_emb_ref, float* __restrict__ g_aligned,
                                    int B, int N, int C, size_t HW) {
  const size_t total = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    const size_t b = i / HW;
    const float* r = emb_ref + b * C * HW + p;
    float* gr = g_emb_ref + b * C * HW + p;
    for (int c = 0; c < C; ++c) gr[c * HW] = 0.f;
    for (int n = 0; n < N; ++n) {
      const size_t bn = b * N + n;
      const float s = cor[bn * HW + p];
      const float* gg = g_gated + bn * C * HW + p;
      const float* al = aligned + bn * C * HW + p;
      float* ga = g_aligned + bn * C * HW + p;
      float gcor = 0.f;
      for (int c = 0; c < C; ++c) {
        const float g = gg[c * HW];
        gcor += g * al[c * HW];
        ga[c * HW] = g * s;
      }
      const float gdot = gcor * s * (1.f - s);
      const float* e = emb + bn * C * HW + p;
      float* ge = g_emb + bn * C * HW + p;
      for (int c = 0; c < C; ++c) {
        ge[c * HW] = gdot * r[c * HW];
        gr[c * HW] += gdot * e[c * HW];
      }
    }
  }
}

// Two-pass variant with N x more threads (the one above runs B*HW threads, 3520 on the inner-step clip:
// 190 us).  Pass 1, thread = (b, n, pixel): g_aligned, g_emb and the per-frame scalar g_dot (kept in
// `gdot`, [B][N][HW]).  Pass 2, thread = (b, c, pixel): g_emb_ref = sum_n g_dot * emb, same order of
// summation over n as the single-pass kernel.
__global__ void tsa_gate_bwd1_kernel(const float* __restrict__ emb_ref, const float* __restrict__ aligned,
                                     const float* __restrict__ cor, const float* __restrict__ g_gated,
                                     float* __restrict__ g_emb, float* __restrict__ g_aligned,
                                     float* __restrict__ gdot, int B, int N, int C, size_t HW) {
  const size_t total = (size_t)B * N * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW, bn = i / HW, b = bn / N;
    const float s = cor[i];
    const float* gg = g_gated + bn * C * HW + p;
    const float* al = aligned + bn * C * HW + p;
    float* ga = g_aligned + bn * C * HW + p;
    float gcor = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
      const float g = gg[c * HW];
      gcor += g * al[c * HW];
      ga[c * HW] = g * s;
    }
    const float gd = gcor * s * (1.f - s);
    gdot[i] = gd;
    const float* r = emb_ref + b * C * HW + p;
    float* ge = g_emb + bn * C * HW + p;
#pragma unroll 8
    for (int c = 0; c < C; ++c) ge[c * HW] = gd * r[c * HW];
  }
}
__global__ void tsa_gate_bwd2_kernel(const float* __restrict__ emb, const float* __restrict__ gdot,
                                     float* __restrict__ g_emb_ref, int B, int N, int C, size_t HW) {
  const size_t total = (size_t)B * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW, bc = i / HW, c = bc % C, b = bc / C;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += gdot[(b * N + n) * HW + p] * emb[((b * N + n) * C + c) * HW + p];
    g_emb_ref[i] = s;
  }
}

// ---- TSA blend: out = fea * sigmoid(att) * 2 + att_add (EDVR_arch.py:200-202) ---------------
__global__ void tsa_blend_fwd_kernel(const float* __restrict__ fea, const float* __restrict__ att,
                                     const float* __restrict__ add, float* __restrict__ out,
                                     size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = fea[i] * sigmoidf_(att[i]) * 2.f + add[i];
}

// 16-byte variants of the pure elementwise kernels (n % 4 == 0, 16-byte aligned pointers)
__global__ void tsa_blend_fwd4_kernel(const f32x4* __restrict__ fea, const f32x4* __restrict__ att,
                                      const f32x4* __restrict__ add, f32x4* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 f = fea[i], a = att[i], d = add[i];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f[j] * sigmoidf_(a[j]) * 2.f + d[j];
    out[i] = o;
  }
}
__global__ void add_inplace4_kernel(f32x4* __restrict__ dst, const f32x4* __restrict__ src, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    dst[i] += src[i];
}
__global__ void act_bwd_inplace4_kernel(f32x4* __restrict__ g, const f32x4* __restrict__ y, size_t n4, int act) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = g[i];
    const f32x4 yy = y[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= act_grad_from_out(yy[j], act);
    g[i] = v;
  }
}

// g_fea = g*2*s; g_att = g*fea*2*s*(1-s) (added to g_att_io, which already holds the gradient
// reaching att through the att_add branch); g_add = g is aliased by the caller.
__global__ void tsa_blend_bwd_kernel(const float* __restrict__ fea, const float* __restrict__ att,
                                     const float* __restrict__ g, float* __restrict__ g_fea,
                                     float* __restrict__ g_att_io, size_t n, int accumulate) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float s = sigmoidf_(att[i]);
    const float gi = g[i];
    g_fea[i] = gi * 2.f * s;
    const float ga = gi * fea[i] * 2.f * s * (1.f - s);
    g_att_io[i] = accumulate ? g_att_io[i] + ga : ga;
  }
}

// ---- generic helpers -------------------------------------------------------------------------
__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    dst[i] += src[i];
}

// g *= act'(y) where y is the saved post-activation output.
__global__ void act_bwd_inplace_kernel(float* __restrict__ g, const float* __restrict__ y, size_t n,
                                       int act) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    g[i] *= act_grad_from_out(y[i], act);
}

// dst[b*dst_bs + e] (+)= sum_{k<cnt} src[(b*cnt + k)*per + e]: gradient of a tensor that entered a
// conv as a broadcast / strided view (the reference frame shared by the N frames of a clip).
__global__ void reduce_frames_kernel(float* __restrict__ dst, long long dst_bs, const float* __restrict__ src,
                                     int B, int cnt, size_t per, int accumulate) {
  const size_t total = (size_t)B * per;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i % per, b = i / per;
    float s = 0.f;
    for (int k = 0; k < cnt; ++k) s += src[(b * cnt + k) * per + e];
    float* d = dst + b * dst_bs + e;
    *d = accumulate ? *d + s : s;
  }
}

// ---- Charbonnier loss: mean(sqrt((x-y)^2 + eps)) (models/loss.py:26-30), two-stage deterministic
// reduction: CHARB_BLOCKS per-block partial sums, then one block folds them in a fixed order.
constexpr int CHARB_BLOCKS = 1024;
// (blockIdx.y = group of a per-group loss: n elements and one row of partial sums each; 1 row for the scalar loss)
__global__ void charbonnier_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                           float* __restrict__ partial, size_t n, float eps) {
  __shared__ float red[256];
  x += blockIdx.y * n; y += blockIdx.y * n; partial += blockIdx.y * (size_t)CHARB_BLOCKS;
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = x[i] - y[i];
    s += sqrtf(d * d + eps);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void charbonnier_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nb,
                                         float inv_n) {
  __shared__ float red[256];
  partial += blockIdx.x * (size_t)CHARB_BLOCKS; out += blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * inv_n;
}
// gx = (*gscale / n) * d / sqrt(d^2 + eps)   (gradient w.r.t. x; the one w.r.t. y is its negative)
__global__ void charbonnier_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                       const float* __restrict__ gscale, float* __restrict__ gx, size_t n,
                                       float eps, float inv_n) {
  const float g = gscale[blockIdx.y] * inv_n;
  x += blockIdx.y * n; y += blockIdx.y * n; gx += blockIdx.y * n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = x[i] - y[i];
    gx[i] = g * d / sqrtf(d * d + eps);
  }
}

// Loss tail of the inner MAML step (test_dynavsr.py:264-274): out = base + weight * mean|x - y|.  Same fixed-order
// two-stage reduction as the Charbonnier loss; `base` is the pixel loss that is already on the device.
__global__ void l1_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ partial,
                                  size_t n) {
  __shared__ float red[256];
  x += blockIdx.y * n; y += blockIdx.y * n; partial += blockIdx.y * (size_t)CHARB_BLOCKS;
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += fabsf(x[i] - y[i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void l1_final_kernel(const float* __restrict__ partial, const float* __restrict__ base, float* __restrict__ out,
                                int nb, float scale) {
  __shared__ float red[256];
  partial += blockIdx.x * (size_t)CHARB_BLOCKS;
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (base ? base[blockIdx.x] : 0.f) + red[0] * scale;
}
// gx = gscale * scale * sign(x - y)   (torch's l1_loss backward: sign(0) = 0)
__global__ void l1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gscale,
                              float* __restrict__ gx, size_t n, float scale) {
  const float g = gscale[blockIdx.y] * scale;
  x += blockIdx.y * n; y += blockIdx.y * n; gx += blockIdx.y * n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = x[i] - y[i];
    gx[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
  }
}

#define LAUNCH(kern, n, st, ...) \
  hipLaunchKernelGGL(kern, dim3(stream_grid(n)), dim3(256), 0, st, __VA_ARGS__)

int upsample_bilinear_fwd(const float* x, float* y, size_t planes, int H, int W, int S, float mul,
                          hipStream_t st) {
  DVSR_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && S >= 1, DVSR_ERR_INVALID,
               "upsample_bilinear_fwd: bad argument");
  const size_t nout = planes * H * W * S * S;
  if (S == 2 && W % 4 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)x & 15) == 0 && nout < (1ull << 32)) {
    // DVSR_UP2=4: four input columns per thread (the round-1 kernel; A/B aid, read once per process)
    static const bool four = [] { const char* v = getenv("DVSR_UP2"); return v && v[0] == '4'; }();
    if (!four) {
      LAUNCH(upsample2x_fwd2_kernel, planes * H * (W / 2), st, x, y, (unsigned)planes, H, W, mul);
      return check_launch("upsample2x_fwd2_kernel");
    }
    LAUNCH(upsample2x_fwd_kernel, planes * H * (W / 4), st, x, y, (unsigned)planes, H, W, mul);
    return check_launch("upsample2x_fwd_kernel");
  }
  if ((W * S) % 4 == 0 && ((uintptr_t)y & 15) == 0 && nout < (1ull << 32)) {
    LAUNCH(upsample_bilinear_fwd4_kernel, nout / 4, st, x, y, (unsigned)planes, H, W, S, mul);
    return check_launch("upsample_bilinear_fwd4_kernel");
  }
  LAUNCH(upsample_bilinear_fwd_kernel, nout, st, x, y, planes, H, W, S, mul);
  return check_launch("upsample_bilinear_fwd_kernel");
}
// S = 2 backward: input pixel (iy, ix) is read by output rows 2iy-1 .. 2iy+2 with the fixed weights
// (0.25, 0.75, 0.75, 0.25) -- at the borders output 0 reads in[0] with weight 1 and output 2H-1 reads
// in[H-1] with 0.75 + 0.25 -- and likewise along x: a 4x4 gather with separable constant weights instead of
// the generic kernel's search over candidate outputs.
__device__ __forceinline__ void up2_bwd_weights(int i, int n, float (&w)[4]) {
  w[0] = i > 0 ? 0.25f : 0.f;                  // output 2i-1 (odd, of input i-1): lambda 0.25 on i
  w[1] = i > 0 ? 0.75f : 1.f;                  // output 2i (even): 0.75 on i, or exactly in[0]
  w[2] = i < n - 1 ? 0.75f : 1.f;              // output 2i+1 (odd): 0.75 on i (+0.25 when i+1 clamps to i)
  w[3] = i < n - 1 ? 0.25f : 0.f;              // output 2i+2 (even, of input i+1): 0.25 on i
}
__global__ void upsample2x_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, unsigned planes, int H,
                                      int W, float mul, int accumulate) {
  const unsigned total = planes * (unsigned)H * (unsigned)W;
  const int Wo = 2 * W;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ix = (int)(i % (unsigned)W);
    const unsigned t = i / (unsigned)W;
    const int iy = (int)(t % (unsigned)H);
    const unsigned p = t / (unsigned)H;
    float wy[4], wx[4];
    up2_bwd_weights(iy, H, wy);
    up2_bwd_weights(ix, W, wx);
    const float* g = gy + (size_t)p * 4 * H * W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oy = 2 * iy - 1 + a;
      if (wy[a] == 0.f) continue;
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (wx[b] != 0.f) row += wx[b] * g[(size_t)oy * Wo + 2 * ix - 1 + b];
      acc += wy[a] * row;
    }
    acc *= mul;
    gx[i] = accumulate ? gx[i] + acc : acc;
  }
}

int upsample_bilinear_bwd(const float* gy, float* gx, size_t planes, int H, int W, int S, float mul,
                          int accumulate, hipStream_t st) {
  DVSR_REQUIRE(gy && gx && planes > 0 && H > 0 && W > 0 && S >= 1, DVSR_ERR_INVALID,
               "upsample_bilinear_bwd: bad argument");
  if (S == 2 && H >= 2 && W >= 2 && planes * H * W < (1ull << 32)) {
    LAUNCH(upsample2x_bwd_kernel, planes * H * W, st, gy, gx, (unsigned)planes, H, W, mul, accumulate);
    return check_launch("upsample2x_bwd_kernel");
  }
  LAUNCH(upsample_bilinear_bwd_kernel, planes * H * W, st, gy, gx, planes, H, W, S, mul, accumulate);
  return check_launch("upsample_bilinear_bwd_kernel");
}
int pool3s2_fwd(const float* x, float* ymax, float* yavg, size_t planes, int H, int W, hipStream_t st) {
  DVSR_REQUIRE(x && ymax && yavg && planes > 0 && H > 0 && W > 0, DVSR_ERR_INVALID,
               "pool3s2_fwd: bad argument");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  LAUNCH(pool3s2_fwd_kernel, planes * Ho * Wo, st, x, ymax, yavg, planes, H, W, Ho, Wo);
  return check_launch("pool3s2_fwd_kernel");
}
int pool3s2_bwd(const float* x, const float* gmax, const float* gavg, float* gx, size_t planes,
                int H, int W, int accumulate, hipStream_t st) {
  DVSR_REQUIRE(x && gmax && gavg && gx && planes > 0, DVSR_ERR_INVALID, "pool3s2_bwd: bad argument");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  LAUNCH(pool3s2_bwd_kernel, planes * H * W, st, x, gmax, gavg, gx, planes, H, W, Ho, Wo, accumulate);
  return check_launch("pool3s2_bwd_kernel");
}
int tsa_gate_fwd(const float* emb, const float* emb_ref, const float* aligned, float* cor,
                 float* gated, int B, int N, int C, size_t HW, hipStream_t st) {
  DVSR_REQUIRE(emb && emb_ref && aligned && cor && gated && B > 0 && N > 0 && C > 0 && HW > 0,
               DVSR_ERR_INVALID, "tsa_gate_fwd: bad argument");
  const bool al16 = (((uintptr_t)emb | (uintptr_t)emb_ref | (uintptr_t)aligned | (uintptr_t)cor | (uintptr_t)gated) & 15) == 0;
  static int vsel = -1;
  if (vsel < 0) { const char* v = getenv("DVSR_TSA_V"); vsel = v ? atoi(v) : 2; }  // 2 measured best: 54.9 (scalar) / 45.1 (x2) / 47.5 us (x4)
  if (HW % 4 == 0 && al16 && vsel == 4) {
    LAUNCH(tsa_gate_fwd_vec_kernel<4>, (size_t)B * N * HW / 4, st, emb, emb_ref, aligned, cor, gated, B, N, C, HW);
    return check_launch("tsa_gate_fwd_vec_kernel");
  }
  if (HW % 2 == 0 && al16 && vsel == 2) {
    LAUNCH(tsa_gate_fwd_vec_kernel<2>, (size_t)B * N * HW / 2, st, emb, emb_ref, aligned, cor, gated, B, N, C, HW);
    return check_launch("tsa_gate_fwd_vec_kernel");
  }
  LAUNCH(tsa_gate_fwd_kernel, (size_t)B * N * HW, st, emb, emb_ref, aligned, cor, gated, B, N, C, HW);
  return check_launch("tsa_gate_fwd_kernel");
}
int tsa_gate_bwd(const float* emb, const float* emb_ref, const float* aligned, const float* cor,
                 const float* g_gated, float* g_emb, float* g_emb_ref, float* g_aligned, int B,
                 int N, int C, size_t HW, hipStream_t st, float* gdot_scratch) {
  DVSR_REQUIRE(emb && emb_ref && aligned && cor && g_gated && g_emb && g_emb_ref && g_aligned,
               DVSR_ERR_INVALID, "tsa_gate_bwd: null pointer");
  if (gdot_scratch) {  // [B][N][HW] floats of scratch: the two-pass, N x more parallel variant
    LAUNCH(tsa_gate_bwd1_kernel, (size_t)B * N * HW, st, emb_ref, aligned, cor, g_gated, g_emb, g_aligned, gdot_scratch,
           B, N, C, HW);
    int rc = check_launch("tsa_gate_bwd1_kernel");
    if (rc) return rc;
    LAUNCH(tsa_gate_bwd2_kernel, (size_t)B * C * HW, st, emb, gdot_scratch, g_emb_ref, B, N, C, HW);
    return check_launch("tsa_gate_bwd2_kernel");
  }
  LAUNCH(tsa_gate_bwd_kernel, (size_t)B * HW, st, emb, emb_ref, aligned, cor, g_gated, g_emb,
         g_emb_ref, g_aligned, B, N, C, HW);
  return check_launch("tsa_gate_bwd_kernel");
}
int tsa_blend_fwd(const float* fea, const float* att, const float* add, float* out, size_t n,
                  hipStream_t st) {
  DVSR_REQUIRE(fea && att && add && out && n > 0, DVSR_ERR_INVALID, "tsa_blend_fwd: bad argument");
  if (n % 4 == 0 && (((uintptr_t)fea | (uintptr_t)att | (uintptr_t)add | (uintptr_t)out) & 15) == 0) {
    LAUNCH(tsa_blend_fwd4_kernel, n / 4, st, (const f32x4*)fea, (const f32x4*)att, (const f32x4*)add, (f32x4*)out, n / 4);
    return check_launch("tsa_blend_fwd4_kernel");
  }
  LAUNCH(tsa_blend_fwd_kernel, n, st, fea, att, add, out, n);
  return check_launch("tsa_blend_fwd_kernel");
}
int tsa_blend_bwd(const float* fea, const float* att, const float* g, float* g_fea, float* g_att_io,
                  size_t n, int accumulate, hipStream_t st) {
  DVSR_REQUIRE(fea && att && g && g_fea && g_att_io && n > 0, DVSR_ERR_INVALID,
               "tsa_blend_bwd: bad argument");
  LAUNCH(tsa_blend_bwd_kernel, n, st, fea, att, g, g_fea, g_att_io, n, accumulate);
  return check_launch("tsa_blend_bwd_kernel");
}
__global__ void add_out_kernel(float* __restrict__ y, const float* __restrict__ a, const float* __restrict__ b,
                               size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = a[i] + b[i];
}
__global__ void add_out4_kernel(f32x4* __restrict__ y, const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    y[i] = a[i] + b[i];
}
int add_out(float* y, const float* a, const float* b, size_t n, hipStream_t st) {
  DVSR_REQUIRE(y && a && b && n > 0, DVSR_ERR_INVALID, "add_out: bad argument");
  if (n % 4 == 0 && (((uintptr_t)y | (uintptr_t)a | (uintptr_t)b) & 15) == 0) {
    LAUNCH(add_out4_kernel, n / 4, st, (f32x4*)y, (const f32x4*)a, (const f32x4*)b, n / 4);
    return check_launch("add_out4_kernel");
  }
  LAUNCH(add_out_kernel, n, st, y, a, b, n);
  return check_launch("add_out_kernel");
}
int add_inplace(float* dst, const float* src, size_t n, hipStream_t st) {
  if (n % 4 == 0 && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
    LAUNCH(add_inplace4_kernel, n / 4, st, (f32x4*)dst, (const f32x4*)src, n / 4);
    return check_launch("add_inplace4_kernel");
  }
  LAUNCH(add_inplace_kernel, n, st, dst, src, n);
  return check_launch("add_inplace_kernel");
}
int act_bwd_inplace(float* g, const float* y, size_t n, int act, hipStream_t st) {
  if (act == ACT_NONE) return DVSR_OK;
  if (n % 4 == 0 && (((uintptr_t)g | (uintptr_t)y) & 15) == 0) {
    LAUNCH(act_bwd_inplace4_kernel, n / 4, st, (f32x4*)g, (const f32x4*)y, n / 4, act);
    return check_launch("act_bwd_inplace4_kernel");
  }
  LAUNCH(act_bwd_inplace_kernel, n, st, g, y, n, act);
  return check_launch("act_bwd_inplace_kernel");
}

int reduce_frames(float* dst, long long dst_bs, const float* src, int B, int cnt, size_t per,
                  int accumulate, hipStream_t st) {
  DVSR_REQUIRE(dst && src && B > 0 && cnt > 0 && per > 0, DVSR_ERR_INVALID, "reduce_frames: bad argument");
  LAUNCH(reduce_frames_kernel, (size_t)B * per, st, dst, dst_bs, src, B, cnt, per, accumulate);
  return check_launch("reduce_frames_kernel");
}

}  // namespace dvsr

using namespace dvsr;

extern "C" size_t dvsr_charbonnier_workspace_bytes(void) { return CHARB_BLOCKS * sizeof(float); }

extern "C" int dvsr_charbonnier_forward_grouped(const float* x, const float* y, float* loss, long long n, int groups,
                                                float eps, void* workspace, size_t workspace_bytes,
                                                dvsr_stream_t stream) {
  DVSR_REQUIRE(x && y && loss && workspace && n > 0 && groups > 0, DVSR_ERR_INVALID, "charbonnier_forward: bad argument");
  DVSR_REQUIRE(workspace_bytes >= (size_t)groups * CHARB_BLOCKS * sizeof(float), DVSR_ERR_WORKSPACE,
               "charbonnier_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int nb = (int)(((size_t)n + 255) / 256);
  if (nb > CHARB_BLOCKS) nb = CHARB_BLOCKS;
  hipLaunchKernelGGL(charbonnier_partial_kernel, dim3(nb, groups), dim3(256), 0, st, x, y, (float*)workspace,
                     (size_t)n, eps);
  hipLaunchKernelGGL(charbonnier_final_kernel, dim3(groups), dim3(256), 0, st, (const float*)workspace, loss, nb,
                     1.f / (float)n);
  return check_launch("charbonnier_forward");
}

extern "C" int dvsr_charbonnier_forward(const float* x, const float* y, float* loss, long long n, float eps,
                                        void* workspace, size_t workspace_bytes, dvsr_stream_t stream) {
  return dvsr_charbonnier_forward_grouped(x, y, loss, n, 1, eps, workspace, workspace_bytes, stream);
}

extern "C" int dvsr_charbonnier_backward_grouped(const float* x, const float* y, const float* grad_loss, float* gx,
                                                 long long n, int groups, float eps, dvsr_stream_t stream) {
  DVSR_REQUIRE(x && y && grad_loss && gx && n > 0 && groups > 0, DVSR_ERR_INVALID, "charbonnier_backward: bad argument");
  hipLaunchKernelGGL(charbonnier_bwd_kernel, dim3(stream_grid((size_t)n), groups), dim3(256), 0, (hipStream_t)stream, x, y,
                     grad_loss, gx, (size_t)n, eps, 1.f / (float)n);
  return check_launch("charbonnier_bwd_kernel");
}

extern "C" int dvsr_charbonnier_backward(const float* x, const float* y, const float* grad_loss, float* gx,
                                         long long n, float eps, dvsr_stream_t stream) {
  return dvsr_charbonnier_backward_grouped(x, y, grad_loss, gx, n, 1, eps, stream);
}

extern "C" int dvsr_l1_tail_forward_grouped(const float* x, const float* y, const float* base, float weight, float* loss,
                                            long long n, int groups, void* workspace, size_t workspace_bytes,
                                            dvsr_stream_t stream) {
  DVSR_REQUIRE(x && y && loss && workspace && n > 0 && groups > 0, DVSR_ERR_INVALID, "l1_tail_forward: bad argument");
  DVSR_REQUIRE(workspace_bytes >= (size_t)groups * CHARB_BLOCKS * sizeof(float), DVSR_ERR_WORKSPACE,
               "l1_tail_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int nb = (int)(((size_t)n + 255) / 256);
  if (nb > CHARB_BLOCKS) nb = CHARB_BLOCKS;
  hipLaunchKernelGGL(l1_partial_kernel, dim3(nb, groups), dim3(256), 0, st, x, y, (float*)workspace, (size_t)n);
  hipLaunchKernelGGL(l1_final_kernel, dim3(groups), dim3(256), 0, st, (const float*)workspace, base, loss, nb,
                     weight / (float)n);
  return check_launch("l1_tail_forward");
}

extern "C" int dvsr_l1_tail_forward(const float* x, const float* y, const float* base, float weight, float* loss,
                                    long long n, void* workspace, size_t workspace_bytes, dvsr_stream_t stream) {
  return dvsr_l1_tail_forward_grouped(x, y, base, weight, loss, n, 1, workspace, workspace_bytes, stream);
}

extern "C" int dvsr_l1_tail_backward_grouped(const float* x, const float* y, const float* grad_loss, float weight,
                                             float* gx, long long n, int groups, dvsr_stream_t stream) {
  DVSR_REQUIRE(x && y && grad_loss && gx && n > 0 && groups > 0, DVSR_ERR_INVALID, "l1_tail_backward: bad argument");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(stream_grid((size_t)n), groups), dim3(256), 0, (hipStream_t)stream, x, y, grad_loss,
                     gx, (size_t)n, weight / (float)n);
  return check_launch("l1_bwd_kernel");
}

extern "C" int dvsr_l1_tail_backward(const float* x, const float* y, const float* grad_loss, float weight, float* gx,
                                     long long n, dvsr_stream_t stream) {
  return dvsr_l1_tail_backward_grouped(x, y, grad_loss, weight, gx, n, 1, stream);
}

extern "C" int dvsr_upsample_bilinear_forward(const float* x, float* y, long long planes, int H,
                                              int W, int scale, float mul, dvsr_stream_t stream) {
  return upsample_bilinear_fwd(x, y, (size_t)planes, H, W, scale, mul, (hipStream_t)stream);
}
extern "C" int dvsr_upsample_bilinear_backward(const float* gy, float* gx, long long planes, int H,
                                               int W, int scale, float mul, int accumulate,
                                               dvsr_stream_t stream) {
  return upsample_bilinear_bwd(gy, gx, (size_t)planes, H, W, scale, mul, accumulate,
                               (hipStream_t)stream);
}
extern "C" int dvsr_pool3s2_forward(const float* x, float* ymax, float* yavg, long long planes,
                                    int H, int W, dvsr_stream_t stream) {
  return pool3s2_fwd(x, ymax, yavg, (size_t)planes, H, W, (hipStream_t)stream);
}
extern "C" int dvsr_pool3s2_backward(const float* x, const float* gmax, const float* gavg,
                                     float* gx, long long planes, int H, int W,
                                     dvsr_stream_t stream) {
  return pool3s2_bwd(x, gmax, gavg, gx, (size_t)planes, H, W, 0, (hipStream_t)stream);
}
extern "C" int dvsr_tsa_gate_forward(const float* emb, const float* emb_ref, const float* aligned,
                                     float* cor, float* gated, int B, int N, int C, long long HW,
                                     dvsr_stream_t stream) {
  return tsa_gate_fwd(emb, emb_ref, aligned, cor, gated, B, N, C, (size_t)HW, (hipStream_t)stream);
}
extern "C" int dvsr_tsa_gate_backward(const float* emb, const float* emb_ref, const float* aligned,
                                      const float* cor, const float* g_gated, float* g_emb,
                                      float* g_emb_ref, float* g_aligned, int B, int N, int C,
                                      long long HW, dvsr_stream_t stream) {
  return tsa_gate_bwd(emb, emb_ref, aligned, cor, g_gated, g_emb, g_emb_ref, g_aligned, B, N, C,
                      (size_t)HW, (hipStream_t)stream);
}
extern "C" int dvsr_tsa_blend_forward(const float* fea, const float* att, const float* att_add,
                                      float* out, long long n, dvsr_stream_t stream) {
  return tsa_blend_fwd(fea, att, att_add, out, (size_t)n, (hipStream_t)stream);
}
extern "C" int dvsr_tsa_blend_backward(const float* fea, const float* att, const float* g,
                                       float* g_fea, float* g_att_io, long long n,
                                       dvsr_stream_t stream) {
  return tsa_blend_bwd(fea, att, g, g_fea, g_att_io, (size_t)n, 1, (hipStream_t)stream);
}
