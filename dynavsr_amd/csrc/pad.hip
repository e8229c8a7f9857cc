// Streaming helpers of the MFDN / SFDN estimators (LRimg_estimator.py:38-117): explicit padding as
// the reference does it (nn.ReflectionPad2d(1) :46,78 / nn.ReplicationPad3d(1) :77) plus the layout
// changes that let every convolution of the estimator run on the dense stride-1 MFMA kernels:
//   PAD_REFLECT      out[n][c][y][x]              = in[n][c][refl(y-1)][refl(x-1)]          (H+2 x W+2)
//   PAD_REFLECT_S2D  out[n][4c+2dy+dx][Y][X]      = reflect-padded in at (2Y+dy, 2X+dx): the 4x4 stride-2
//                    convolutions (:83-85) become 2x2 stride-1 convolutions over 4C channels
//   PAD_REPL_T3      out[b*T+t][3c+dt][y][x]      = in[b*T+clamp(t+dt-1)][c][clamp(y-1)][clamp(x-1)]: the
//                    3x3x3 Conv3d over ReplicationPad3d(1) (:76,88) becomes a 3x3 conv over 3C channels
//                    whose weight tensor [Cout][C][3][3][3] is already laid out as [Cout][3C][3][3]
// and their adjoints (gradient folds), the per-frame mean subtraction / re-addition (:93,116) with
// the B,C,T,H,W <-> (B*T),C,H,W transposes folded in, and the 4x4 -> 2x2-over-4C weight re-layout.
// All of these are HBM-bound elementwise kernels (<1 FLOP/B).
#include <cstdint>

#include "common.h"
#include "kernels.h"

namespace dvsr {

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ int clamp0(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// one thread per output element; blockIdx.y walks the output planes (n, output channel), so the only
// per-thread integer division is pixel -> (row, column)
__global__ void pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int mode, int C, int H,
                               int W, int T) {
  const int Hp = H + 2, Wp = W + 2;
  const int Ho = mode == PAD_REFLECT_S2D ? Hp / 2 : Hp, Wo = mode == PAD_REFLECT_S2D ? Wp / 2 : Wp;
  const int opix = Ho * Wo;
  const int cmul = mode == PAD_REFLECT ? 1 : (mode == PAD_REFLECT_S2D ? 4 : 3);
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const int n = plane / (cmul * C), cc = plane - n * cmul * C;
    const float* src;
    int dy = 0, dx = 0;
    if (mode == PAD_REFLECT) {
      src = x + (size_t)plane * H * W;
    } else if (mode == PAD_REFLECT_S2D) {
      src = x + ((size_t)n * C + (cc >> 2)) * H * W;
      dy = (cc >> 1) & 1; dx = cc & 1;
    } else {
      const int c = cc / 3, dt = cc - 3 * c;
      const int b = n / T, tt = n - b * T;
      src = x + (((size_t)b * T + clamp0(tt + dt - 1, T)) * C + c) * H * W;
    }
    float* dst = y + (size_t)plane * opix;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < opix; i += gridDim.x * blockDim.x) {
      const int r = i / Wo, q = i - r * Wo;
      int sy, sx;
      if (mode == PAD_REFLECT) { sy = reflect1(r - 1, H); sx = reflect1(q - 1, W); }
      else if (mode == PAD_REFLECT_S2D) { sy = reflect1(2 * r + dy - 1, H); sx = reflect1(2 * q + dx - 1, W); }
      else { sy = clamp0(r - 1, H); sx = clamp0(q - 1, W); }
      dst[i] = src[sy * W + sx];
    }
  }
}

// adjoint: one thread per INPUT element, gathering every padded position that maps to it
// gmask_padded: the mask tensor is the PADDED forward tensor itself (same layout as gy) -- the producing conv stored
// straight into it (PS_PAD_REFLECT*), a dense copy of its output does not exist; element (yy, xx) sits at padded
// position (yy + 1, xx + 1).
template <int mode>
__global__ void pad_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int planes, int C,
                               int H, int W, int T, int accumulate, const float* __restrict__ gmask,
                               int gmask_act, int gmask_padded) {
  const int Hp = H + 2, Wp = W + 2, ipix = H * W;
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {  // plane = n*C + c
    const int n = plane / C, c = plane - n * C;
    // source planes: 1 (reflect), 4 interleaved (space-to-depth) or up to 5 (t, dt) pairs (temporal gather)
    const float* sp[5];
    int np = 0;
    if (mode == PAD_REFLECT) {
      sp[np++] = gy + (size_t)plane * Hp * Wp;
    } else if (mode == PAD_REFLECT_S2D) {
      sp[np++] = gy + ((size_t)n * 4 * C + c * 4) * (size_t)(Hp / 2) * (Wp / 2);
    } else {
      const int b = n / T, tt = n - b * T;
      for (int dt = 0; dt < 3; ++dt) {  // (t, dt) with clamp(t + dt - 1) == tt
        const int t0 = tt - dt + 1;
        if (t0 >= 0 && t0 < T) sp[np++] = gy + (((size_t)b * T + t0) * 3 * C + c * 3 + dt) * (size_t)Hp * Wp;
      }
      if (tt == 0) sp[np++] = gy + (((size_t)b * T) * 3 * C + c * 3) * (size_t)Hp * Wp;                  // -1 -> 0
      if (tt == T - 1) sp[np++] = gy + (((size_t)b * T + T - 1) * 3 * C + c * 3 + 2) * (size_t)Hp * Wp;  // T -> T-1
    }
    float* dst = gx + (size_t)plane * ipix;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ipix; i += gridDim.x * blockDim.x) {
      const int yy = i / W, xx = i - yy * W;
      // the padded position every element is read at: (yy + 1, xx + 1); elements next to the border are
      // read a second (third) time through the padding ring -- handled in the rare branch below
      auto at = [&](int k, int r, int q) -> float {
        if (mode == PAD_REFLECT_S2D) {
          const int Wh = Wp / 2, Hh = Hp / 2;
          return sp[0][((size_t)((r & 1) * 2 + (q & 1)) * Hh + (r >> 1)) * Wh + (q >> 1)];
        }
        return sp[k][r * Wp + q];
      };
      constexpr int NPL = mode == PAD_REPL_T3 ? 5 : 1;  // compile-time bound: sp[] stays in registers
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NPL; ++k)
        if (k < np) s += at(k, yy + 1, xx + 1);
      const int lo_edge = mode == PAD_REPL_T3 ? 0 : 1;  // replicate: the border itself; reflect: one inside
      const bool ry0 = yy == lo_edge, ry1 = yy == H - 1 - lo_edge, cx0 = xx == lo_edge, cx1 = xx == W - 1 - lo_edge;
      if (ry0 || ry1 || cx0 || cx1) {
        int rows[3], cols[3], nr = 0, nc = 0;
        rows[nr++] = yy + 1;
        cols[nc++] = xx + 1;
        if (ry0) rows[nr++] = 0;
        if (ry1) rows[nr++] = Hp - 1;
        if (cx0) cols[nc++] = 0;
        if (cx1) cols[nc++] = Wp - 1;
#pragma unroll
        for (int k = 0; k < NPL; ++k)
          if (k < np)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b)
                if (a < nr && b < nc && (a || b)) s += at(k, rows[a], cols[b]);
      }
      if (accumulate) s += dst[i];
      if (gmask) {  // fused activation backward of the layer that produced the padded tensor
        const float neg = gmask_act == ACT_LRELU ? 0.1f : (gmask_act == ACT_RELU ? 0.f : 1.f);
        float mv;
        if (!gmask_padded) mv = gmask[(size_t)plane * ipix + i];
        else if (mode == PAD_REFLECT_S2D) {
          const int Wh = Wp / 2, Hh = Hp / 2, r = yy + 1, q = xx + 1;
          mv = gmask[((size_t)n * 4 * C + c * 4 + (r & 1) * 2 + (q & 1)) * (size_t)Hh * Wh + (size_t)(r >> 1) * Wh + (q >> 1)];
        } else mv = gmask[(size_t)plane * Hp * Wp + (size_t)(yy + 1) * Wp + xx + 1];
        s *= mv > 0.f ? 1.f : neg;
      }
      dst[i] = s;
    }
  }
}

// Four consecutive columns per thread (W % 4 == 0): the interior loads of the four elements are issued
// together and the mask / output move as 16-byte accesses.  The one-element kernel above is latency-bound
// (two dependent round trips per thread: 235 us for 217 MB at 5x64x176x320).
template <int mode, bool ACC, bool MASK>
__global__ void pad_bwd4_kernel(const float* __restrict__ gy, float* __restrict__ gx, int planes, int C,
                                int H, int W, int T, int accumulate, const float* __restrict__ gmask,
                                int gmask_act, int gmask_padded) {
  const int Hp = H + 2, Wp = W + 2, Wq = W >> 2, iq = H * Wq;
  constexpr int NPL = mode == PAD_REPL_T3 ? 5 : 1;
  const int lo_edge = mode == PAD_REPL_T3 ? 0 : 1;
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const int n = plane / C, c = plane - n * C;
    const float* sp[NPL];
    int np = 0;
    if (mode == PAD_REFLECT) {
      sp[np++] = gy + (size_t)plane * Hp * Wp;
    } else if (mode == PAD_REFLECT_S2D) {
      sp[np++] = gy + ((size_t)n * 4 * C + c * 4) * (size_t)(Hp / 2) * (Wp / 2);
    } else {
      const int b = n / T, tt = n - b * T;
#pragma unroll
      for (int k = 0; k < NPL; ++k) sp[k] = gy;
      for (int dt = 0; dt < 3; ++dt) {
        const int t0 = tt - dt + 1;
        if (t0 >= 0 && t0 < T) sp[np++] = gy + (((size_t)b * T + t0) * 3 * C + c * 3 + dt) * (size_t)Hp * Wp;
      }
      if (tt == 0) sp[np++] = gy + (((size_t)b * T) * 3 * C + c * 3) * (size_t)Hp * Wp;
      if (tt == T - 1) sp[np++] = gy + (((size_t)b * T + T - 1) * 3 * C + c * 3 + 2) * (size_t)Hp * Wp;
    }
    auto at = [&](int k, int r, int q) -> float {
      if (mode == PAD_REFLECT_S2D) {
        const int Wh = Wp / 2, Hh = Hp / 2;
        return sp[0][((size_t)((r & 1) * 2 + (q & 1)) * Hh + (r >> 1)) * Wh + (q >> 1)];
      }
      return sp[k][r * Wp + q];
    };
    float* dst = gx + (size_t)plane * H * W;
    // four consecutive elements (row r, columns q0 .. q0 + 3, q0 odd) of a padded plane: 4-byte aligned vector loads (the
    // padded pitch is W + 2: rows start at every alignment); in the space-to-depth layout the four are two pairs in the two
    // column-parity planes of row parity r & 1
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    auto row4 = [&](const float* base, int r, int q0) -> f32x4 {
      if (mode == PAD_REFLECT_S2D) {
        const int Wh = Wp / 2, Hh = Hp / 2;
        const float* pr = base + ((size_t)((r & 1) * 2) * Hh + (r >> 1)) * Wh;
        const f2u odd = *reinterpret_cast<const f2u*>(pr + (size_t)Hh * Wh + (q0 >> 1));   // q0, q0 + 2
        const f2u evn = *reinterpret_cast<const f2u*>(pr + ((q0 + 1) >> 1));               // q0 + 1, q0 + 3
        return f32x4{odd[0], evn[0], odd[1], evn[1]};
      }
      const f4u v = *reinterpret_cast<const f4u*>(base + (size_t)r * Wp + q0);
      return f32x4{v[0], v[1], v[2], v[3]};
    };
    const float* mbase = nullptr;   // the mask plane in the layout it has
    if (gmask) {
      if (!gmask_padded) mbase = gmask + (size_t)plane * H * W;
      else if (mode == PAD_REFLECT_S2D) mbase = gmask + ((size_t)n * 4 * C + c * 4) * (size_t)(Hp / 2) * (Wp / 2);
      else mbase = gmask + (size_t)plane * Hp * Wp;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < iq; i += gridDim.x * blockDim.x) {
      const int yy = i / Wq, x0 = (i - yy * Wq) * 4;
      // every load of the element group goes out before anything is consumed (the border branch below used to sit between the
      // gradient and the mask loads: two dependent round trips per thread)
      f32x4 v[NPL];
#pragma unroll
      for (int k = 0; k < NPL; ++k)
        if (k < np) v[k] = row4(sp[k], yy + 1, x0 + 1);
      f32x4* d4 = reinterpret_cast<f32x4*>(dst + (size_t)yy * W + x0);
      f32x4 m = {1.f, 1.f, 1.f, 1.f}, prev = {0.f, 0.f, 0.f, 0.f};
      if constexpr (MASK) {   // (one load at a selected address: no branch between the loads)
        if (mode == PAD_REFLECT_S2D) {
          m = gmask_padded ? row4(mbase, yy + 1, x0 + 1) : *reinterpret_cast<const f32x4*>(mbase + (size_t)yy * W + x0);
        } else {
          const float* ma = gmask_padded ? mbase + (size_t)(yy + 1) * Wp + x0 + 1 : mbase + (size_t)yy * W + x0;
          const f4u mv = *reinterpret_cast<const f4u*>(ma);
          m = f32x4{mv[0], mv[1], mv[2], mv[3]};
        }
      }
      if constexpr (ACC) prev = *d4;
      // The ring's adjoint is separable: output (yy, xx) sums g over rows {yy + 1} (+ ring row 0 / Hp - 1 when yy is the row
      // the ring mirrors or replicates) x columns {xx + 1} (+ ring column 0 / Wp - 1 likewise).  Per row that is the four-element
      // vector plus, for the thread at either end of the row, ONE ring element -- loaded by every lane at a clamped address and
      // weighted 0 / 1 (no divergent load chains: nearly every wave holds a row end).
      const int jl = lo_edge, jr = 3 - lo_edge;                  // which of the four elements the ring column folds into
      const float wl = x0 == 0 ? 1.f : 0.f, wr = x0 == W - 4 ? 1.f : 0.f;
      float el[NPL], er[NPL];
#pragma unroll
      for (int k = 0; k < NPL; ++k)
        if (k < np) { el[k] = at(k, yy + 1, 0); er[k] = at(k, yy + 1, Wp - 1); }
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NPL; ++k)
        if (k < np) {
#pragma unroll
          for (int j = 0; j < 4; ++j) s[j] += v[k][j];
          s[jl] += wl * el[k];
          s[jr] += wr * er[k];
        }
      const bool ry0 = yy == lo_edge, ry1 = yy == H - 1 - lo_edge;
      if (ry0 || ry1) {   // (two rows of a plane; both at once only when H - 1 = 2 lo_edge)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (!(e ? ry1 : ry0)) continue;
          const int rr = e ? Hp - 1 : 0;
#pragma unroll
          for (int k = 0; k < NPL; ++k)
            if (k < np) {
              const f32x4 u = row4(sp[k], rr, x0 + 1);
              const float ul = at(k, rr, 0), ur = at(k, rr, Wp - 1);
#pragma unroll
              for (int j2 = 0; j2 < 4; ++j2) s[j2] += u[j2];
              s[jl] += wl * ul;
              s[jr] += wr * ur;
            }
        }
      }
      f32x4 o = {s[0], s[1], s[2], s[3]};
      if constexpr (ACC) o += prev;
      if constexpr (MASK) {
        const float neg = gmask_act == ACT_LRELU ? 0.1f : (gmask_act == ACT_RELU ? 0.f : 1.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] *= m[j] > 0.f ? 1.f : neg;
      }
      *d4 = o;
    }
  }
}

static int grid_for(size_t total) {
  const size_t b = (total + 255) / 256;
  return (int)(b < 65535u * 16u ? (b ? b : 1) : 65535u * 16u);
}

size_t pad_out_numel(int mode, size_t N, int C, int H, int W) {
  if (mode == PAD_REFLECT) return N * C * (size_t)(H + 2) * (W + 2);
  if (mode == PAD_REFLECT_S2D) return N * 4 * C * (size_t)((H + 2) / 2) * ((W + 2) / 2);
  return N * 3 * C * (size_t)(H + 2) * (W + 2);
}

int pad_fwd(const float* x, float* y, int mode, int N, int C, int H, int W, int T, hipStream_t st) {
  DVSR_REQUIRE(x && y, DVSR_ERR_INVALID, "pad_fwd: null pointer");
  DVSR_REQUIRE(mode >= PAD_REFLECT && mode <= PAD_REPL_T3, DVSR_ERR_INVALID, "pad_fwd: mode %d", mode);
  DVSR_REQUIRE(mode == PAD_REPL_T3 || (H >= 2 && W >= 2), DVSR_ERR_INVALID, "pad_fwd: reflect needs H, W >= 2");
  DVSR_REQUIRE(mode != PAD_REFLECT_S2D || (H % 2 == 0 && W % 2 == 0), DVSR_ERR_INVALID,
               "pad_fwd: space-to-depth needs even H, W (got %dx%d)", H, W);
  DVSR_REQUIRE(mode != PAD_REPL_T3 || (T > 0 && N % T == 0), DVSR_ERR_INVALID, "pad_fwd: N=%d not a multiple of T=%d", N, T);
  const int cmul = mode == PAD_REFLECT ? 1 : (mode == PAD_REFLECT_S2D ? 4 : 3);
  const int planes = N * C * cmul;
  const int opix = (int)(pad_out_numel(mode, 1, 1, H, W) / cmul);
  const dim3 grid(ceil_div(opix, 1024) > 0 ? ceil_div(opix, 1024) : 1, planes < 65535 ? planes : 65535);
  hipLaunchKernelGGL(pad_fwd_kernel, grid, dim3(256), 0, st, x, y, planes, mode, C, H, W, T);
  return check_launch("pad_fwd_kernel");
}

int pad_bwd(const float* gy, float* gx, int mode, int N, int C, int H, int W, int T, int accumulate,
            hipStream_t st, const float* gmask, int gmask_act, int gmask_padded) {
  DVSR_REQUIRE(gy && gx, DVSR_ERR_INVALID, "pad_bwd: null pointer");
  DVSR_REQUIRE(mode >= PAD_REFLECT && mode <= PAD_REPL_T3, DVSR_ERR_INVALID, "pad_bwd: mode %d", mode);
  const int planes = N * C;
  if (W % 4 == 0 && (((uintptr_t)gx | (gmask_padded ? (uintptr_t)0 : (uintptr_t)gmask)) & 15) == 0) {
    const dim3 g4(ceil_div(H * (W / 4), 256), planes < 65535 ? planes : 65535);
    auto go = [&](auto mode_, auto acc_, auto mask_) {
      hipLaunchKernelGGL((pad_bwd4_kernel<decltype(mode_)::value, decltype(acc_)::value, decltype(mask_)::value>), g4, dim3(256), 0, st,
                         gy, gx, planes, C, H, W, T, accumulate, gmask, gmask_act, gmask_padded);
    };
    auto pick = [&](auto mode_) {
      if (accumulate) { if (gmask) go(mode_, std::true_type{}, std::true_type{}); else go(mode_, std::true_type{}, std::false_type{}); }
      else { if (gmask) go(mode_, std::false_type{}, std::true_type{}); else go(mode_, std::false_type{}, std::false_type{}); }
    };
    if (mode == PAD_REFLECT) pick(std::integral_constant<int, PAD_REFLECT>{});
    else if (mode == PAD_REFLECT_S2D) pick(std::integral_constant<int, PAD_REFLECT_S2D>{});
    else pick(std::integral_constant<int, PAD_REPL_T3>{});
    return check_launch("pad_bwd4_kernel");
  }
  const dim3 grid(ceil_div(H * W, 256), planes < 65535 ? planes : 65535);
  if (mode == PAD_REFLECT)
    hipLaunchKernelGGL(pad_bwd_kernel<PAD_REFLECT>, grid, dim3(256), 0, st, gy, gx, planes, C, H, W, T, accumulate, gmask,
                       gmask_act, gmask_padded);
  else if (mode == PAD_REFLECT_S2D)
    hipLaunchKernelGGL(pad_bwd_kernel<PAD_REFLECT_S2D>, grid, dim3(256), 0, st, gy, gx, planes, C, H, W, T, accumulate,
                       gmask, gmask_act, gmask_padded);
  else
    hipLaunchKernelGGL(pad_bwd_kernel<PAD_REPL_T3>, grid, dim3(256), 0, st, gy, gx, planes, C, H, W, T, accumulate, gmask,
                       gmask_act, gmask_padded);
  return check_launch("pad_bwd_kernel");
}

// ---- per-frame mean (LRimg_estimator.py:93: x.mean(-1).mean(-2)) ------------------------------------
// x: [B][C][T][H][W]  ->  xm: [(B*T)][C][H][W] = x - mean,  mean: [B][C][T].
// Pass 1: MS_SLICES workgroups per plane sum the row means of their rows (mean over W, then over H, like
// the reference); pass 2 finishes the mean from the slice sums and subtracts.
constexpr int MS_SLICES = 16;
__global__ __launch_bounds__(256) void rowmean_kernel(const float* __restrict__ x, float* __restrict__ part, int H,
                                                      int W) {
  const int plane = blockIdx.y, slice = blockIdx.x;
  const float* src = x + (size_t)plane * H * W;
  __shared__ float s_part[4];
  float acc = 0.f;
  for (int row = slice * 4 + (threadIdx.x >> 6); row < H; row += 4 * MS_SLICES) {  // one wave per row
    float s = 0.f;
    for (int col = threadIdx.x & 63; col < W; col += 64) s += src[(size_t)row * W + col];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    acc += s / (float)W;
  }
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[plane * MS_SLICES + slice] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ __launch_bounds__(256) void meansub_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                      float* __restrict__ xm, float* __restrict__ mean, int C, int T,
                                                      int H, int W) {
  const int plane = blockIdx.y;  // (b*C + c)*T + t
  const int t = plane % T, c = (plane / T) % C, b = plane / (T * C);
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < MS_SLICES; ++k) m += part[plane * MS_SLICES + k];
  m /= (float)H;
  if (blockIdx.x == 0 && threadIdx.x == 0) mean[plane] = m;
  const int HW = H * W;
  const float* src = x + (size_t)plane * HW;
  float* dst = xm + (((size_t)b * T + t) * C + c) * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) dst[i] = src[i] - m;
}

// out[b][c][t][p] = y[(b*T+t)][c][p] + mean[b][c][t]
__global__ void addmean_kernel(const float* __restrict__ y, const float* __restrict__ mean, float* __restrict__ out,
                               size_t total, int C, int T, size_t HW) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    size_t r = i / HW;  // (b*C + c)*T + t
    const int t = (int)(r % T);
    const size_t bc = r / T;
    const int c = (int)(bc % C);
    const size_t b = bc / C;
    out[i] = y[((b * T + t) * C + c) * HW + p] + mean[r];
  }
}

// gy[(b*T+t)][c][p] = gout[b][c][t][p]
__global__ void addmean_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gy, size_t total, int C, int T,
                                   size_t HW) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % HW;
    size_t r = i / HW;  // (b*T + t)*C + c
    const int c = (int)(r % C);
    const size_t bt = r / C;
    const int t = (int)(bt % T);
    const size_t b = bt / T;
    gy[i] = gout[((b * C + c) * T + t) * HW + p];
  }
}

// part: B*C*T*MS_SLICES floats of scratch
int meansub_fwd(const float* x, float* xm, float* mean, float* part, int B, int C, int T, int H, int W,
                hipStream_t st) {
  DVSR_REQUIRE(x && xm && mean && part, DVSR_ERR_INVALID, "meansub_fwd: null pointer");
  const int planes = B * C * T;
  DVSR_REQUIRE(planes <= 65535, DVSR_ERR_UNSUPPORTED, "meansub_fwd: %d planes", planes);
  hipLaunchKernelGGL(rowmean_kernel, dim3(MS_SLICES, planes), dim3(256), 0, st, x, part, H, W);
  int rc = check_launch("rowmean_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(meansub_kernel, dim3(ceil_div(H * W, 1024), planes), dim3(256), 0, st, x, part, xm, mean, C, T,
                     H, W);
  return check_launch("meansub_kernel");
}
int meansub_slices() { return MS_SLICES; }

int addmean_fwd(const float* y, const float* mean, float* out, int B, int C, int T, size_t HW, hipStream_t st) {
  DVSR_REQUIRE(y && mean && out, DVSR_ERR_INVALID, "addmean_fwd: null pointer");
  const size_t total = (size_t)B * C * T * HW;
  hipLaunchKernelGGL(addmean_kernel, dim3(grid_for(total)), dim3(256), 0, st, y, mean, out, total, C, T, HW);
  return check_launch("addmean_kernel");
}

int addmean_bwd(const float* gout, float* gy, int B, int C, int T, size_t HW, hipStream_t st) {
  DVSR_REQUIRE(gout && gy, DVSR_ERR_INVALID, "addmean_bwd: null pointer");
  const size_t total = (size_t)B * C * T * HW;
  hipLaunchKernelGGL(addmean_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, gout, gy, total, C, T, HW);
  return check_launch("addmean_bwd_kernel");
}

// ---- 4x4 stride-2 weights <-> 2x2 weights over the space-to-depth input -------------------------------
// w2[o][4c + 2dy + dx][ky][kx] = w[o][c][2ky + dy][2kx + dx]   (inverse = 1: the same map, other direction)
__global__ void w4_s2d_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t total, int C,
                              int inverse) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i indexes w2: [o][c4][ky][kx]
    const int kx = (int)(i & 1), ky = (int)((i >> 1) & 1);
    const size_t r = i >> 2;
    const int c4 = (int)(r % (4 * C));
    const size_t o = r / (4 * C);
    const int c = c4 >> 2, dy = (c4 >> 1) & 1, dx = c4 & 1;
    const size_t j = ((o * C + c) * 4 + 2 * ky + dy) * 4 + 2 * kx + dx;
    if (inverse) dst[j] = src[i];
    else dst[i] = src[j];
  }
}

int w4_to_s2d(const float* w, float* w2, int Cout, int C, int inverse, hipStream_t st) {
  DVSR_REQUIRE(w && w2, DVSR_ERR_INVALID, "w4_to_s2d: null pointer");
  const size_t total = (size_t)Cout * C * 16;
  hipLaunchKernelGGL(w4_s2d_kernel, dim3(grid_for(total)), dim3(256), 0, st, w, w2, total, C, inverse);
  return check_launch("w4_s2d_kernel");
}

}  // namespace dvsr
