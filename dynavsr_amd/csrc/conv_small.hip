// 3x3 convolution with very few output channels (EDVR's conv_last: 64 -> 3 on the 720x1280 HR grid).
//
// On the MFMA tile this layer computes a 32-wide cout block to keep 3 of it (9.9 TFLOP/s useful,
// 322 us in the r01 profile).  It is really an HBM-bound streaming op (read 64 planes once, write 3):
// one thread = one output pixel x all COUT channels on the VALU, the 8-channel halo tile staged in LDS
// exactly like conv2d.hip, weights read with wave-uniform (scalar) loads.  Bias, activation and the
// residual add (the bilinear base frame, EDVR_arch.py:311-312) are fused.
#include "common.h"
#include "kernels.h"

namespace dvsr {

struct SmallK {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int N, C, H, W, Cout, act, tiles_x, tiles_y, ntiles;
  int wdiv; long long w_gs; int b_gs;   // per-sample weight sets (common.h: wset_ptr)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One thread = TWO horizontally adjacent output pixels (one packed fp32 pair: v_pk_fma_f32 does 2 FMAs per
// lane per instruction) x COUT channels, tile = 8 rows x 64 columns, 8 input channels per chunk.  A window
// row for both pixels is one ds_read_b128.  The weights (w[o][ci][tap] at ci*WPC + tap*COUT + o) are staged
// in LDS once and read as broadcast float4s: scalar loads share lgkmcnt with the LDS reads and cost a full
// SMEM round trip several times per channel (125 -> 105 us).  Measured alternatives that were slower:
// 16-row tiles (130 us), four pixels per thread with 4-channel chunks (133 us).
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_cout_kernel(SmallK a) {
  constexpr int CC = 8, TH = 8, TW = 64, IH = TH + 2, IW = TW + 2, PLANE = IH * IW, E = (PLANE + 255) / 256;
  constexpr int WPC = ((9 * COUT + 3) / 4) * 4;
  __shared__ __attribute__((aligned(16))) float s_in[2][CC * PLANE];
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [ceil(C/8)*8][WPC]
  // an XCD (blockIdx.x & 7: workgroups are dealt round-robin) owns a band of tile rows, so the two halo rows a
  // tile shares with its vertical neighbours are L2 hits
  const int tpx = (a.ntiles + 7) >> 3;
  const int tile = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= tpx || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * TW;
  const int tid = threadIdx.x;
  const int py = tid >> 5, px2 = (tid & 31) * 2;
  const size_t HW = (size_t)a.H * a.W;
  const float* xn = a.x + (size_t)n * a.C * HW;
  const float* wn = wset_ptr(a.w, a.w_gs, n, a.wdiv);
  const float* bn = wset_ptr(a.bias, a.b_gs, n, a.wdiv);

  unsigned eoff[E];
  bool evalid[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = tid + 256 * e;
    const int iy = idx / IW, ix = idx - iy * IW;
    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
    evalid[e] = idx < PLANE && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    eoff[e] = evalid[e] ? (unsigned)(gy * a.W + gx) * 4u : 0u;
  }
  f32x2 acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = f32x2{0.f, 0.f};
  float rin[CC][E];
  const int nch = ceil_div(a.C, CC);
  for (int i = tid; i < nch * CC * WPC; i += 256) {
    const int ci = i / WPC, r = i - ci * WPC;
    const int tap = r / COUT, o = r - tap * COUT;
    s_w[i] = (ci < a.C && tap < 9) ? wn[((size_t)o * a.C + ci) * 9 + tap] : 0.f;
  }
  auto prefetch = [&](int k) {
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      const int ci = k * CC + c;
      const char* src = reinterpret_cast<const char*>(xn + (size_t)(ci < a.C ? ci : 0) * HW);  // scalar base
#pragma unroll
      for (int e = 0; e < E; ++e) rin[c][e] = *reinterpret_cast<const float*>(src + eoff[e]);
    }
  };
  prefetch(0);
  for (int k = 0; k < nch; ++k) {
    float* s = s_in[k & 1];
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int idx = tid + 256 * e;
        if (e + 1 < E || idx < PLANE) s[c * PLANE + idx] = (evalid[e] && k * CC + c < a.C) ? rin[c][e] : 0.f;
      }
    __syncthreads();  // (also orders the one-time weight staging before its first use)
    if (k + 1 < nch) prefetch(k + 1);
    const float* p0 = s + py * IW + px2;
#pragma unroll
    for (int c = 0; c < CC; ++c) {  // channels >= C hold zeros in both LDS images: no branch needed
      f32x4 wq[WPC / 4];
#pragma unroll
      for (int q = 0; q < WPC / 4; ++q)
        wq[q] = *reinterpret_cast<const f32x4*>(s_w + (size_t)(k * CC + c) * WPC + 4 * q);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const f32x2 lo = *reinterpret_cast<const f32x2*>(p0 + c * PLANE + ty * IW);
        const f32x2 hi = *reinterpret_cast<const f32x2*>(p0 + c * PLANE + ty * IW + 2);
        const float v[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            const int wi = (ty * 3 + tx) * COUT + o;
            const float wv = wq[wi >> 2][wi & 3];
            acc[o] = __builtin_elementwise_fma(f32x2{wv, wv}, f32x2{v[tx], v[tx + 1]}, acc[o]);
          }
      }
    }
  }
  const int oy = oy0 + py, ox = ox0 + px2;
  if (oy >= a.H || ox >= a.W) return;
  const bool two = ox + 1 < a.W;
  // (bias and residual of all COUT channels first: read inside the store loop every load waited behind the previous store)
  float bvs[COUT];
  f32x2 rvs[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    bvs[o] = bn ? bn[o] : 0.f;
    const size_t idx = ((size_t)n * a.Cout + o) * HW + (size_t)oy * a.W + ox;
    rvs[o] = (a.res && two && (idx & 1) == 0) ? *reinterpret_cast<const f32x2*>(a.res + idx) : f32x2{0.f, 0.f};
  }
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    const float bv = bvs[o];
    f32x2 v = {apply_act(acc[o][0] + bv, a.act), apply_act(acc[o][1] + bv, a.act)};
    const size_t idx = ((size_t)n * a.Cout + o) * HW + (size_t)oy * a.W + ox;
    if (two && (idx & 1) == 0) {  // whole, 8-byte aligned pair
      v += rvs[o];
      *reinterpret_cast<f32x2*>(a.y + idx) = v;
    } else {
      a.y[idx] = v[0] + (a.res ? a.res[idx] : 0.f);
      if (two) a.y[idx + 1] = v[1] + (a.res ? a.res[idx + 1] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same layer on the fp32 matrix pipe.  The VALU kernel above is bound by its LDS reads (per channel a wave reads
// seven broadcast float4s of weights and six 8-byte window pairs for 27 packed FMAs: 100 us at 1 x 64 x 720 x 1280, 2.6 TB/s).
// A 3x3 conv with COUT <= 3 outputs is a 1x1 conv with 9 COUT <= 27 outputs followed by a shift-add:
//     P[(tap, o)][q] = sum_c w[o][c][tap] x[c][q]          (a 32 x 64 x positions GEMM: v_mfma_f32_32x32x2_f32, M padded to 32)
//     y[o][p]        = sum_tap P[(tap, o)][p + shift(tap)]  (27 adds per output through the LDS)
// over the output tile's HALO tile (zero outside the image, which is the conv's zero padding).  Workgroup = 6 x 56 outputs,
// halo tile 8 rows x 64 columns starting 4 columns left of the tile (16-byte aligned groups, each wholly inside or outside the
// image: W % 4 == 0): wave w owns halo rows 2 w, 2 w + 1 = 128 positions = four 32-position MFMA blocks with the position index
// permuted so that block j holds positions {4 n + j} -- lane n's B operands of a K step are ONE 16-byte load straight from
// global memory (no LDS staging: every x value is used by exactly one lane; buffer loads, out-of-image groups carry an offset
// beyond num_records and read zeros) and its results per P row are four consecutive positions, one ds_write_b128.  The K index
// is permuted too (step i of a 16-channel chunk contracts channels c0 + i and c0 + 8 + i): the 32 x 64 weight matrix is
// staged once per workgroup and lives in 32 registers per lane.  1.52 x the pixels go through the MFMAs (halo), 38 us of
// matrix-pipe time at 720 x 1280; P is 54 KB, two workgroups per CU.
// MEASURED (profiles/r05_conv_last_mfma.txt): 106 us against the VALU kernel's 101 at 1 x 64 x 720 x 1280 -- NOT faster, so it
// is off by default (DVSR_CONV_LAST_MFMA=1 enables it; its parity test runs it in a child).  Ablations: without the x loads
// 65 us (38 of MFMAs + weight staging and shift-add per workgroup), without the MFMAs 70 us (353 MB through the L2 with
// the halo, 5 TB/s): the two phases of a workgroup do not overlap -- all loads go out at its start, nothing is in flight for
// the next tile while it multiplies, and two co-resident workgroups interleave them only by chance.  What it would take:
// persistent workgroups that fetch tile i + 1 under the MFMAs of tile i (the B registers are there: 128), taller tiles
// (14 x 56 on eight waves: 1.31 x halo).
struct SmallMfmaShape {
  static constexpr int TH = 6, TW = 56, RH = 8, RW = 64, X0 = 4, NPOS = RH * RW, APITCH = 68;
  static constexpr size_t LDS_BYTES = (size_t)(32 * APITCH + 27 * NPOS) * sizeof(float);
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sm_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 sm_load16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
}

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_small_cout_mfma_kernel(SmallK a) {   // a.C == 64
  using Sh = SmallMfmaShape;
  constexpr int TH = Sh::TH, TW = Sh::TW, RW = Sh::RW, NPOS = Sh::NPOS, AP = Sh::APITCH, NM = 9 * COUT;
  extern __shared__ __attribute__((aligned(16))) float s_mf[];
  float* const s_a = s_mf;             // [32][AP]: row m = tap * COUT + o, columns = input channels
  float* const s_p = s_mf + 32 * AP;   // [27][NPOS]
  const int tpx = (a.ntiles + 7) >> 3;   // an XCD owns a band of tile rows (as above)
  const int tile = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= tpx || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * TW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 31, kk = lane >> 5;
  const unsigned HW4 = (unsigned)a.H * a.W * 4u;
  const float* wn = wset_ptr(a.w, a.w_gs, n, a.wdiv);
  const float* bn = wset_ptr(a.bias, a.b_gs, n, a.wdiv);

  // this lane's four halo positions: row 2 wave + nn / 16, columns 4 (nn % 16) ..; one 16-byte group, inside or outside the image
  const int gy = oy0 - 1 + 2 * wave + (nn >> 4), gx = ox0 - Sh::X0 + 4 * (nn & 15);
  const bool inimg = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
  const unsigned voff = inimg ? (unsigned)(8 * kk) * HW4 + (unsigned)(gy * a.W + gx) * 4u : 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrs = sm_rsrc(a.x + (size_t)n * 64 * a.H * a.W);
  f32x4 B[2][16];   // [half of the K steps in flight][step]
  auto issue = [&](int half) __attribute__((always_inline)) {   // K steps 16 half .. + 15: chunks 2 half, 2 half + 1
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int c = 16 * (2 * half + (s >> 3)) + (s & 7);       // (+ 8 kk: in voff)
      B[half][s] = sm_load16(xrs, voff, (unsigned)c * HW4);
    }
  };
  issue(0);
  // the weight matrix, once: s_a[m][c] = w[o][c][tap]
  for (int i = tid; i < 32 * 64; i += 256) {
    const int m = i >> 6, c = i & 63;
    const int tap = m / COUT, o = m - tap * COUT;
    s_a[m * AP + c] = m < NM ? wn[((size_t)o * 64 + c) * 9 + tap] : 0.f;
  }
  issue(1);
  __syncthreads();
  f32x4 A[8];   // A[2 ch + h] = weights of row nn for channels 16 ch + 8 kk + 4 h ..
#pragma unroll
  for (int q = 0; q < 8; ++q) A[q] = *reinterpret_cast<const f32x4*>(s_a + nn * AP + 16 * (q >> 1) + 8 * kk + 4 * (q & 1));

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int ch = 2 * half + (s >> 3), i = s & 7;
      const float av = A[2 * ch + (i >> 2)][i & 3];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, B[half][s][j], acc[j], 0, 0, 0);
    }

  // P rows of this lane: m = (r & 3) + 8 (r >> 2) + 4 kk; positions 128 wave + 4 nn + j
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * kk;
    if (m < NM)
      *reinterpret_cast<f32x4*>(s_p + m * NPOS + 128 * wave + 4 * nn) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
  }
  __syncthreads();

  // shift-add: thread = two horizontally adjacent outputs
  if (tid >= TH * (TW / 2)) return;
  const int y = tid / (TW / 2), xp = (tid - y * (TW / 2)) * 2;
  const int oy = oy0 + y, ox = ox0 + xp;
  if (oy >= a.H || ox >= a.W) return;
  float s0[COUT], s1[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) { s0[o] = 0.f; s1[o] = 0.f; }
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const float* row = s_p + (y + dy) * RW + xp + Sh::X0 - 1;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        const float* pm = row + ((dy * 3 + dx) * COUT + o) * NPOS + dx;
        s0[o] += pm[0];
        s1[o] += pm[1];
      }
  }
  const size_t HW = (size_t)a.H * a.W;
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    const float bv = bn ? bn[o] : 0.f;
    f32x2 v = {apply_act(s0[o] + bv, a.act), apply_act(s1[o] + bv, a.act)};
    const size_t idx = ((size_t)n * a.Cout + o) * HW + (size_t)oy * a.W + ox;   // (ox even, W % 4 == 0: an aligned pair)
    if (a.res) v += *reinterpret_cast<const f32x2*>(a.res + idx);
    *reinterpret_cast<f32x2*>(a.y + idx) = v;
  }
}

template <int COUT>
static int launch_small_mfma(SmallK k, hipStream_t st) {
  using Sh = SmallMfmaShape;
  k.tiles_x = ceil_div(k.W, Sh::TW); k.tiles_y = ceil_div(k.H, Sh::TH); k.ntiles = k.tiles_x * k.tiles_y * k.N;
  auto kern = conv3x3_small_cout_mfma_kernel<COUT>;
  static PerDeviceOnce once;
  set_dyn_lds_once(once, (const void*)kern, Sh::LDS_BYTES);
  hipLaunchKernelGGL(kern, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), Sh::LDS_BYTES, st, k);
  return check_launch("conv3x3_small_cout_mfma_kernel");
}

// x [N][C][H][W] (dense), w [Cout][C][3][3], Cout <= 4, stride 1, pad 1.
int conv3x3_small_cout_run(const float* x, const float* w, const float* bias, const float* res, float* y, int N,
                           int C, int H, int W, int Cout, int act, hipStream_t st, int wdiv, long long w_gs, int b_gs) {
  DVSR_REQUIRE(x && w && y && Cout >= 1 && Cout <= 4, DVSR_ERR_INVALID, "conv3x3_small_cout: bad argument");
  SmallK k;
  k.x = x; k.w = w; k.bias = bias; k.res = res; k.y = y; k.N = N; k.C = C; k.H = H; k.W = W; k.Cout = Cout; k.act = act;
  k.wdiv = wdiv > 0 ? wdiv : 1; k.w_gs = w_gs; k.b_gs = b_gs;
  // the matrix-pipe form: 64 input channels, up to 3 outputs, rows of whole 16-byte groups, one sample below 2 GB
  // (DVSR_CONV_LAST_MFMA=1 takes it: measured no faster than the VALU kernel, see above; read once per process)
  static const bool mfma_on = [] { const char* v = getenv("DVSR_CONV_LAST_MFMA"); return v && v[0] == '1'; }();
  if (mfma_on && C == 64 && Cout <= 3 && W % 4 == 0 && (unsigned long long)C * H * W * 4 < 0x7fffffffull &&
      ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0) && w_gs % 4 == 0) {
    if (Cout == 1) return launch_small_mfma<1>(k, st);
    if (Cout == 2) return launch_small_mfma<2>(k, st);
    return launch_small_mfma<3>(k, st);
  }
  k.tiles_x = ceil_div(W, 64); k.tiles_y = ceil_div(H, 8); k.ntiles = k.tiles_x * k.tiles_y * N;
  auto wbytes = [&](int cout) { return (size_t)ceil_div(C, 8) * 8 * (((9 * cout + 3) / 4) * 4) * sizeof(float); };
  switch (Cout) {  // exact channel count: no wasted accumulators
    case 1: hipLaunchKernelGGL(conv3x3_small_cout_kernel<1>, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), wbytes(1), st, k); break;
    case 2: hipLaunchKernelGGL(conv3x3_small_cout_kernel<2>, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), wbytes(2), st, k); break;
    case 3: hipLaunchKernelGGL(conv3x3_small_cout_kernel<3>, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), wbytes(3), st, k); break;
    default: hipLaunchKernelGGL(conv3x3_small_cout_kernel<4>, dim3(8 * ceil_div(k.ntiles, 8)), dim3(256), wbytes(4), st, k); break;
  }
  return check_launch("conv3x3_small_cout_kernel");
}

}  // namespace dvsr

extern "C" int dvsr_conv3x3_small_cout(const float* x, const float* w, const float* bias, const float* res, float* y, int N,
                                       int C, int H, int W, int Cout, int act, dvsr_stream_t stream) {
  DVSR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, DVSR_ERR_INVALID, "conv3x3_small_cout: empty tensor");
  return dvsr::conv3x3_small_cout_run(x, w, bias, res, y, N, C, H, W, Cout, act, (hipStream_t)stream, 1, 0, 0);
}
