// 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) with the sixteen transformed-domain GEMMs on the bf16 matrix
// pipe at fp32 accuracy: every fp32 operand (transformed weight U, transformed input V) is split EXACTLY into three bf16
// pieces, x = hi + mid + lo (8 + 8 + 8 significand bits), and the six partial products above 2^-24,
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid,
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (the arithmetic of the direct kernels' bf16_mfma = 2 mode, DESIGN
// 3.1b; EDVR_arch.py:254-313 is what is being computed).  Why: conv2d_wino.hip runs the same GEMMs on v_mfma_f32_32x32x2_f32,
// which IS the fp32 vector datapath -- nothing (input transform, LDS traffic, DMA issue) overlaps it, and the kernel stays
// at half of the pipe (r03: 49 % busy).  The bf16 pipe retires the six products of 8 channels x 32 x 32 in 3 x 32 cycles
// instead of 4 x 64, and VALU / LDS instructions issue beside it.
//
// FOUR FORMS of the kernel body share the prologue below (template parameter BLK, DVSR_CONV_WINO3_BLK; DESIGN 3.1f has the
// measurements): 0 -- the first one, described next; 1 -- one xn per wave and phase with 2 x 2 MFMA blocks (every operand
// fragment feeds two MFMAs; the epilogue becomes an all-to-all through the LDS); 2 -- 1 + the U fragments straight from global
// memory into a second register set (no other wave reads them); 3, THE DEFAULT -- 2 + the V images of two chunks in the LDS U
// no longer needs and one barrier per chunk.
//
// Work split of form 0 (one workgroup = 8 waves = one CU):
//   * workgroup tile = 64 couts x 64 tiles of 2x2 output pixels (TC tile columns: 4x64 or 8x32 pixels), K loop over chunks
//     of 8 input channels; a chunk is processed as TWO phases p = 0, 1 = the transformed-patch rows xi in {2p, 2p+1} (8 of
//     the 16 xn): the U and V images of a phase are 24 KB each, so U (three buffers: fetched two phases ahead), V (two) and
//     the raw halo (two chunks) fit the CU's LDS: 3 x 24 + 2 x 24 + 2 x 13.5 KB = 147 KB;
//   * MFMA role of wave (mh, tr, xq): 32 couts x 32 tiles x the FOUR xn of row xi = 2p + xq in phase p (acc[4 p + nu]);
//     per xn three MFMAs with K = 16 = 8 channels x 2 pieces: lanes 0-31 / 32-63 carry
//         A1 = (hi | hi), B1 = (hi | mid);   A2 = (mid | hi), B2 = (hi | lo);   A3 = (lo | mid), B3 = B1;
//   * transform role of wave (q = wave & 3, r = wave >> 2), lane = tile: the channel PAIR (2q, 2q+1) of the chunk, row
//     xi = 2p + r of B^T d B (each row of B^T d combines exactly two raw rows): 12 ds_read_b64, 14 adds, then per xn the
//     two channels' values are split together (v_cvt_pk_bf16_f32 packs the pair) and written as 4-byte words into
//     V[piece][xl][qh][tile][pair] -- conflict-free 8-byte operand reads (4 channels per read), 2-way (free) on the writes;
//   * U: pack_weights_wino3_kernel computes G g G^T in fp32 and stores the three pieces as the exact LDS image of a phase,
//     [piece][xl][cout][8 channels] bf16 = 16-byte records read by ds_read_b128; staged by 16-byte buffer-load DMA.
// One barrier per phase, in front of the last xn's MFMAs (as conv2d_wino.hip); the waits are counted by hand: the DMA of
// U(s+2) stays in flight across the barrier of phase s.
// Epilogue: as conv2d_wino.hip (each wave reduces its two rows of M to a partial 2x2 output, the two waves of a pair swap
// halves through LDS), with rows {xq, xq + 2} per wave instead of {2 xh, 2 xh + 1}.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

#ifdef DVSR_CONV_TRACE
#define W3_ABLATE(a) ((a).ablate)
#define W3_STAMP(i)                                                                                       \
  do {                                                                                                    \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
// stamps INSIDE one phase (s == 6), by lane 0 of waves 0 and 4 (the two waves of one SIMD): slots 20 + i / 30 + i
#define W3_FINE(i)                                                                                                      \
  do {                                                                                                                  \
    if (a.trace && s == 6 && lane == 0 && (wave & 3) == 0)                                                              \
      a.trace[(size_t)blockIdx.x * 64 + (wave ? 30 : 20) + (i)] = __builtin_readcyclecounter();                          \
  } while (0)
#else
#define W3_ABLATE(a) 0
#define W3_STAMP(i) \
  do {              \
  } while (0)
#define W3_FINE(i) \
  do {             \
  } while (0)
#endif

typedef float w3f2 __attribute__((ext_vector_type(2)));
typedef __bf16 w3bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 w3bf2 __attribute__((ext_vector_type(2)));

// P16[cb][k][p][piece][xl][cout 64][slot 8] = piece of (G g G^T)[xi = 2p + (xl >> 2)][nu = xl & 3] of (cout = cb*64 + col,
// cin = k*8 + slot).  One thread = one (cout, cin) pair, as pack_weights_wino_kernel.
__global__ void pack_weights_wino3_kernel(PackTable t) {
  const PackEntry& e = t.e[blockIdx.y];
  if (e.perm != 4) return;
  __bf16* const P16 = reinterpret_cast<__bf16*>(e.P);
  const size_t total = (size_t)e.ncb * e.nchunks * 512;   // (cout, cin) pairs incl. padding
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i & 7), col = (int)((i >> 3) & 63);
    const size_t ck = i >> 9;
    const int k = (int)(ck % e.nchunks), cb = (int)(ck / e.nchunks);
    const int co = cb * 64 + col, ci = k * 8 + c8;
    float g[3][3];
    const bool ok = co < e.Cout && ci < e.Ctot;
    const float* src = !e.wt ? e.w + ((size_t)co * e.Ctot + ci) * 9 : e.w + ((size_t)ci * e.w_ctot + e.w_coff + co) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float v = ok ? src[e.wt ? 8 - tap : tap] : 0.f;
      g[tap / 3][tap % 3] = v;
    }
    float c[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      c[0][b] = g[0][b];
      c[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      c[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      c[3][b] = g[2][b];
    }
    __bf16* dst = P16 + ck * 24576 + (size_t)col * 8 + c8;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      const float u[4] = {c[xi][0], 0.5f * (c[xi][0] + c[xi][1] + c[xi][2]), 0.5f * (c[xi][0] - c[xi][1] + c[xi][2]), c[xi][2]};
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        const int p = xi >> 1, xl = (xi & 1) * 4 + nu;
        const __bf16 h = (__bf16)u[nu];
        const float r1 = u[nu] - (float)h;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        __bf16* d = dst + (size_t)p * 12288 + (size_t)xl * 512;
        d[0] = h;
        d[4096] = m;
        d[8192] = l;
      }
    }
  }
}

int pack_weights_wino3_run(const PackTable& t, hipStream_t st) {
  hipLaunchKernelGGL(pack_weights_wino3_kernel, dim3(64, t.n), dim3(256), 0, st, t);
  return check_launch("pack_weights_wino3_kernel");
}

__device__ __forceinline__ void w3_dma16(const float* base, float* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000),
                                           (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ unsigned w3_lds_addr(const float* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const float*)p);
}
__device__ __forceinline__ void w3_read_b64(w3f2& dst, unsigned addr, int off) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off));
}
// one lgkmcnt(0) for a set of asm reads: every register of the set is "modified" so that no use can move above the wait
template <bool WAIT, typename T, int N0, int N1, int N2>
__device__ __forceinline__ void w3_lgkm_wait(T (&r)[N0][N1][N2]) {
  if (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N0; ++i)
#pragma unroll
    for (int j = 0; j < N1; ++j)
#pragma unroll
      for (int k = 0; k < N2; ++k) asm volatile("" : "+v"(r[i][j][k]));
}
// v_cvt_pk_bf16_f32: {bf16(a) (round to nearest even) in bits 15:0, bf16(b) in bits 31:16}
__device__ __forceinline__ unsigned w3_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int TC>
struct Wino3Shape {
  static constexpr int CC = 8, NTILE = 64, TRW = NTILE / TC;
  static constexpr int OH = 2 * TRW, OW = 2 * TC;      // output pixels of the workgroup tile
  static constexpr int IH = OH + 2, RP = OW + 8, GR = RP / 4;
  static constexpr int NG = CC * IH * GR;              // 16-byte groups of one chunk's raw halo image
  static constexpr int NI = (NG + 511) / 512;
  static constexpr int RAW_FLOATS = NG * 4;
  static constexpr int SUB = 6144;                     // floats (24 KB) of one phase's U (or V) image
  // LDS (floats): V0 | raw0 | raw1 | U0 | U1 | U2 | V1
  static constexpr int OFF_V0 = 0, OFF_R = SUB, OFF_U = SUB + 2 * RAW_FLOATS, OFF_V1 = OFF_U + 3 * SUB;
  static constexpr size_t LDS_BYTES = (size_t)(OFF_V1 + SUB) * sizeof(float);
};

// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14])
#define W3_WAIT_VM3_LGKM0 0x0073
#define W3_WAIT_VM0_LGKM0 0x0070

template <int TC, int BLK>
__global__ __launch_bounds__(512, 2) void conv2d_wino3_kernel(ConvK2 a) {
  using Sh = Wino3Shape<TC>;
  constexpr int IH = Sh::IH, RP = Sh::RP, GR = Sh::GR, NI = Sh::NI, SUB = Sh::SUB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_r0 = smem + Sh::OFF_R;
  float* const s_ub = smem + Sh::OFF_U;

  const int id = blockIdx.x;
  const int q_ = id >> 3;  // XCD-aware order, as conv2d_pipe_item
  const int cbi = q_ % a.ncb;
  const int j_ = q_ / a.ncb;
  const int tile = (id & 7) * a.tiles_per_xcd + j_;
  if (j_ >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * Sh::OH, ox0 = tx_ * Sh::OW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int mh = wave & 1, tr = (wave >> 1) & 1, xq = wave >> 2;      // MFMA role
  const int tq = wave & 3, trr = wave >> 2;                            // transform role: channel pair, row of the phase
  const size_t HW = (size_t)a.H * a.W;
  const float* x0n = a.x0 + (size_t)n * a.x0_bs;
  const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : x0n;
  const int nph = 2 * a.nchunks;

  // raw halo groups this lane moves: group L = 64 * (wave + 8 jj) + lane = (channel, row, column group)
  unsigned hoff[NI];
  bool hval[NI];
#pragma unroll
  for (int jj = 0; jj < NI; ++jj) {
    const int L = 64 * (wave + 8 * jj) + lane;
    const int c = L / (IH * GR), r = L - c * (IH * GR);
    const int iy = r / GR, g = r - iy * GR;
    const int gy = oy0 - 1 + iy, gx = ox0 - 4 + 4 * g;
    const bool ok = L < Sh::NG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    hoff[jj] = ok ? (unsigned)(((size_t)c * HW + (size_t)gy * a.W + gx) * 4) : 0u;
    hval[jj] = ok;
    if (L < Sh::NG && !ok) {
      *reinterpret_cast<f32x4*>(s_r0 + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(s_r0 + Sh::RAW_FLOATS + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  f32x16 acc[8];   // acc[4 p + nu] = M[xi = 2 p + xq][nu] (first written by the MFMAs of chunk 0)

  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + (size_t)cbi * a.nchunks * (2 * SUB);

  const unsigned uoff = (unsigned)(lane * 16 + wave * 1024);
  const unsigned chunk_bytes = (unsigned)(Sh::CC * HW * 4);
  auto issue_raw = [&](int k) {
    const int cbase = k * Sh::CC;
    const bool second = cbase >= a.c0;  // only possible when c1 > 0; a chunk never straddles the two inputs
    const unsigned soff = (unsigned)((second ? k - a.c0 / Sh::CC : k)) * chunk_bytes;
    float* dst = s_r0 + (k & 1) * Sh::RAW_FLOATS;
#pragma unroll
    for (int jj = 0; jj < NI; ++jj)
      if (hval[jj]) {
        if (second)
          w3_dma16(x1n, dst + 256 * (wave + 8 * jj), hoff[jj], soff);
        else
          w3_dma16(x0n, dst + 256 * (wave + 8 * jj), hoff[jj], soff);
      }
  };
  auto issue_u_piece = [&](int s, int ub, int j) {   // piece j (8 KB) of phase image s (24 KB) -> U buffer ub
    w3_dma16(wp_cb, s_ub + ub * SUB + (j * 8 + wave) * 256, uoff, (unsigned)(s * (SUB * 4) + j * 8192));
  };
  auto issue_u = [&](int s, int ub) {
#pragma unroll
    for (int j = 0; j < 3; ++j) issue_u_piece(s, ub, j);
  };

  // ---- input transform of one phase: lane = tile, channels 2 tq and 2 tq + 1, row xi = 2 p + trr
  const int trow_t = lane / TC, tcol_t = lane - trow_t * TC;
  const int roff = (2 * tq * IH + 2 * trow_t) * RP + 2 * tcol_t + 2;
  const int vwoff = (trr * 8 + (tq >> 1)) * 128 + lane * 2 + (tq & 1);   // + piece * 2048 + nu * 256
  float tdl[2][2], tdr[2][2];   // [channel][raw row A / B]: columns +3 and +6 of the patch row
  w3f2 tdm[2][2];               // columns +4, +5
  float vv[2][4];
  auto tf_load_c = [&](auto p_, int rbuf, int c) __attribute__((always_inline)) {   // the two raw rows of channel 2 tq + c
    constexpr int P = decltype(p_)::value;
    const int rowA = P == 0 ? trr : (trr ? 1 : 2), rowB = P == 0 ? 2 : (trr ? 3 : 1);
    // (inline asm: the compiler would merge neighbours into ds_read2 forms and keep six registers per row; the results are
    // waited for by the lgkmcnt(0) in front of their first use)
    const unsigned aA = w3_lds_addr(s_r0 + rbuf * Sh::RAW_FLOATS + roff + rowA * RP);
    const unsigned aB = w3_lds_addr(s_r0 + rbuf * Sh::RAW_FLOATS + roff + rowB * RP);
    if (c == 0) {
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tdl[0][0]) : "v"(aA), "i"((1) * 4));
      w3_read_b64(tdm[0][0], aA, (2) * 4);
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tdr[0][0]) : "v"(aA), "i"((4) * 4));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tdl[0][1]) : "v"(aB), "i"((1) * 4));
      w3_read_b64(tdm[0][1], aB, (2) * 4);
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tdr[0][1]) : "v"(aB), "i"((4) * 4));
    } else {
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tdl[1][0]) : "v"(aA), "i"((IH * RP + 1) * 4));
      w3_read_b64(tdm[1][0], aA, (IH * RP + 2) * 4);
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tdr[1][0]) : "v"(aA), "i"((IH * RP + 4) * 4));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tdl[1][1]) : "v"(aB), "i"((IH * RP + 1) * 4));
      w3_read_b64(tdm[1][1], aB, (IH * RP + 2) * 4);
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tdr[1][1]) : "v"(aB), "i"((IH * RP + 4) * 4));
    }
  };
  auto tf_load = [&](auto p_, int rbuf) __attribute__((always_inline)) {
    tf_load_c(p_, rbuf, 0);
    tf_load_c(p_, rbuf, 1);
  };
  auto tf_rows_cols = [&](auto p_, bool wait) __attribute__((always_inline)) {
    constexpr int P = decltype(p_)::value;
    // rows of B^T d: xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const float sg = (P == 0 && trr) ? 1.f : -1.f;
    if (wait) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int r = 0; r < 2; ++r) { asm volatile("" : "+v"(tdl[c][r])); asm volatile("" : "+v"(tdm[c][r])); asm volatile("" : "+v"(tdr[c][r])); }
      const float c0 = tdl[c][0] + sg * tdl[c][1], c3 = tdr[c][0] + sg * tdr[c][1];
      const w3f2 cm = tdm[c][0] + sg * tdm[c][1];
      const float c1 = cm[0], c2 = cm[1];
      vv[c][0] = c0 - c2;
      vv[c][1] = c1 + c2;
      vv[c][2] = c2 - c1;
      vv[c][3] = c1 - c3;
    }
  };
  auto tf_split = [&](int nu, float* vdst) {   // three exact bf16 pieces of the channel pair -> V words
    const float x0 = vv[0][nu], x1 = vv[1][nu];
    const unsigned h = w3_cvt_pk(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = w3_cvt_pk(r0, r1);
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    const unsigned l = w3_cvt_pk(q0, q1);
    float* d = vdst + vwoff + nu * 256;
    d[0] = __builtin_bit_cast(float, h);
    d[2048] = __builtin_bit_cast(float, m);
    d[4096] = __builtin_bit_cast(float, l);
  };

  if constexpr (BLK == 0) {
    // bias: every output of a tile receives M[1][1] with weight one, so the bias is the INITIAL value of xn = 5
    // (xi = 1: phase 0 of the xq = 1 waves, nu = 1); the other accumulators start from the MFMA's inline zero
    const int co_block = cbi * 64 + mh * 32;
    f32x16 cinit5;
    {
      const float* bias = wset_ptr(a.bias, a.b_gs, n, a.wdiv);
  #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_block + (r & 3) + 8 * (r >> 2) + 4 * hi;
        cinit5[r] = (bias && xq == 1) ? bias[co < a.Cout ? co : a.Cout - 1] : 0.f;
      }
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // operand addresses (floats).  A: U[piece][xl][cout] x 16 B; B: V[piece][xl][qh][tile] x 8 B.  Lane halves read the
    // pieces (A1: hi|hi, A2: mid|hi, A3: lo|mid; B1 = B3: hi|mid, B2: hi|lo).
    const int arow = (4 * xq * 64 + mh * 32 + lo) * 4;
    const int abase0 = arow, abase1 = (hi ? 0 : 1) * 2048 + arow, abase2 = (hi ? 1 : 2) * 2048 + arow;
    const int brow = (4 * xq * 128 + tr * 32 + lo) * 2;
    const int bbase0 = (hi ? 1 : 0) * 2048 + brow, bbase1 = (hi ? 2 : 0) * 2048 + brow;
    f32x4 A[2][3];      // [xn of the pair][A1 / A2 / A3]
    f32x4 B[2][2];      // [xn of the pair][B1 / B2] (qh = 0 | 1 halves)
    // All operand reads are inline asm (single ds_read_b128 / ds_read_b64) into ONE register set that is recycled inside the
    // MFMA block: as soon as the two MFMAs of a product index j have been issued, the registers they read are reloaded with
    // the next xn pair's operands (the MFMA has read its sources long before the LDS returns), so the LDS traffic of a pair
    // runs under the MFMAs of the pair before it; ONE lgkmcnt(0) at the top of the next block waits for it.
    auto load_A = [&](unsigned au, int i, int e, int j) {
      const int ab = j == 0 ? abase0 : (j == 1 ? abase1 : abase2);
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[e][j]) : "v"(au + ab * 4), "i"(i * 1024));
    };
    auto load_B = [&](unsigned av, int i, int e, int m) {
      const int bb = m ? bbase1 : bbase0;
      // the two 8-byte halves (qh = 0, 1: 512 bytes apart) in one instruction: offsets in units of 64 x 8 bytes
      asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(B[e][m]) : "v"(av + bb * 4), "i"(2 * i), "i"(2 * i + 1));
    };
    auto load_pair = [&](const float* s_u, const float* s_v, int h) {
      const unsigned au = w3_lds_addr(s_u), av = w3_lds_addr(s_v);
  #pragma unroll
      for (int e = 0; e < 2; ++e) {
  #pragma unroll
        for (int j = 0; j < 3; ++j) load_A(au, 2 * h + e, e, j);
        load_B(av, 2 * h + e, e, 0);
        load_B(av, 2 * h + e, e, 1);
      }
    };
    auto wait_ops = [&](bool wait) __attribute__((always_inline)) {
      if (wait) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  #pragma unroll
      for (int e = 0; e < 2; ++e) {
  #pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(A[e][j]));
  #pragma unroll
        for (int m = 0; m < 2; ++m) asm volatile("" : "+v"(B[e][m]));
      }
    };

    // One phase:  M0 (six MFMAs of the first xn pair, the second pair's operand loads between them) -- V (the whole input
    // transform of the NEXT phase: its raw rows were read at the end of the previous phase) -- barrier -- M1 (second pair, the
    // next phase's first operands between the MFMAs) -- tail (raw reads for the phase after next, DMA issue: U three phases
    // ahead into the buffer this phase has just finished with, the raw halo two chunks ahead).
    int ub = 0;   // U buffer of the current phase (s % 3)
    auto phase = [&](int s, auto p_, auto first_, auto next_) __attribute__((always_inline)) {
      constexpr int P = decltype(p_)::value;
      constexpr bool FIRST = decltype(first_)::value, HAS_NEXT = decltype(next_)::value;
      const float* s_u = s_ub + ub * SUB;
      const float* s_v = smem + (P ? Sh::OFF_V1 : Sh::OFF_V0);
      float* const v_next = smem + (P ? Sh::OFF_V0 : Sh::OFF_V1);
      const int ub1 = ub == 2 ? 0 : ub + 1;
      // six MFMAs of pair h; nu / nv = LDS addresses of the U / V image the NEXT pair (xn 2 hn, 2 hn + 1) is read from.
      // `fill(j)` runs behind the two MFMAs of product j (and the reloads of their registers): everything else the phase has
      // to issue is cut into three such pieces per block, so that no long stretch of non-MFMA instructions is left anywhere
      // (tools/wino_trace.py: issued in one go behind the blocks, the raw reads + DMA of the tail took 640 cycles of a
      // 2500-cycle phase, an LDS-DMA instruction alone costs 60 - 180 cycles of issue).
      auto blkM = [&](auto h_, auto ld_, unsigned nu, unsigned nv, int hn, auto&& fill) __attribute__((always_inline)) {
        constexpr int h = decltype(h_)::value;
        constexpr bool ld = decltype(ld_)::value;
        static_for<0, 3>([&](auto j_) {
          constexpr int j = decltype(j_)::value;
  #pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int i = 2 * h + e;
            const w3bf8 av = __builtin_bit_cast(w3bf8, A[e][j]);
            const int m = j == 1 ? 1 : 0;
            const w3bf8 bv = __builtin_bit_cast(w3bf8, B[e][m]);
            if (FIRST && j == 0)
              acc[4 * P + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, (P == 0 && i == 1) ? cinit5 : zero16, 0, 0, 0);
            else
              acc[4 * P + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[4 * P + i], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (ld && !(W3_ABLATE(a) & 8)) {   // the registers of product j are free: A_j of both xn; B2 after j = 1, B1 after j = 2
  #pragma unroll
            for (int e = 0; e < 2; ++e) {
              load_A(nu, 2 * hn + e, e, j);
              if (j == 1) load_B(nv, 2 * hn + e, e, 1);
              if (j == 2) load_B(nv, 2 * hn + e, e, 0);
            }
          }
          fill(j_);
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      using H0 = std::integral_constant<int, 0>;
      using H1 = std::integral_constant<int, 1>;
      using PN = std::integral_constant<int, P ^ 1>;
      W3_FINE(0);
      wait_ops(true);
      W3_FINE(1);
      // M0 + the loads of pair 1 + the NEXT phase's input transform (its raw rows were read in the previous phase's M1)
      blkM(H0{}, std::true_type{}, w3_lds_addr(s_u), w3_lds_addr(s_v), 1, [&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        if (HAS_NEXT && !(W3_ABLATE(a) & 2)) {
          if (j == 0) tf_rows_cols(PN{}, false);
          if (j == 1) { tf_split(0, v_next); tf_split(1, v_next); }
          if (j == 2) { tf_split(2, v_next); tf_split(3, v_next); }
        }
      });
      W3_FINE(2);
      W3_FINE(3);
      if (HAS_NEXT) {
        if (!(W3_ABLATE(a) & 16)) {
          if (s + 2 < nph) __builtin_amdgcn_s_waitcnt(W3_WAIT_VM3_LGKM0);   // U(s+1) and the raw halo landed; U(s+2) may still fly
          else __builtin_amdgcn_s_waitcnt(W3_WAIT_VM0_LGKM0);
          W3_FINE(4);
          __builtin_amdgcn_s_barrier();
        }
        W3_FINE(5);
        wait_ops(false);
        // M1 + the next phase's first operands + the raw reads of the transform after next + the DMA issue (U three phases
        // ahead into the buffer this phase has finished with, the raw halo two chunks ahead)
        const int rb2 = ((s + 2) >> 1) & 1;
        blkM(H1{}, std::true_type{}, w3_lds_addr(s_ub + ub1 * SUB), w3_lds_addr(v_next), 0, [&](auto j_) __attribute__((always_inline)) {
          constexpr int j = decltype(j_)::value;
          if (s + 2 < nph && !(W3_ABLATE(a) & 2)) {
            if (j == 0) tf_load_c(std::integral_constant<int, P>{}, rb2, 0);
            if (j == 1) tf_load_c(std::integral_constant<int, P>{}, rb2, 1);
          }
          if (!(W3_ABLATE(a) & 4)) {
            if (s + 3 < nph) issue_u_piece(s + 3, ub, j);
            if (j == 2 && P == 0 && (s >> 1) + 2 < a.nchunks) issue_raw((s >> 1) + 2);
          }
        });
        W3_FINE(6);
        W3_FINE(7);
      } else {
        wait_ops(true);
        blkM(H1{}, std::false_type{}, 0u, 0u, 0, [&](auto) __attribute__((always_inline)) {});
      }
      __builtin_amdgcn_sched_barrier(0);
      ub = ub1;
    };

    W3_STAMP(0);
  #ifdef DVSR_CONV_TRACE
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memrealtime();
  #endif
    issue_u(0, 0);
    issue_raw(0);
    issue_u(1, 1);
    issue_raw(1);   // (at least two chunks: conv2d_packed_prepare)
    issue_u(2, 2);
    __syncthreads();
    W3_STAMP(1);
    tf_load(std::integral_constant<int, 0>{}, 0);
    tf_rows_cols(std::integral_constant<int, 0>{}, true);
    tf_split(0, smem + Sh::OFF_V0); tf_split(1, smem + Sh::OFF_V0); tf_split(2, smem + Sh::OFF_V0); tf_split(3, smem + Sh::OFF_V0);
    __syncthreads();
    load_pair(s_ub, smem + Sh::OFF_V0, 0);
    tf_load(std::integral_constant<int, 1>{}, 0);
    W3_STAMP(2);
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;
    phase(0, P0{}, T{}, T{});
    phase(1, P1{}, T{}, T{});
    W3_STAMP(3);
    for (int k = 1; k + 1 < a.nchunks; ++k) {
      phase(2 * k, P0{}, F{}, T{});
      phase(2 * k + 1, P1{}, F{}, T{});
      if (k < 30) W3_STAMP(3 + k);
    }
    phase(nph - 2, P0{}, F{}, T{});
    phase(nph - 1, P1{}, F{}, F{});
    W3_STAMP(40);

    // ---- epilogue.  Y = A^T M A is linear in the rows of M: this wave reduces ITS two rows (xi = xq and xq + 2) to a partial
    // 2x2 output per (cout, tile), the two waves of a pair swap halves through LDS (the last phase reads only U[(nph-1) % 3]
    // and V1: V0, the raw buffers and the two other U buffers are idle) and each finishes 8 of the 16 cout registers.
    // (the lane index passes through an opaque asm: nothing of the epilogue's per-lane addressing can be hoisted above the
    // K loop, where every register is spoken for)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lo_e = lane_e & 31, hi_e = lane_e >> 5;
    const int ttw = tr * 32 + lo_e;                      // this lane's tile
    const int orow = oy0 + 2 * (ttw / TC), ocol = ox0 + 2 * (ttw % TC);
    const size_t HWo = (size_t)a.Ho * a.Wo;
    const float slope = a.act == ACT_LRELU ? 0.1f : (a.act == ACT_RELU ? 0.f : 1.f);
    const float neg = a.gmask_act == ACT_LRELU ? 0.1f : (a.gmask_act == ACT_RELU ? 0.f : 1.f);
    const bool full = oy0 + Sh::OH <= a.Ho && ox0 + Sh::OW <= a.Wo && cbi * 64 + 64 <= a.Cout;
    const int q = mh + 2 * tr;   // the pair
    // exchange area of pair q (16 KB each): V0 | raw0 + raw1 | the two idle U buffers (ub is now nph % 3 = the one after the last)
    const int ubl = ub == 0 ? 2 : ub - 1;   // buffer the last phase read
    const int uf0 = ubl == 0 ? 1 : 0, uf1 = ubl == 2 ? 1 : 2;
    float* const xarea = q == 0 ? smem + Sh::OFF_V0 : (q == 1 ? s_r0 : s_ub + (q == 2 ? uf0 : uf1) * SUB);
    float* const xch = xarea + lane_e * 4;   // slot [receiving half][rr][lane]
    const char* const ybase = reinterpret_cast<const char*>(a.y + ((size_t)n * a.Cout + co_block) * HWo);
    const unsigned lane_off = (unsigned)(((size_t)(4 * hi_e) * HWo + (size_t)orow * a.Wo + ocol) * 4);
    auto finish = [&](auto xq_) {   // (one instantiation per half: register indices stay compile-time constants)
      constexpr int XH = decltype(xq_)::value;
      // pp[P][e] = (y00, y01, y10, y11) of the register PAIR (2P, 2P + 1) = output channels (co, co + 1), as packed pairs
      w3f2 pp[8][4];
  #pragma unroll
      for (int P = 0; P < 8; ++P) {
        w3f2 s0[4], s1[4];
  #pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          const w3f2 ma = {acc[nu][2 * P], acc[nu][2 * P + 1]}, mb = {acc[4 + nu][2 * P], acc[4 + nu][2 * P + 1]};
          // XH = 0: rows 0, 2 of M: Y0 += M0 + M2, Y1 += -M2;  XH = 1: rows 1, 3: Y0 += M1, Y1 += M1 - M3
          s0[nu] = XH == 0 ? ma + mb : ma;
          s1[nu] = XH == 0 ? mb : ma - mb;
        }
        pp[P][0] = s0[0] + s0[1] + s0[2];
        pp[P][1] = s0[1] - s0[2] - s0[3];
        if (XH == 1) {
          pp[P][2] = s1[0] + s1[1] + s1[2];
          pp[P][3] = s1[1] - s1[2] - s1[3];
        } else {
          pp[P][2] = -s1[0] - s1[1] - s1[2];
          pp[P][3] = s1[2] + s1[3] - s1[1];
        }
      }
  #pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int P = (XH ^ 1) * 4 + q4;
        *reinterpret_cast<f32x4*>(xch + (((XH ^ 1) * 4 + q4) * 2 + 0) * 256) = f32x4{pp[P][0][0], pp[P][0][1], pp[P][1][0], pp[P][1][1]};
        *reinterpret_cast<f32x4*>(xch + (((XH ^ 1) * 4 + q4) * 2 + 1) * 256) = f32x4{pp[P][2][0], pp[P][2][1], pp[P][3][0], pp[P][3][1]};
      }
      const bool plain = !a.res && !a.accum && !a.gmask;
      w3f2 ex[4][2][2];
      if (full && !plain && a.ps == 0) {
  #pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int r = 8 * XH + 2 * q4;
          const int rc = (r & 3) + 8 * (r >> 2);
  #pragma unroll
          for (int c = 0; c < 2; ++c)
  #pragma unroll
            for (int i = 0; i < 2; ++i) {
              const size_t sb = (((size_t)n * a.Cout + co_block + rc + c) * HWo + (size_t)i * a.Wo) * 4;   // scalar
              w3f2 e = {0.f, 0.f};
              if (a.res) e = *reinterpret_cast<const w3f2*>(reinterpret_cast<const char*>(a.res) + sb + lane_off);
              if (a.accum) e += *reinterpret_cast<const w3f2*>(reinterpret_cast<const char*>(a.y) + sb + lane_off);
              ex[q4][c][i] = e;
            }
        }
      }
      __syncthreads();
      w3f2 o[4][4];   // own pairs, activated
  #pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int P = XH * 4 + q4;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(xch + ((XH * 4 + q4) * 2 + 0) * 256);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(xch + ((XH * 4 + q4) * 2 + 1) * 256);
        const w3f2 in[4] = {w3f2{r0[0], r0[1]}, w3f2{r0[2], r0[3]}, w3f2{r1[0], r1[1]}, w3f2{r1[2], r1[3]}};
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          const w3f2 v = pp[P][e] + in[e];
          o[q4][e] = __builtin_elementwise_max(v, v * slope);
        }
      }
      if (a.ps == 0) {
  #pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int r = 8 * XH + 2 * q4;                       // registers r, r + 1 = channels co, co + 1
          const int rc = (r & 3) + 8 * (r >> 2);               // channel of register r relative to co_block + 4 hi
  #pragma unroll
          for (int c = 0; c < 2; ++c) {
  #pragma unroll
            for (int i = 0; i < 2; ++i) {     // the two rows of the tile
              w3f2 v = {o[q4][2 * i][c], o[q4][2 * i + 1][c]};
              if (full) {
                const size_t sb = ((size_t)(rc + c) * HWo + (size_t)i * a.Wo) * 4;   // scalar
                float* dst = reinterpret_cast<float*>(const_cast<char*>(ybase) + sb + lane_off);
                if (!plain) {
                  v += ex[q4][c][i];
                  if (a.gmask) {   // (data-gradient launches: the activation mask of the producer, read late -- registers)
                    const size_t sg = (((size_t)n * a.Cout + co_block + rc + c) * HWo + (size_t)i * a.Wo) * 4;
                    const w3f2 m = *reinterpret_cast<const w3f2*>(reinterpret_cast<const char*>(a.gmask) + sg + lane_off);
                    v = w3f2{v[0] * (m[0] > 0.f ? 1.f : neg), v[1] * (m[1] > 0.f ? 1.f : neg)};
                  }
                }
                *reinterpret_cast<w3f2*>(dst) = v;
                continue;
              }
              const int co = co_block + rc + c + 4 * hi_e;
              const int oy = orow + i;
              const size_t idx = ((size_t)n * a.Cout + co) * HWo + (size_t)oy * a.Wo + ocol;
              const bool ok0 = co < a.Cout && oy < a.Ho && ocol < a.Wo;
              const bool ok1 = ok0 && ocol + 1 < a.Wo;
              if (!ok0) continue;
  #pragma unroll
              for (int j = 0; j < 2; ++j) {
                if (j == 1 && !ok1) continue;
                float w = v[j];
                if (a.res) w += a.res[idx + j];
                if (a.accum) w += a.y[idx + j];
                if (a.gmask) w *= a.gmask[idx + j] > 0.f ? 1.f : neg;
                a.y[idx + j] = w;
              }
            }
          }
        }
      } else {
        // PixelShuffle(2): channels 4 cq .. 4 cq + 3 (registers 4 g .. 4 g + 3) are the 2x2 sub-pixels (dy, dx) of channel cq
  #pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int co = co_block + 8 * (2 * XH + gg) + 4 * hi_e;
          const int cq = co >> 2;
          if (co >= a.Cout) continue;
  #pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int oy = orow + i;
            if (!full && (oy >= a.Ho || ocol >= a.Wo)) continue;
  #pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
              const w3f2 e0 = o[2 * gg + dy][2 * i], e1 = o[2 * gg + dy][2 * i + 1];
              const f32x4 v = f32x4{e0[0], e0[1], e1[0], e1[1]};
              float* dst = a.y + (((size_t)n * (a.Cout >> 2) + cq) * (2 * a.Ho) + (2 * oy + dy)) * (size_t)(2 * a.Wo) + 2 * ocol;
              if (full || ocol + 1 < a.Wo) *reinterpret_cast<f32x4*>(dst) = v;
              else *reinterpret_cast<w3f2*>(dst) = w3f2{v[0], v[1]};
            }
          }
        }
      }
    };
    if (xq == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
  } else {
    // ================= BLK == 1: ONE xn per wave and phase, all 64 couts x 64 tiles of it (2 x 2 MFMA blocks) =================
    // MFMA role of wave (xq = wave >> 2, nu = wave & 3): xn = (xi = 2 p + xq, nu) in phase p, acc[4 p + 2 mh + tr].  Every
    // operand fragment feeds TWO MFMAs instead of one: 10 KB of operand reads per wave and phase instead of 20 (the kernel is
    // bound by LDS bandwidth and issue).  The price is the epilogue: no wave holds a whole row of M any more, so
    // all sixteen xn meet in the LDS (two rounds of 128 KB, one per cout half) instead of half of them in registers.
    const int nu_w = wave & 3;
    const int xl_w = 4 * xq + nu_w;
    // bias of the couts this lane finishes in the epilogue (wave (tr_o, rq_o) = (wave & 1, wave >> 1): registers 4 rq_o .. + 3
    // of cout half R), loaded here: in the epilogue the latency would be exposed
    float bk[2][4];
    {
      const float* bias = wset_ptr(a.bias, a.b_gs, n, a.wdiv);
#pragma unroll
      for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int co = cbi * 64 + R * 32 + 8 * (wave >> 1) + 4 * hi + k;
          bk[R][k] = bias ? bias[co < a.Cout ? co : a.Cout - 1] : 0.f;
        }
    }
    // byte offsets inside a phase image.  A: U[piece][xl][cout] x 16 B (+ 512 for mh = 1); B: V[piece][xl][qh][tile] x 8 B
    // (ds_read2_b64: qh = 0 | 1 at offsets 0 | 64 and tr = 1 at + 32, in units of 8 bytes).  Lane halves read the pieces
    // (A1: hi|hi, A2: mid|hi, A3: lo|mid; B1 = B3: hi|mid, B2: hi|lo).
    const unsigned aoff0 = (unsigned)(xl_w * 1024 + lo * 16);
    const unsigned aoff1 = aoff0 + (hi ? 0u : 8192u), aoff2 = aoff0 + (hi ? 8192u : 16384u);
    const unsigned boff0 = (unsigned)(xl_w * 1024 + lo * 8) + (hi ? 8192u : 0u);
    const unsigned boff1 = (unsigned)(xl_w * 1024 + lo * 8) + (hi ? 16384u : 0u);
    // BLK == 2: no other wave reads this wave's U fragments, so they skip the LDS -- buffer loads straight into a second
    // register set, issued one phase ahead (the packed image IS the fragment layout: 16-byte records, 32 couts contiguous).
    constexpr bool GA = BLK >= 2;
    f32x4 A[GA ? 2 : 1][2][3];   // [set: phase parity (BLK == 2)][mh][A1 / A2 / A3]
    f32x4 B[2][2];               // [tr][B1 / B2]
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp_cb), 0, -1, 0x00020000);
    auto gldA = [&](int set, int s_) __attribute__((always_inline)) {   // the six A fragments of phase s_ -> register set `set`
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          A[set][m][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              wrsrc, (int)((j == 0 ? aoff0 : (j == 1 ? aoff1 : aoff2)) + m * 512), s_ * (SUB * 4), 0));
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // (non-generic lambdas: clang rejects asm operands that name captured arrays inside a generic lambda)
    auto ldA = [&](unsigned au, int mh_, int j_) __attribute__((always_inline)) {
      const unsigned ad = au + (j_ == 0 ? aoff0 : (j_ == 1 ? aoff1 : aoff2));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[0][mh_][j_]) : "v"(ad), "i"(mh_ * 512));
    };
    auto ldB = [&](unsigned av, int tr_, int m_) __attribute__((always_inline)) {
      const unsigned ad = av + (m_ ? boff1 : boff0);
      asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(B[tr_][m_]) : "v"(ad), "i"(32 * tr_), "i"(32 * tr_ + 64));
    };
    auto pinA = [&](int j_) __attribute__((always_inline)) {
      if (GA) return;   // (compiler-tracked loads)
      asm volatile("" : "+v"(A[0][0][j_]));
      asm volatile("" : "+v"(A[0][1][j_]));
    };
    auto pinB = [&](int m_) __attribute__((always_inline)) {
      asm volatile("" : "+v"(B[0][m_]));
      asm volatile("" : "+v"(B[1][m_]));
    };
    auto pinT = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 2; ++r) { asm volatile("" : "+v"(tdl[c][r])); asm volatile("" : "+v"(tdm[c][r])); asm volatile("" : "+v"(tdr[c][r])); }
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mm = [&](auto p_, auto mh_, auto tr_, auto j_, auto zero_) __attribute__((always_inline)) {
      constexpr int P = decltype(p_)::value, MH = decltype(mh_)::value, TR = decltype(tr_)::value, J = decltype(j_)::value;
      constexpr bool Z = decltype(zero_)::value;
      if (W3_ABLATE(a) & 32) return;
      const w3bf8 av = __builtin_bit_cast(w3bf8, A[GA ? P : 0][MH][J]);
      const w3bf8 bv = __builtin_bit_cast(w3bf8, B[TR][J == 1 ? 1 : 0]);
      if (Z) acc[4 * P + 2 * MH + TR] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, zero16, 0, 0, 0);
      else acc[4 * P + 2 * MH + TR] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[4 * P + 2 * MH + TR], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

    if constexpr (BLK != 3) {
      // One phase: the product A1 x B1 and A3 x B1 (eight MFMAs) with the NEXT phase's input transform between them (its raw
      // rows were read in the previous phase) -- barrier -- the next phase's A1, B1, A3 reads, the product A2 x B2 (four MFMAs)
      // with the raw reads of the transform after next and the DMA issue between them (U three phases ahead into the buffer
      // this phase has finished with, the raw halo two chunks ahead), the next phase's A2, B2 reads.  LDS results return in
      // order: lgkmcnt(4) at the top leaves exactly the A2, B2 reads in flight.
      int ub = 0;   // U buffer of the current phase (s % 3)
      auto phase = [&](int s, auto p_, auto first_, auto next_) __attribute__((always_inline)) {
        constexpr int P = decltype(p_)::value;
        constexpr bool FIRST = decltype(first_)::value, HAS_NEXT = decltype(next_)::value;
        using PP = std::integral_constant<int, P>;
        using PN = std::integral_constant<int, P ^ 1>;
        using Z = std::integral_constant<bool, FIRST>;
        using NZ = std::false_type;
        float* const v_next = smem + (P ? Sh::OFF_V0 : Sh::OFF_V1);
        const int ub1 = ub == 2 ? 0 : ub + 1;
        if (GA) {
          asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");   // (only the B2 reads in flight)
          if (HAS_NEXT) gldA(P ^ 1, s + 1);                       // the set the previous phase has finished with
        } else {
          asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        }
        pinA(0); pinB(0); pinA(2);
        if (HAS_NEXT) pinT();   // (the raw rows of the next phase's transform: read before the A2, B2 reads were issued)
        fence();
        mm(PP{}, I0{}, I0{}, I0{}, Z{});
        if (HAS_NEXT) tf_rows_cols(PN{}, false);
        fence();
        mm(PP{}, I0{}, I1{}, I0{}, Z{});
        if (HAS_NEXT) tf_split(0, v_next);
        fence();
        mm(PP{}, I1{}, I0{}, I0{}, Z{});
        if (HAS_NEXT) tf_split(1, v_next);
        fence();
        mm(PP{}, I1{}, I1{}, I0{}, Z{});
        mm(PP{}, I0{}, I0{}, I2{}, NZ{});
        if (HAS_NEXT) tf_split(2, v_next);
        fence();
        mm(PP{}, I0{}, I1{}, I2{}, NZ{});
        if (HAS_NEXT) tf_split(3, v_next);
        fence();
        mm(PP{}, I1{}, I0{}, I2{}, NZ{});
        mm(PP{}, I1{}, I1{}, I2{}, NZ{});
        if (HAS_NEXT) {
          if (GA) {
            // (the raw halo the next reads take was issued two phases ago, BEFORE the A fragments this phase waited for at its
            // top: loads return in order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          } else {
            if (s + 2 < nph) __builtin_amdgcn_s_waitcnt(W3_WAIT_VM3_LGKM0);   // U(s+1) and the raw halo landed; U(s+2) may still fly
            else __builtin_amdgcn_s_waitcnt(W3_WAIT_VM0_LGKM0);
          }
          __builtin_amdgcn_s_barrier();
          pinA(1); pinB(1);
          const unsigned un = w3_lds_addr(s_ub + ub1 * SUB), vn = w3_lds_addr(v_next);
          if (!GA) { ldA(un, 0, 0); ldA(un, 1, 0); }
          ldB(vn, 0, 0); ldB(vn, 1, 0);
          if (!GA) { ldA(un, 0, 2); ldA(un, 1, 2); }
          fence();
          const int rb2 = ((s + 2) >> 1) & 1;
          mm(PP{}, I0{}, I0{}, I1{}, NZ{});
          if (s + 2 < nph) tf_load_c(PP{}, rb2, 0);
          if (!GA && s + 3 < nph) issue_u_piece(s + 3, ub, 0);
          fence();
          mm(PP{}, I0{}, I1{}, I1{}, NZ{});
          if (s + 2 < nph) tf_load_c(PP{}, rb2, 1);
          if (!GA && s + 3 < nph) issue_u_piece(s + 3, ub, 1);
          fence();
          mm(PP{}, I1{}, I0{}, I1{}, NZ{});
          if (!GA && s + 3 < nph) issue_u_piece(s + 3, ub, 2);
          if (P == 0 && (s >> 1) + 2 < a.nchunks) issue_raw((s >> 1) + 2);
          fence();
          mm(PP{}, I1{}, I1{}, I1{}, NZ{});
          if (!GA) { ldA(un, 0, 1); ldA(un, 1, 1); }
          ldB(vn, 0, 1); ldB(vn, 1, 1);
          fence();
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          pinA(1); pinB(1);
          fence();
          mm(PP{}, I0{}, I0{}, I1{}, NZ{});
          mm(PP{}, I0{}, I1{}, I1{}, NZ{});
          mm(PP{}, I1{}, I0{}, I1{}, NZ{});
          mm(PP{}, I1{}, I1{}, I1{}, NZ{});
        }
        ub = ub1;
      };

      W3_STAMP(0);
#ifdef DVSR_CONV_TRACE
      if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memrealtime();
#endif
      if (GA) {
        gldA(0, 0);
        issue_raw(0);
        issue_raw(1);   // (at least two chunks: conv2d_packed_prepare)
      } else {
        issue_u(0, 0);
        issue_raw(0);
        issue_u(1, 1);
        issue_raw(1);
        issue_u(2, 2);
      }
      __syncthreads();
      W3_STAMP(1);
      tf_load(std::integral_constant<int, 0>{}, 0);
      tf_rows_cols(std::integral_constant<int, 0>{}, true);
      tf_split(0, smem + Sh::OFF_V0); tf_split(1, smem + Sh::OFF_V0); tf_split(2, smem + Sh::OFF_V0); tf_split(3, smem + Sh::OFF_V0);
      __syncthreads();
      {
        const unsigned u0 = w3_lds_addr(s_ub), v0 = w3_lds_addr(smem + Sh::OFF_V0);
        if (!GA) { ldA(u0, 0, 0); ldA(u0, 1, 0); }
        ldB(v0, 0, 0); ldB(v0, 1, 0);
        if (!GA) { ldA(u0, 0, 2); ldA(u0, 1, 2); }
        tf_load(std::integral_constant<int, 1>{}, 0);
        if (!GA) { ldA(u0, 0, 1); ldA(u0, 1, 1); }
        ldB(v0, 0, 1); ldB(v0, 1, 1);
      }
      W3_STAMP(2);
      using P0 = std::integral_constant<int, 0>;
      using P1 = std::integral_constant<int, 1>;
      using T = std::true_type;
      using F = std::false_type;
      phase(0, P0{}, T{}, T{});
      phase(1, P1{}, T{}, T{});
      W3_STAMP(3);
      for (int k = 1; k + 1 < a.nchunks; ++k) {
        phase(2 * k, P0{}, F{}, T{});
        phase(2 * k + 1, P1{}, F{}, T{});
        if (k < 30) W3_STAMP(3 + k);
      }
      phase(nph - 2, P0{}, F{}, T{});
      phase(nph - 1, P1{}, F{}, F{});
    } else {
      // ---- BLK == 3: ONE barrier per chunk.  With U out of the LDS there is room for the V images of two whole chunks
      // (4 x 24 KB), so phase s = 2 k + P reads V[k & 1][P] and transforms V[(k + 1) & 1][P] -- the SAME rows of the next
      // chunk -- and only the phase P = 1 carries the barrier that hands the next chunk's images over (its raw rows are read
      // one phase ahead, as before).  Everything else as BLK == 2.
      static_assert(BLK != 3 || GA, "");
      auto v_img = [&](int set, int par) { return smem + (set ? Sh::OFF_U + par * SUB : (par ? Sh::OFF_V1 : Sh::OFF_V0)); };
      auto phase = [&](int s, auto p_, auto first_, auto next_, auto tf_) __attribute__((always_inline)) {
        constexpr int P = decltype(p_)::value;
        constexpr bool FIRST = decltype(first_)::value, HAS_NEXT = decltype(next_)::value;
        // (debug-build ablations, results wrong: 2 no input transform, 4 no global loads (U fragments, raw halo), 8 no V operand
        // reads, 16 no barrier, 32 no MFMAs; W3_ABLATE is the constant 0 in the product build)
        const bool TF = decltype(tf_)::value && !(W3_ABLATE(a) & 2);
        using PP = std::integral_constant<int, P>;
        using PN = std::integral_constant<int, P ^ 1>;
        using Z = std::integral_constant<bool, FIRST>;
        using NZ = std::false_type;
        const int kc = s >> 1;
        float* const v_wr = v_img((kc + 1) & 1, P);                      // V(s + 2)
        const float* const v_nx = P ? v_img((kc + 1) & 1, 0) : v_img(kc & 1, 1);   // V(s + 1)
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");   // (only the B2 reads in flight)
        if (HAS_NEXT && !(W3_ABLATE(a) & 4)) gldA(P ^ 1, s + 1);   // the set the previous phase has finished with
        // the raw halo two chunks ahead, into the buffer whose rows were last read before the previous chunk's barrier: two
        // phases to land (the input of the 720p layers comes from HBM, not from the last-level cache)
        if (P == 0 && kc + 2 < a.nchunks && !(W3_ABLATE(a) & 4)) issue_raw(kc + 2);
        pinB(0);
        if (TF) pinT();
        fence();
        mm(PP{}, I0{}, I0{}, I0{}, Z{});
        if (TF) tf_rows_cols(PP{}, false);
        fence();
        mm(PP{}, I0{}, I1{}, I0{}, Z{});
        if (TF) tf_split(0, v_wr);
        fence();
        mm(PP{}, I1{}, I0{}, I0{}, Z{});
        if (TF) tf_split(1, v_wr);
        fence();
        mm(PP{}, I1{}, I1{}, I0{}, Z{});
        mm(PP{}, I0{}, I0{}, I2{}, NZ{});
        if (TF) tf_split(2, v_wr);
        fence();
        mm(PP{}, I0{}, I1{}, I2{}, NZ{});
        if (TF) tf_split(3, v_wr);
        fence();
        mm(PP{}, I1{}, I0{}, I2{}, NZ{});
        mm(PP{}, I1{}, I1{}, I2{}, NZ{});
        if (P == 1 && HAS_NEXT) {
          // the raw halo issued in the phase before this one has landed (only the six A loads of this phase's top may still
          // fly), this wave's V words are written
          __builtin_amdgcn_s_waitcnt(0x0076);   // vmcnt(6) lgkmcnt(0)
          if (!(W3_ABLATE(a) & 16)) __builtin_amdgcn_s_barrier();
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        pinB(1);
        if (HAS_NEXT) {
          const unsigned vn = w3_lds_addr(v_nx);
          if (!(W3_ABLATE(a) & 8)) { ldB(vn, 0, 0); ldB(vn, 1, 0); }
          fence();
          const int rb3 = ((s + 3) >> 1) & 1;
          mm(PP{}, I0{}, I0{}, I1{}, NZ{});
          if (s + 3 < nph && !(W3_ABLATE(a) & 2)) tf_load_c(PN{}, rb3, 0);
          fence();
          mm(PP{}, I0{}, I1{}, I1{}, NZ{});
          if (s + 3 < nph && !(W3_ABLATE(a) & 2)) tf_load_c(PN{}, rb3, 1);
          fence();
          mm(PP{}, I1{}, I0{}, I1{}, NZ{});
          mm(PP{}, I1{}, I1{}, I1{}, NZ{});
          if (!(W3_ABLATE(a) & 8)) { ldB(vn, 0, 1); ldB(vn, 1, 1); }
          fence();
        } else {
          fence();
          mm(PP{}, I0{}, I0{}, I1{}, NZ{});
          mm(PP{}, I0{}, I1{}, I1{}, NZ{});
          mm(PP{}, I1{}, I0{}, I1{}, NZ{});
          mm(PP{}, I1{}, I1{}, I1{}, NZ{});
        }
      };

      W3_STAMP(0);
#ifdef DVSR_CONV_TRACE
      if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memrealtime();
#endif
      gldA(0, 0);
      issue_raw(0);
      issue_raw(1);   // (at least two chunks: conv2d_packed_prepare)
      __syncthreads();
      W3_STAMP(1);
      tf_load(std::integral_constant<int, 0>{}, 0);
      tf_rows_cols(std::integral_constant<int, 0>{}, true);
      tf_load(std::integral_constant<int, 1>{}, 0);   // (under the splits of the first rows)
      tf_split(0, v_img(0, 0)); tf_split(1, v_img(0, 0)); tf_split(2, v_img(0, 0)); tf_split(3, v_img(0, 0));
      tf_rows_cols(std::integral_constant<int, 1>{}, true);
      tf_split(0, v_img(0, 1)); tf_split(1, v_img(0, 1)); tf_split(2, v_img(0, 1)); tf_split(3, v_img(0, 1));
      __syncthreads();
      {
        const unsigned v0 = w3_lds_addr(v_img(0, 0));
        ldB(v0, 0, 0); ldB(v0, 1, 0);
        tf_load(std::integral_constant<int, 0>{}, 1);   // rows of phase 2 (chunk 1)
        ldB(v0, 0, 1); ldB(v0, 1, 1);
      }
      W3_STAMP(2);
      using P0 = std::integral_constant<int, 0>;
      using P1 = std::integral_constant<int, 1>;
      using T = std::true_type;
      using F = std::false_type;
      phase(0, P0{}, T{}, T{}, T{});
      phase(1, P1{}, T{}, T{}, T{});
      W3_STAMP(3);
      for (int k = 1; k + 1 < a.nchunks; ++k) {
        phase(2 * k, P0{}, F{}, T{}, T{});
        phase(2 * k + 1, P1{}, F{}, T{}, T{});
        if (k < 30) W3_STAMP(3 + k);
      }
      phase(nph - 2, P0{}, F{}, T{}, F{});
      phase(nph - 1, P1{}, F{}, F{}, F{});
    }
    W3_STAMP(40);

    // ---- epilogue.  Y = A^T M A, A^T = [[1, 1, 1, 0], [0, 1, -1, -1]].  This wave holds M[xq][nu] and M[xq + 2][nu] of ALL
    // 64 couts x 64 tiles: it forms C_i = sum_xi A^T[i][xi] M[xi][nu] of its two rows in place (xq = 0: C0 = M0 + M2,
    // C1 = -M2; xq = 1: C0 = M1, C1 = M1 - M3), then per cout half the eight waves lay their C_i out in the (now idle) LDS
    // and wave (tr_o, rq_o) finishes Y[i][j] = sum_waves A^T[j][nu] C_i of tile half tr_o and accumulator registers
    // 4 rq_o .. 4 rq_o + 3 (four consecutive couts): bias, activation, residual / accumulate / gradient mask, store.
    // (the lane index passes through an opaque asm: nothing of the epilogue's per-lane addressing can be hoisted above the
    // K loop, where every register is spoken for)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lo_e = lane_e & 31, hi_e = lane_e >> 5;
    // (xq = 0 keeps +M2 in place of C1 = -M2: the readers subtract it)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (xq == 0) acc[b] += acc[4 + b];
      else acc[4 + b] = acc[b] - acc[4 + b];
    }
    const int tr_o = wave & 1, rq_o = wave >> 1;
    const int ttw = tr_o * 32 + lo_e;                      // this lane's tile
    const int orow = oy0 + 2 * (ttw / TC), ocol = ox0 + 2 * (ttw % TC);
    const size_t HWo = (size_t)a.Ho * a.Wo;
    const float slope = a.act == ACT_LRELU ? 0.1f : (a.act == ACT_RELU ? 0.f : 1.f);
    const float neg = a.gmask_act == ACT_LRELU ? 0.1f : (a.gmask_act == ACT_RELU ? 0.f : 1.f);
    const bool full = oy0 + Sh::OH <= a.Ho && ox0 + Sh::OW <= a.Wo && cbi * 64 + 64 <= a.Cout;
    // exchange image: [source wave 8][i 2][tr 2][rq 4][lane 64] x 16 B = 128 KB
    float* const xw = smem + wave * 4096 + lane_e * 4;
    const float* const xr = smem + (tr_o * 4 + rq_o) * 256 + lane_e * 4;
    const bool plain = !a.res && !a.accum && !a.gmask;
    const unsigned lane_off = (unsigned)(((size_t)(4 * hi_e) * HWo + (size_t)orow * a.Wo + ocol) * 4);
    // (raw barriers: __syncthreads() would also wait for the global stores of the first round to be acknowledged)
    auto lds_barrier = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    W3_STAMP(50);
    lds_barrier();   // every wave is past its last operand read
    W3_STAMP(51);
    // one round = one cout half R: write (all C_i of the half), barrier, read + sum (this wave's slice); the stores of the
    // first round are issued behind the LDS writes of the second, under the other waves' writes
    auto write_round = [&](auto r_, w3f2 (&ex)[4][2]) __attribute__((always_inline)) {
      constexpr int R = decltype(r_)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const f32x16& m = acc[4 * i + 2 * R + t];
            *reinterpret_cast<f32x4*>(xw + ((i * 2 + t) * 4 + rq) * 256) = f32x4{m[4 * rq], m[4 * rq + 1], m[4 * rq + 2], m[4 * rq + 3]};
          }
      // residual / accumulate operands of the round, loaded under the exchange
      if (full && !plain && a.ps == 0) {
        const int cob = cbi * 64 + R * 32 + 8 * rq_o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const size_t sb = (((size_t)n * a.Cout + cob + k) * HWo + (size_t)i * a.Wo) * 4;   // scalar
            w3f2 e = {0.f, 0.f};
            if (a.res) e = *reinterpret_cast<const w3f2*>(reinterpret_cast<const char*>(a.res) + sb + lane_off);
            if (a.accum) e += *reinterpret_cast<const w3f2*>(reinterpret_cast<const char*>(a.y) + sb + lane_off);
            ex[k][i] = e;
          }
      }
    };
    auto read_round = [&](f32x4 (&y)[2][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v[8];
#pragma unroll
        for (int src = 0; src < 8; ++src) v[src] = *reinterpret_cast<const f32x4*>(xr + (src * 16 + i * 8) * 256);
        // sources src (xq = 0) and src + 4 (xq = 1) hold the same nu; i = 1: the xq = 0 waves left +M2 where C1 = -M2
        f32x4 nn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) nn[q] = i == 0 ? v[q + 4] + v[q] : v[q + 4] - v[q];
        y[i][0] = nn[0] + nn[1] + nn[2];
        y[i][1] = nn[1] - nn[2] - nn[3];
      }
    };
    auto finish_round = [&](auto r_, f32x4 (&y)[2][2], w3f2 (&ex)[4][2]) __attribute__((always_inline)) {
      constexpr int R = decltype(r_)::value;
      const int cob = cbi * 64 + R * 32 + 8 * rq_o;   // scalar; the lane's couts are cob + 4 hi + k
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float v = y[i][j][k] + bk[R][k];
            y[i][j][k] = fmaxf(v, v * slope);
          }
      if (a.ps == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            w3f2 v = {y[i][0][k], y[i][1][k]};
            if (full) {
              const size_t sb = (((size_t)n * a.Cout + cob + k) * HWo + (size_t)i * a.Wo) * 4;   // scalar
              if (!plain) {
                v += ex[k][i];
                if (a.gmask) {   // (data-gradient launches: the activation mask of the producer)
                  const w3f2 m = *reinterpret_cast<const w3f2*>(reinterpret_cast<const char*>(a.gmask) + sb + lane_off);
                  v = w3f2{v[0] * (m[0] > 0.f ? 1.f : neg), v[1] * (m[1] > 0.f ? 1.f : neg)};
                }
              }
              *reinterpret_cast<w3f2*>(reinterpret_cast<char*>(a.y) + sb + lane_off) = v;
              continue;
            }
            const int co = cob + 4 * hi_e + k;
            const int oy = orow + i;
            const size_t idx = ((size_t)n * a.Cout + co) * HWo + (size_t)oy * a.Wo + ocol;
            const bool ok0 = co < a.Cout && oy < a.Ho && ocol < a.Wo;
            const bool ok1 = ok0 && ocol + 1 < a.Wo;
            if (!ok0) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (j == 1 && !ok1) continue;
              float w = v[j];
              if (a.res) w += a.res[idx + j];
              if (a.accum) w += a.y[idx + j];
              if (a.gmask) w *= a.gmask[idx + j] > 0.f ? 1.f : neg;
              a.y[idx + j] = w;
            }
          }
        }
      } else {
        // PixelShuffle(2): channels co0 .. co0 + 3 are the 2x2 sub-pixels (dy, dx) of channel co0 / 4
        const int co0 = cob + 4 * hi_e;
        const int cq = co0 >> 2;
        if (co0 < a.Cout) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int oy = orow + i;
            if (!full && (oy >= a.Ho || ocol >= a.Wo)) continue;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
              const f32x4 v = f32x4{y[i][0][2 * dy], y[i][0][2 * dy + 1], y[i][1][2 * dy], y[i][1][2 * dy + 1]};
              float* dst = a.y + (((size_t)n * (a.Cout >> 2) + cq) * (2 * a.Ho) + (2 * oy + dy)) * (size_t)(2 * a.Wo) + 2 * ocol;
              if (full || ocol + 1 < a.Wo) *reinterpret_cast<f32x4*>(dst) = v;
              else *reinterpret_cast<w3f2*>(dst) = w3f2{v[0], v[1]};
            }
          }
        }
      }
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    f32x4 y0[2][2], y1[2][2];
    w3f2 ex0[4][2], ex1[4][2];
    write_round(R0{}, ex0);
    W3_STAMP(52);
    lds_barrier();
    W3_STAMP(53);
    read_round(y0);
    W3_STAMP(54);
    lds_barrier();   // the reads of the first round are done
    W3_STAMP(55);
    write_round(R1{}, ex1);
    W3_STAMP(56);
    finish_round(R0{}, y0, ex0);
    W3_STAMP(57);
    lds_barrier();
    read_round(y1);
    W3_STAMP(58);
    finish_round(R1{}, y1, ex1);
  }
#ifdef DVSR_CONV_TRACE
  W3_STAMP(41);
  __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
  W3_STAMP(42);
  if (a.trace && threadIdx.x == 0) {
    a.trace[(size_t)blockIdx.x * 64 + 61] = __builtin_amdgcn_s_memrealtime();
    a.trace[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_ID
  }
#endif
}

template <int TC, int BLK>
static int launch_wino3(ConvK2 k, hipStream_t st) {
  using Sh = Wino3Shape<TC>;
  auto kern = conv2d_wino3_kernel<TC, BLK>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, Sh::LDS_BYTES);k.tiles_x = ceil_div(k.Wo, Sh::OW); k.tiles_y = ceil_div(k.Ho, Sh::OH); k.ntiles = k.tiles_x * k.tiles_y * k.N;
  k.ncb = ceil_div(k.Cout, 64);
  k.tiles_per_xcd = ceil_div(k.ntiles, 8);
  k.nitems = k.tiles_per_xcd * 8 * k.ncb;
  hipLaunchKernelGGL(kern, dim3(k.nitems), dim3(512), Sh::LDS_BYTES, st, k);
  return check_launch("conv2d_wino3_kernel");
}

// th = 4: 4 x 64-pixel workgroup tiles (TC = 32), th = 8: 8 x 32 (TC = 16), th = 16: 16 x 16 (TC = 8)
// DVSR_CONV_WINO3_BLK: 4 (default, round 5) the B operand built in registers (conv2d_wino4.hip); 0-3 the forms of this file
// (A/B aids: 0 four xn per wave, 1 one xn per wave, 2 + U from global, 3 + one barrier per chunk = the round-4 default)
int conv2d_wino3_launch(const ConvK2& k, int th, hipStream_t st) {
  static const int blk = [] { const char* v = getenv("DVSR_CONV_WINO3_BLK"); return v ? atoi(v) : 4; }();
  if (blk == 4) return conv2d_wino4_launch(k, th, st);   // conv2d_wino4.hip: the B operand built in registers
  if (th == 16) {   // 16 x 16-pixel tiles (TC = 8)
    if (blk == 0) return launch_wino3<8, 0>(k, st);
    if (blk == 1) return launch_wino3<8, 1>(k, st);
    if (blk == 2) return launch_wino3<8, 2>(k, st);
    return launch_wino3<8, 3>(k, st);
  }
  if (blk == 0) return th == 8 ? launch_wino3<16, 0>(k, st) : launch_wino3<32, 0>(k, st);
  if (blk == 2) return th == 8 ? launch_wino3<16, 2>(k, st) : launch_wino3<32, 2>(k, st);
  if (blk == 3) return th == 8 ? launch_wino3<16, 3>(k, st) : launch_wino3<32, 3>(k, st);
  return th == 8 ? launch_wino3<16, 1>(k, st) : launch_wino3<32, 1>(k, st);
}

}  // namespace dvsr
