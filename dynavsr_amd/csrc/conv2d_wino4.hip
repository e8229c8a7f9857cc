// 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) on the bf16 matrix pipe at fp32 accuracy (the exact 3-way
// operand split of conv2d_wino3.hip; EDVR_arch.py:254-313 is what is being computed) -- round 5, "form 4": the transformed
// input V NEVER TOUCHES THE LDS.
//
// Why.  In forms 0-3 (conv2d_wino3.hip) every wave transforms a slice of the chunk, writes its V words to the LDS
// (192 ds_write_b32 wave-instructions per chunk and CU), a barrier hands them over and every wave reads its B fragments
// back (64 ds_read2_b64).  tools/mfma_overlap.hip measures what that costs on MI355X: a wave issues one ds_write_b32 per
// 32 cycles, four waves together get 32 B/clk -- the V writes of a chunk alone keep the LDS store path busy for as long
// as the chunk's 192 MFMAs keep the matrix pipes (1536 cycles), and PMC showed 29 % of all LDS-array cycles as bank
// conflicts (the 2-way V writes and the stride-2 ds_read_b32 of the patch columns).  The chunk loop ran at 3100-3370
// cycles.
//
// Here the wave that multiplies with a transformed-domain point (xi, nu) computes that point's V values ITSELF, for all
// 64 tiles x 8 channels, straight into the MFMA's B-fragment registers:
//   * wave (r = wave & 3, np = wave >> 2) owns the two points (xi = r, nu in the pair np): slot 0 = nu 0 (np 0) / nu 3
//     (np 1) -- patch columns (c0, c2) / (c1, c3), i.e. "X - Y" with Y two floats right of X --, slot 1 = nu 1 / nu 2 --
//     the aligned column pair (c1, c2), "Q + s P";
//   * lane = tile (lanes 0-31: tile half tr = 0, lanes 32-63: tr = 1).  Per slot the lane reads the two raw rows xi
//     combines, of all 8 channels of the chunk (slot 0: 32 ds_read_b32, slot 1: 16 ds_read_b64), forms the 8 V values
//     (3 VALU each), splits channel pairs into three bf16 pieces (v_cvt_pk_bf16_f32, 11 VALU per pair) and
//   * v_permlane32_swap builds the fragments: with H = the hi pieces of (lower lanes: tr 0 | upper lanes: tr 1) and M the
//     mid pieces, ONE swap per register turns (H, M) into B1[tr 0] = (hi | mid of tr 0) and B1[tr 1] = (hi | mid of tr 1)
//     -- the K = 16 = 8 channels x 2 pieces layout of v_mfma_f32_32x32x16_bf16 (lanes 32-63 carry K 8..15) --, a copy of H
//     and L give B2 = (hi | lo) the same way: 12 VALU per slot instead of 12 LDS writes + 4 LDS reads + a barrier.
// A fragments (the packed transformed weights, conv2d_wino3.hip's image) come straight from global memory as before, each
// reloaded with the next chunk's right behind its last MFMA.  Per chunk a wave issues 24 MFMAs (slot 0: A1 B1, A3 B1,
// A2 B2 on the four 32 x 32 blocks; then slot 1), ~165 VALU, 48 LDS reads, 12 global loads and its share of the raw halo
// DMA: no LDS write, no V image, ONE loose barrier per chunk (it only hands the raw halo over).  LDS: three raw chunks
// (3 x 16 KB) during the loop; the epilogue's exchange (128 KB) reuses it.
//
// The halo DMA is UNCONDITIONAL: lanes whose 16-byte group lies outside the image carry a byte offset beyond the buffer
// resource's num_records (the hardware drops the access; the group was zeroed once in the prologue), so every wave issues
// exactly NI LDS-DMA instructions per chunk and every s_waitcnt vmcnt below is an exact count (with the predicated DMA of
// forms 0-3 the compiler had to assume none was issued, which forced the halo to land within half a chunk).
//
// Epilogue: as form 3 with the roles of rows and columns exchanged (a wave holds two nu of one xi): in place
// slot0 += slot1 (np 0: D_0 = M0 + M1, D_1 = M1; np 1: slot0 = M3 + M2 = -D_1, slot1 = M2 = D_0), all-to-all through the
// LDS per cout half, reader sums S_j over np and Y[0][j] = S_j[0] + S_j[1] + S_j[2], Y[1][j] = S_j[1] - S_j[2] - S_j[3].
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

#ifdef DVSR_CONV_TRACE
// cycle stamps of the debug build (tools/wino_trace.py): thread 0 of every workgroup, slots as in conv2d_wino3.hip; W4_FINE:
// lane 0 of waves 0 and 4 (the two waves of one SIMD) inside chunk 3, slots 20 + i / 30 + i
#define W4_STAMP(i)                                                                                       \
  do {                                                                                                    \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#define W4_FINE(i)                                                                                                      \
  do {                                                                                                                  \
    if (a.trace && k == 3 && lane == 0 && (wave & 3) == 0)                                                              \
      a.trace[(size_t)blockIdx.x * 64 + (wave ? 30 : 20) + (i)] = __builtin_readcyclecounter();                          \
  } while (0)
#else
#define W4_STAMP(i) \
  do {              \
  } while (0)
#define W4_FINE(i) \
  do {             \
  } while (0)
#endif

typedef float w4f2 __attribute__((ext_vector_type(2)));
typedef __bf16 w4bf8 __attribute__((ext_vector_type(8)));
typedef unsigned w4u4 __attribute__((ext_vector_type(4)));

template <int TC>
struct Wino4Shape {
  static constexpr int CC = 8, NTILE = 64, TRW = NTILE / TC;
  static constexpr int OH = 2 * TRW, OW = 2 * TC;      // output pixels of the workgroup tile
  static constexpr int IH = OH + 2, RP = OW + 8, GR = RP / 4;
  static constexpr int NG = CC * IH * GR;              // 16-byte groups of one chunk's raw halo image
  static constexpr int NI = (NG + 511) / 512;
  static constexpr int RAWPAD = NI * 512 * 4;          // floats of one raw buffer (every lane of every DMA has a slot)
  static constexpr int XCH = 32768;                    // floats of the epilogue's exchange image (128 KB)
  static constexpr int NBUF = 3;                       // raw chunks in the LDS: being read, landed for the next chunk, in flight
  static constexpr size_t LDS_BYTES = (size_t)(NBUF * RAWPAD > XCH ? NBUF * RAWPAD : XCH) * sizeof(float);
};

// s_waitcnt immediate (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14]); lgkm = 15: no wait
constexpr int w4_waitcnt(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

__device__ __forceinline__ unsigned w4_lds_addr(const float* p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const float*)p);
}
// v_cvt_pk_bf16_f32: {bf16(a) (round to nearest even) in bits 15:0, bf16(b) in bits 31:16}.  (As a vector conversion, not
// inline asm: behind an asm the compiler pads every dependent use with an s_nop -- 24 per chunk in conv2d_wino3.hip -- because it
// cannot see which hazards the instruction has; this form it schedules itself.)
typedef __bf16 w4bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned w4_cvt_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(w4f2{a, b}, w4bf2));
}

// one LDS-DMA instruction: 64 lanes x 16 bytes at lds + 16 lane (in a function of its own: used directly inside the kernel
// template the builtin keeps the HOST pass from instantiating the kernel's stub)
__device__ __forceinline__ void w4_dma16(__amdgpu_buffer_rsrc_t rs, float* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_rsrc(const float* base, int num_records) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, num_records, 0x00020000);
}

template <int TC>
__global__ __launch_bounds__(512, 2) void conv2d_wino4_kernel(ConvK2 a) {
  using Sh = Wino4Shape<TC>;
  constexpr int IH = Sh::IH, RP = Sh::RP, GR = Sh::GR, NI = Sh::NI;
  constexpr int SUB = 6144;                 // floats (24 KB) of one phase image of the packed weights
  constexpr int CHB = IH * RP * 4;          // bytes between two channels of the raw image
  extern __shared__ __attribute__((aligned(16))) float smem[];

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
  const int r = wave & 3, np = wave >> 2;   // xi row, nu pair
  const size_t HW = (size_t)a.H * a.W;
  const float* x0n = a.x0 + (size_t)n * a.x0_bs;
  const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : x0n;

  // raw halo groups this lane moves: group L = 64 * (wave + 8 jj) + lane = (channel, row, column group); a group outside the
  // image (or past the image: L >= NG) carries an offset the buffer resource rejects
  unsigned hoff[NI];
#pragma unroll
  for (int jj = 0; jj < NI; ++jj) {
    const int L = 64 * (wave + 8 * jj) + lane;
    const int c = L / (IH * GR), rr = L - c * (IH * GR);
    const int iy = rr / GR, g = rr - iy * GR;
    const int gy = oy0 - 1 + iy, gx = ox0 - 4 + 4 * g;
    const bool ok = L < Sh::NG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    hoff[jj] = ok ? (unsigned)(((size_t)c * HW + (size_t)gy * a.W + gx) * 4) : 0x80000000u;
#ifndef W4_NOZERO
    if (L < Sh::NG && !ok)
#else
    if (false)   // (probe build: does the LDS-DMA write zeros for lanes the buffer resource rejects?)
#endif
    {
#pragma unroll
      for (int bb = 0; bb < Sh::NBUF; ++bb) *reinterpret_cast<f32x4*>(smem + bb * Sh::RAWPAD + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const unsigned chunk_bytes = (unsigned)(Sh::CC * HW * 4);
  // (the resource is SELECTED, not branched on -- second input; past the last chunk an empty one: nothing is read -- so that
  // every path issues the same NI instructions and the compiler's own vmcnt bookkeeping for the A fragments stays exact)
  int dbuf = 0;   // buffer of the next raw chunk to be fetched (chunk k lives in buffer k % NBUF)
  auto issue_raw = [&](int k) __attribute__((always_inline)) {
    const bool live = k < a.nchunks;
    const bool second = k * Sh::CC >= a.c0;  // only possible when c1 > 0; a chunk never straddles the two inputs
    const unsigned soff = live ? (unsigned)(second ? k - a.c0 / Sh::CC : k) * chunk_bytes : 0u;
    const __amdgpu_buffer_rsrc_t rs = w4_rsrc(second ? x1n : x0n, live ? 0x7fffffff : 0);
    float* dst = smem + dbuf * Sh::RAWPAD;
#pragma unroll
    for (int jj = 0; jj < NI; ++jj)
      w4_dma16(rs, dst + 256 * (wave + 8 * jj), hoff[jj], soff);
    dbuf = dbuf == Sh::NBUF - 1 ? 0 : dbuf + 1;
  };

  // ---- A fragments: packed image P16[cb][k][p][piece][xl][cout 64][8 ch] bf16 (pack_weights_wino3_kernel).  Lane halves read
  // the pieces (A1: hi|hi, A2: mid|hi, A3: lo|mid); mh = 1 at + 512 bytes.
  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + (size_t)cbi * a.nchunks * (2 * SUB);
  const __amdgpu_buffer_rsrc_t wrsrc = w4_rsrc(wp_cb, -1);
  const unsigned av0 = (unsigned)(lo * 16);
  const unsigned av1 = av0 + (hi ? 0u : 8192u), av2 = av0 + (hi ? 8192u : 16384u);
  // slot 0: nu = 0 (np 0) / 3 (np 1); slot 1: nu = 1 / 2
  const int sb0 = (r >> 1) * (SUB * 4) + ((r & 1) * 4 + (np ? 3 : 0)) * 1024;
  const int sb1 = (r >> 1) * (SUB * 4) + ((r & 1) * 4 + (np ? 2 : 1)) * 1024;
  f32x4 A[2][2][3];   // [slot][mh][A1 / A2 / A3]
  auto gldA = [&](auto e_, auto j_, int k) __attribute__((always_inline)) {   // fragments A_j of slot e, chunk k (both couts halves)
    constexpr int E = decltype(e_)::value, J = decltype(j_)::value;
#ifdef W4_NOA   // (probe build, results wrong: only the first chunk's fragments are loaded -- what do the A loads cost?)
    if (k > 0) return;
#endif
    const int soff = k * (2 * SUB * 4) + (E ? sb1 : sb0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
      A[E][m][J] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)((J == 0 ? av0 : (J == 1 ? av1 : av2)) + m * 512), soff, 0));
  };

  // ---- the wave's rows and columns of the 4x4 patch.  Rows of B^T d: xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3;
  // the patch of tile (trow, tcol) starts at image float (2 trow) RP + 2 tcol + 3
  const int trow_t = lane / TC, tcol_t = lane - trow_t * TC;
  const int ra = r == 0 ? 0 : (r == 2 ? 2 : 1), rb = r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
  const float sg = r == 1 ? 1.f : -1.f;      // row xi = d[ra] + sg d[rb]
  const float s1 = np ? -1.f : 1.f;          // slot 1: nu 1 = c1 + c2, nu 2 = c2 - c1 = Q + s1 P
  const int pb = 2 * trow_t * RP + 2 * tcol_t + 3;
  const unsigned lbase = w4_lds_addr(smem);
  // slot 0 reads X at +0 and Y at +8 bytes of these (np 0: c0, c2; np 1: c1, c3); slot 1 reads the aligned pair (c1, c2)
  unsigned aA0 = lbase + (unsigned)((pb + ra * RP + (np ? 1 : 0)) * 4), aB0 = lbase + (unsigned)((pb + rb * RP + (np ? 1 : 0)) * 4);
  unsigned aAp = lbase + (unsigned)((pb + ra * RP + 1) * 4), aBp = lbase + (unsigned)((pb + rb * RP + 1) * 4);
  int hbuf = 0, lbuf = 0;   // the raw buffer the slot-0 / slot-1 reads point into
  auto rotate = [&](unsigned& x0, unsigned& x1, int& buf) __attribute__((always_inline)) {   // on to the next chunk's buffer
    const int d = buf == Sh::NBUF - 1 ? -(Sh::NBUF - 1) * Sh::RAWPAD * 4 : Sh::RAWPAD * 4;
    x0 += d; x1 += d;
    buf = buf == Sh::NBUF - 1 ? 0 : buf + 1;
  };

  // B fragments of the two slots: after `finalize`, Bh = B1[tr 0], Bm = B1[tr 1], Bc = B2[tr 0], Bl = B2[tr 1]
  unsigned Bh[2][4], Bm[2][4], Bl[2][4], Bc[2][4];
  float t[8];        // raw values of one channel pair, slot 0: [c][XA, YA, XB, YB]
  w4f2 tp[4];        // slot 1: [c][(P, Q) of row a, (P, Q) of row b]
  // (inline asm: single ds_read_b32 / ds_read_b64 with immediate channel offsets into ONE recycled register set.  Their
  // results are waited for by `wait_lds` -- an lgkmcnt(0) CLOSED BY A SCHEDULING FENCE, so that no consumer can move above it;
  // no "+v" pins on the registers: behind an asm that defines a register the compiler pads the first use with an s_nop.
  // These are the only LDS operations of the chunk loop.)
  auto wait_lds = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // (non-generic lambdas: clang rejects asm operands that name captured arrays inside a generic lambda)
  auto load0 = [&](int P) __attribute__((always_inline)) {   // slot 0, channel pair P
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t[4 * c + 0]) : "v"(aA0), "i"((2 * P + c) * CHB));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t[4 * c + 1]) : "v"(aA0), "i"((2 * P + c) * CHB + 8));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t[4 * c + 2]) : "v"(aB0), "i"((2 * P + c) * CHB));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t[4 * c + 3]) : "v"(aB0), "i"((2 * P + c) * CHB + 8));
    }
  };
  auto load1 = [&](int P) __attribute__((always_inline)) {   // slot 1, channel pair P
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tp[2 * c + 0]) : "v"(aAp), "i"((2 * P + c) * CHB));
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tp[2 * c + 1]) : "v"(aBp), "i"((2 * P + c) * CHB));
    }
  };
  // the pair's two V values -> three exact bf16 pieces each, packed per piece (channel 2 P in the low half)
  auto comb_split = [&](auto e_, auto p_) __attribute__((always_inline)) {
    constexpr int E = decltype(e_)::value, P = decltype(p_)::value;
    float v[2];
    if constexpr (E == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float fx = __builtin_fmaf(sg, t[4 * c + 2], t[4 * c + 0]);   // column X of row xi
        const float fy = __builtin_fmaf(sg, t[4 * c + 3], t[4 * c + 1]);   // column Y
        v[c] = fx - fy;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float fp = __builtin_fmaf(sg, tp[2 * c + 1][0], tp[2 * c + 0][0]);   // column c1 of row xi
        const float fq = __builtin_fmaf(sg, tp[2 * c + 1][1], tp[2 * c + 0][1]);   // column c2
        v[c] = __builtin_fmaf(s1, fp, fq);
      }
    }
    const unsigned h = w4_cvt_pk(v[0], v[1]);
    const float r0 = v[0] - __builtin_bit_cast(float, h << 16), r1 = v[1] - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = w4_cvt_pk(r0, r1);
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    Bh[E][P] = h;
    Bm[E][P] = m;
    Bl[E][P] = w4_cvt_pk(q0, q1);
  };
  // (two steps, each over the four registers: a swap must not read a register written by the instruction in front of it)
  auto finalize_a = [&](auto e_) __attribute__((always_inline)) {
    constexpr int E = decltype(e_)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[E][i] = Bh[E][i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const auto s0 = __builtin_amdgcn_permlane32_swap(Bh[E][i], Bm[E][i], false, false);
      Bh[E][i] = s0[0]; Bm[E][i] = s0[1];
    }
  };
  auto finalize_b = [&](auto e_) __attribute__((always_inline)) {
    constexpr int E = decltype(e_)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const auto s0 = __builtin_amdgcn_permlane32_swap(Bc[E][i], Bl[E][i], false, false);
      Bc[E][i] = s0[0]; Bl[E][i] = s0[1];
    }
  };

  f32x16 acc[8];   // acc[4 slot + 2 mh + tr]
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  auto mm = [&](auto e_, auto mh_, auto tr_, auto j_, auto zero_) __attribute__((always_inline)) {
    constexpr int E = decltype(e_)::value, MH = decltype(mh_)::value, TR = decltype(tr_)::value, J = decltype(j_)::value;
    constexpr bool Z = decltype(zero_)::value;
    const w4bf8 av = __builtin_bit_cast(w4bf8, A[E][MH][J]);
    const w4u4 bu = J == 1 ? (TR ? w4u4{Bl[E][0], Bl[E][1], Bl[E][2], Bl[E][3]} : w4u4{Bc[E][0], Bc[E][1], Bc[E][2], Bc[E][3]})
                           : (TR ? w4u4{Bm[E][0], Bm[E][1], Bm[E][2], Bm[E][3]} : w4u4{Bh[E][0], Bh[E][1], Bh[E][2], Bh[E][3]});
    const w4bf8 bv = __builtin_bit_cast(w4bf8, bu);
    if (Z) acc[4 * E + 2 * MH + TR] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, zero16, 0, 0, 0);
    else acc[4 * E + 2 * MH + TR] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[4 * E + 2 * MH + TR], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using T = std::true_type;
  using F = std::false_type;

  // One half of a chunk: the twelve MFMAs of slot E (A1 B1 and A3 B1 on the four blocks, then A2 B2) with, in the gaps, the
  // WHOLE next fragment set of the other slot -- slot 1 of this chunk under slot 0, slot 0 of the next chunk under slot 1 --
  // (per channel pair: wait + combine + split, the next pair's reads issued right behind the combine) and the reloads of
  // this slot's A fragments with the next chunk's, each right behind its last MFMA.  `build`: there is a set to build;
  // `anext`: there is a next chunk; `lnext`: the first reads of the set built in the NEXT half may be issued at the end
  // (slot 1: the raw chunk it reads has landed; slot 0's reads wait for the mid-chunk barrier).
  auto half = [&](auto e_, auto z_, int k, bool build, bool anext) __attribute__((always_inline)) {
    constexpr int E = decltype(e_)::value;
    using EE = std::integral_constant<int, E>;
    using EN = std::integral_constant<int, E ^ 1>;
    using Z = std::integral_constant<bool, decltype(z_)::value>;
    using NZ = std::false_type;
    auto ld = [&](int p) __attribute__((always_inline)) {
      if constexpr (E == 0) load1(p); else load0(p);
    };
    mm(EE{}, I0{}, I0{}, I0{}, Z{});
    if (build) { wait_lds(); comb_split(EN{}, I0{}); ld(1); }
    fence();
    mm(EE{}, I0{}, I1{}, I0{}, Z{});
    mm(EE{}, I1{}, I0{}, I0{}, Z{});
    if (build) { wait_lds(); comb_split(EN{}, I1{}); ld(2); }
    fence();
    mm(EE{}, I1{}, I1{}, I0{}, Z{});
    if (anext) gldA(EE{}, I0{}, k + 1);
    fence();
    mm(EE{}, I0{}, I0{}, I2{}, NZ{});
    if (build) { wait_lds(); comb_split(EN{}, I2{}); ld(3); }
    fence();
    mm(EE{}, I0{}, I1{}, I2{}, NZ{});
    mm(EE{}, I1{}, I0{}, I2{}, NZ{});
    if (build) { wait_lds(); comb_split(EN{}, I3{}); }
    fence();
    mm(EE{}, I1{}, I1{}, I2{}, NZ{});
    if (anext) gldA(EE{}, I2{}, k + 1);
    fence();
    mm(EE{}, I0{}, I0{}, I1{}, NZ{});
    if (build) finalize_a(EN{});
    fence();
    mm(EE{}, I0{}, I1{}, I1{}, NZ{});
    if (build) finalize_b(EN{});
    fence();
    mm(EE{}, I1{}, I0{}, I1{}, NZ{});
    mm(EE{}, I1{}, I1{}, I1{}, NZ{});
    if (anext) gldA(EE{}, I1{}, k + 1);
    fence();
  };
  // One chunk.  At its top raw(k + 1) has landed (this wave's share: everything but the six newest loads -- the A fragments
  // of slot 1 for this chunk -- has returned; the barrier covers the other waves' shares) and every wave is through with
  // raw(k - 1) -- its last reader was slot 1 of chunk k - 1, built in that chunk's first half --, whose buffer takes
  // raw(k + 2): a whole chunk to land.  No barrier, DMA issue or exposed LDS latency sits between the two halves: the first
  // reads of the set the second half builds are issued at the end of the first (a stamped run of the two-buffer version, which
  // had all three there, measured the second half at 1500-1700 cycles against 790-1090 for the first).
  auto chunk = [&](auto z_, int k, bool has_next) __attribute__((always_inline)) {
    W4_FINE(0);
    if (has_next) {
      __builtin_amdgcn_s_waitcnt(w4_waitcnt(6, 15));
      W4_FINE(1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      W4_FINE(2);
      issue_raw(k + 2);
    }
    half(I0{}, z_, k, true, has_next);
    if (has_next) {
      load0(0);
      fence();
    }
    W4_FINE(3);
    half(I1{}, z_, k, has_next, has_next);
    if (has_next) {
      rotate(aA0, aB0, hbuf);
      rotate(aAp, aBp, lbuf);
      load1(0);
      fence();
    }
    W4_FINE(4);
    if (k < 30) W4_STAMP(3 + k);
  };

  // ---- prologue: two raw chunks and the first A fragments in flight; slot 0 of chunk 0 is built without MFMAs to hide under
  W4_STAMP(0);
#ifdef DVSR_CONV_TRACE
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memrealtime();
#endif
  issue_raw(0);
  issue_raw(1);   // (at least two chunks: conv2d_packed_prepare)
  gldA(I0{}, I0{}, 0); gldA(I0{}, I2{}, 0); gldA(I0{}, I1{}, 0);
  gldA(I1{}, I0{}, 0); gldA(I1{}, I2{}, 0); gldA(I1{}, I1{}, 0);
  __builtin_amdgcn_s_waitcnt(w4_waitcnt(NI + 12, 0));   // raw(0) (and the zero fill) of this wave
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  W4_STAMP(1);
  load0(0);
  wait_lds(); comb_split(I0{}, I0{}); load0(1); fence();
  wait_lds(); comb_split(I0{}, I1{}); load0(2); fence();
  wait_lds(); comb_split(I0{}, I2{}); load0(3); fence();
  wait_lds(); comb_split(I0{}, I3{});
  finalize_a(I0{}); finalize_b(I0{});
  rotate(aA0, aB0, hbuf);   // slot 0 of chunk 1 reads raw buffer 1
  load1(0);
  fence();
  W4_STAMP(2);

  // (a static s_setprio 1 for the second-dispatched half of the workgroup, which loses the VALU arbitration on its SIMD to the
  // older wave and arrives last at every barrier, measured equal: 104.3 / 181.3 us against 103.2 / 185.8 without)
  chunk(T{}, 0, true);
  for (int k = 1; k + 1 < a.nchunks; ++k) chunk(F{}, k, true);
  chunk(F{}, a.nchunks - 1, false);
  W4_STAMP(40);

  // ---- epilogue.  Y = A^T M A, A^T = [[1, 1, 1, 0], [0, 1, -1, -1]].  (the lane index passes through an opaque asm: nothing
  // of the epilogue's per-lane addressing can be hoisted above the K loop, where every register is spoken for)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int lo_e = lane_e & 31, hi_e = lane_e >> 5;
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] += acc[4 + b];
  const int tr_o = wave & 1, rq_o = wave >> 1;
  const int ttw = tr_o * 32 + lo_e;                      // this lane's tile
  const int orow = oy0 + 2 * (ttw / TC), ocol = ox0 + 2 * (ttw % TC);
  const size_t HWo = (size_t)a.Ho * a.Wo;
  const float slope = a.act == ACT_LRELU ? 0.1f : (a.act == ACT_RELU ? 0.f : 1.f);
  const float neg = a.gmask_act == ACT_LRELU ? 0.1f : (a.gmask_act == ACT_RELU ? 0.f : 1.f);
  const bool full = oy0 + Sh::OH <= a.Ho && ox0 + Sh::OW <= a.Wo && cbi * 64 + 64 <= a.Cout;
  // bias of the couts this lane finishes (wave (tr_o, rq_o): registers 4 rq_o .. + 3 of cout half R)
  float bk[2][4];
  {
    const float* bias = wset_ptr(a.bias, a.b_gs, n, a.wdiv);
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int co = cbi * 64 + R * 32 + 8 * rq_o + 4 * hi_e + k;
        bk[R][k] = bias ? bias[co < a.Cout ? co : a.Cout - 1] : 0.f;
      }
  }
  // exchange image: [source wave 8][slot 2][tr 2][rq 4][lane 64] x 16 B = 128 KB
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
  W4_STAMP(50);
  lds_barrier();   // every wave is past its last raw read
  W4_STAMP(51);
  auto write_round = [&](auto r_, w4f2 (&ex)[4][2]) __attribute__((always_inline)) {
    constexpr int R = decltype(r_)::value;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x16& m = acc[4 * e + 2 * R + tt];
          *reinterpret_cast<f32x4*>(xw + ((e * 2 + tt) * 4 + rq) * 256) = f32x4{m[4 * rq], m[4 * rq + 1], m[4 * rq + 2], m[4 * rq + 3]};
        }
    // residual / accumulate operands of the round, loaded under the exchange
    if (full && !plain && a.ps == 0) {
      const int cob = cbi * 64 + R * 32 + 8 * rq_o;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const size_t sb = (((size_t)n * a.Cout + cob + k) * HWo + (size_t)i * a.Wo) * 4;   // scalar
          w4f2 e = {0.f, 0.f};
          if (a.res) e = *reinterpret_cast<const w4f2*>(reinterpret_cast<const char*>(a.res) + sb + lane_off);
          if (a.accum) e += *reinterpret_cast<const w4f2*>(reinterpret_cast<const char*>(a.y) + sb + lane_off);
          ex[k][i] = e;
        }
    }
  };
  // y[i][j]: output row i, column j of the tile, four consecutive couts.  Source wave r' + 4 np' wrote (np' = 0) D_0, D_1
  // or (np' = 1) -D_1, D_0 of xi = r' into its slots 0, 1.
  auto read_round = [&](f32x4 (&y)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 s[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(xr + (rr * 16 + j * 8) * 256);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(xr + ((rr + 4) * 16 + (1 - j) * 8) * 256);
        s[rr] = j == 0 ? v0 + v1 : v0 - v1;
      }
      y[0][j] = s[0] + s[1] + s[2];
      y[1][j] = s[1] - s[2] - s[3];
    }
  };
  auto finish_round = [&](auto r_, f32x4 (&y)[2][2], w4f2 (&ex)[4][2]) __attribute__((always_inline)) {
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
      // (data-gradient launches: the activation mask of the producer.  Its eight loads go out TOGETHER, ahead of the round's
      // stores: read inside the store loop -- as rounds 4-5 had it, to save registers -- every load waited behind the previous
      // row's store, which the compiler may not reorder it with: eight dependent round trips per thread and round.)
      w4f2 gmv[4][2];
      if (full && a.gmask) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const size_t sb = (((size_t)n * a.Cout + cob + k) * HWo + (size_t)i * a.Wo) * 4;   // scalar
            gmv[k][i] = *reinterpret_cast<const w4f2*>(reinterpret_cast<const char*>(a.gmask) + sb + lane_off);
          }
      }
      if (!full) {
        // Edge tiles (44 x 80 is 5.5 x 2.5 tiles of 8 x 32: 44 % of the inner step's workgroups): every access goes through a
        // bounds-checked raw buffer of this image with the byte offset forced out of range for invalid (row, column, channel)
        // combinations, as in store_mfma_tile's edge path -- ALL loads of the round first, then the stores.  (Round 6: the
        // per-element `if (ok) { w += res[..]; w += y[..]; w *= mask[..]; y[..] = w; }` form was up to 48 dependent round
        // trips per thread and round.)
        const size_t img = (size_t)a.Cout * HWo;
        const __amdgpu_buffer_rsrc_t ry = image_rsrc(a.y + (size_t)n * img, img);
        const __amdgpu_buffer_rsrc_t rr = image_rsrc(a.res ? a.res + (size_t)n * img : a.y, a.res ? img : 0);
        const __amdgpu_buffer_rsrc_t rg = image_rsrc(a.gmask ? a.gmask + (size_t)n * img : a.y, a.gmask ? img : 0);
        const __amdgpu_buffer_rsrc_t ra = image_rsrc(a.y + (size_t)n * img, a.accum ? img : 0);
        unsigned off[4][2][2];
        float e[4][2][2], g[4][2][2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int co = cob + 4 * hi_e + k, oy = orow + i;
            const bool ok0 = co < a.Cout && oy < a.Ho && ocol < a.Wo;
            const unsigned o0 = (unsigned)(((size_t)co * HWo + (size_t)oy * a.Wo + ocol) * 4);
            off[k][i][0] = ok0 ? o0 : 0xFFFFFFFFu;
            off[k][i][1] = (ok0 && ocol + 1 < a.Wo) ? o0 + 4u : 0xFFFFFFFFu;
          }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              e[k][i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, off[k][i][j], 0, 0)) +
                           __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, off[k][i][j], 0, 0));
              g[k][i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, off[k][i][j], 0, 0));
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              float w = y[i][j][k] + e[k][i][j];
              w *= (a.gmask && g[k][i][j] <= 0.f) ? neg : 1.f;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, w), ry, off[k][i][j], 0, 0);
            }
      } else
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          w4f2 v = {y[i][0][k], y[i][1][k]};
          if (full) {
            const size_t sb = (((size_t)n * a.Cout + cob + k) * HWo + (size_t)i * a.Wo) * 4;   // scalar
            if (!plain) {
              v += ex[k][i];
              if (a.gmask) {
                const w4f2 m = gmv[k][i];
                v = w4f2{v[0] * (m[0] > 0.f ? 1.f : neg), v[1] * (m[1] > 0.f ? 1.f : neg)};
              }
            }
            *reinterpret_cast<w4f2*>(reinterpret_cast<char*>(a.y) + sb + lane_off) = v;
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
            else *reinterpret_cast<w4f2*>(dst) = w4f2{v[0], v[1]};
          }
        }
      }
    }
  };
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  f32x4 y0[2][2], y1[2][2];
  w4f2 ex0[4][2], ex1[4][2];
  write_round(R0{}, ex0);
  W4_STAMP(52);
  lds_barrier();
  W4_STAMP(53);
  read_round(y0);
  W4_STAMP(54);
  lds_barrier();   // the reads of the first round are done
  W4_STAMP(55);
  write_round(R1{}, ex1);
  W4_STAMP(56);
  finish_round(R0{}, y0, ex0);
  W4_STAMP(57);
  lds_barrier();
  read_round(y1);
  W4_STAMP(58);
  finish_round(R1{}, y1, ex1);
#ifdef DVSR_CONV_TRACE
  W4_STAMP(41);
  __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
  W4_STAMP(42);
  if (a.trace && threadIdx.x == 0) {
    a.trace[(size_t)blockIdx.x * 64 + 61] = __builtin_amdgcn_s_memrealtime();
    a.trace[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_ID
  }
#endif
}

// (Round 5 also built a WIDE form -- 128 couts x 32 tiles per workgroup for Cout >= 128, so that a B fragment fed four cout
// blocks.  Parity-green and 20 % SLOWER on every Cout >= 128 layer (L1_om 354 -> 425 us, profiles/r05_wino4_wide.txt): an A
// fragment then feeds ONE MFMA, 192 KB of weight fragments per CU and chunk through the vector memory path.  Retired in round 6;
// the F(4x4, 3x3) kernel, conv2d_wino5.hip, is what those layers run now.)

template <int TC>
static int launch_wino4(ConvK2 k, hipStream_t st) {
  using Sh = Wino4Shape<TC>;
  auto kern = conv2d_wino4_kernel<TC>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, Sh::LDS_BYTES);
  k.tiles_x = ceil_div(k.Wo, Sh::OW); k.tiles_y = ceil_div(k.Ho, Sh::OH); k.ntiles = k.tiles_x * k.tiles_y * k.N;
  k.ncb = ceil_div(k.Cout, 64);
  k.tiles_per_xcd = ceil_div(k.ntiles, 8);
  k.nitems = k.tiles_per_xcd * 8 * k.ncb;
  hipLaunchKernelGGL(kern, dim3(k.nitems), dim3(512), Sh::LDS_BYTES, st, k);
  return check_launch("conv2d_wino4_kernel");
}

// th = 4: 4 x 64-pixel workgroup tiles (TC = 32), th = 8: 8 x 32 (TC = 16), th = 16: 16 x 16 (TC = 8)
int conv2d_wino4_launch(const ConvK2& k, int th, hipStream_t st) {
  if (th == 16) return launch_wino4<8>(k, st);
  return th == 8 ? launch_wino4<16>(k, st) : launch_wino4<32>(k, st);
}

}  // namespace dvsr
