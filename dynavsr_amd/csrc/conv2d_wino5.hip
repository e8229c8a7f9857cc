// 3x3 / stride 1 / pad 1 convolution by Winograd F(4x4, 3x3) on the bf16 matrix pipe at fp32 accuracy (the exact 3-way operand
// split of conv2d_wino3.hip / conv2d_wino4.hip; EDVR_arch.py:254-313 is what is being computed) -- round 6, "form 5".
//
// Why.  Three rounds of re-scheduling F(2x2, 3x3) (forms 0-4) ended issue-bound: ~295 instructions per wave and chunk for 24
// MFMAs.  F(4x4, 3x3) changes the arithmetic: 36 transformed-domain points per 16 outputs instead of 16 per 4 -- MFMAs, packed
// weight traffic, transformed values to build and epilogue exchange bytes per output all x 0.5625.  Points 0, +-1, +-2, inf
// (Lavin & Gray's matrices): measured rel-L2 of a 64-channel layer against fp64 9e-7 (F(2x2): 1.4e-7, the direct fp32 sum
// 4.3e-7; oracle/winograd.py restates the transforms, tests/test_gpu_ops.py holds the kernel to 4e-6).
//
// Shape.  A workgroup owns 32 tiles of 4x4 outputs (TC tiles per tile row: 8 x 64 or 16 x 32 pixels) x 64 couts x 36 points
// = 1152 accumulator registers per lane -- the whole CU holds 2048, so one workgroup per CU: SIXTEEN waves of 128 registers,
// twelve that multiply and four that transform (the first version gave every wave both roles: its ten spilled registers were
// reloaded behind `s_waitcnt vmcnt(0)` in the chunk loop and serialised the DMA and the weight loads -- 142 us where this
// form takes 80):
//   * CONSUMER waves 0..11: wave (xi = wave % 6, mh = wave / 6) owns the six points (xi, nu = 0..5) of ONE 32-cout half: 6 x 16
//     accumulator registers, 18 MFMAs per 8-channel chunk, weight fragments from global memory (L2) through a ring of three
//     register slots, V fragments by 16-byte LDS reads two MFMAs ahead;
//   * PRODUCER waves 12..15: wave pw moves its quarter of the raw halo (16-byte buffer-load LDS-DMA, six instructions per
//     chunk; lanes outside the image carry an offset the resource rejects) and transforms all 8 channels of tiles 8 pw ..+7:
//     lane (tile, channel pair, parity) reads the six raw rows of its channel once, the row pass shares e +- o between rows
//     (1, 2) and (3, 4); per row pair (1, 2), (3, 4), (0, 5): column pass, v_permlane32_swap pairs the even and the odd
//     channel (lower lanes take the first row, upper lanes the second), the exact 3-way split (v_cvt_pk_bf16_f32) and 18
//     word stores -- ~430 instructions per chunk, none of them between a consumer and its MFMAs.
// The transformed input V goes THROUGH THE LDS as pure pieces, V[point][piece][tile][channel pair] (1536 bytes per point,
// 54 KB per chunk, two images) + two raw buffers of 24 KB: 159,744 bytes.  With 36 points the register-built operand of form 4
// does not fit, and the image decouples who transforms from who multiplies.
//
// Products.  x = hi + mid + lo for both operands, six of nine partial products (those above 2^-24), two per MFMA (K = 16 =
// 8 channels x 2 pieces, lanes 32-63 carry K 8..15).  Per point PAIR (nu even, nu odd) three 1 KB weight fragments instead of
// four: X = (Uh | Um) of the even point, X' = (Um | Uh) of the odd point, L = (Ul odd | Ul even); each X serves two MFMAs
//     X . (Vh | Vh) -> Uh Vh + Um Vh        X . (Vm | Vm) -> Uh Vm + Um Vm
// and is then blended with L by lane half (one v_cndmask per register) into W = (Uh | Ul) resp. W' = (Ul | Uh) for the third,
//     W . (Vl | Vh) -> Uh Vl + Ul Vh        W' . (Vh | Vl) -> Ul Vh + Uh Vl.
// The V fragments' lane halves simply point at different pieces.  Pack: P16[cb][k][xi][q = nu / 2][mh][X | X' | L][lane half]
// [cout 32][8 ch] (pack_weights_wino5_kernel: G g G^T in fp64, rounded once to fp32, split).
//
// Pipeline.  Chunk k: consumers multiply V(k) (image k & 1) while producers build V(k + 1) from raw(k + 1) into the other
// image and the LDS-DMA fetches raw(k + 2) into the raw buffer raw(k) has left; ONE barrier per chunk.
//
// Epilogue.  Y = A^T M A.  The six nu of a row are in one wave: the column transform (6 -> 4 values) is done in registers,
// then per cout half one exchange round through the LDS ([xi][j][register quad][lane] x 16 B = 96 KB), eight reader waves
// (tile, 2 couts) apply the row transform, bias, activation, residual and store 16-byte output rows (PixelShuffle(2): 32);
// a ragged last tile row stores only its rows inside the image.  Used for NO-GRAD forwards only (engine.hip: Op::geo_ng).
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

#ifdef DVSR_CONV_TRACE
// cycle stamps of the debug build (tools/wino5_trace.py): thread 0 of every workgroup
#define W5_STAMP(i)                                                                                       \
  do {                                                                                                    \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define W5_STAMP(i) \
  do {              \
  } while (0)
#endif

typedef float w5f2 __attribute__((ext_vector_type(2)));
typedef __bf16 w5bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 w5bf2 __attribute__((ext_vector_type(2)));
typedef unsigned w5u4 __attribute__((ext_vector_type(4)));

constexpr int W5_PCH = 18 * 2 * 3 * 256;   // fp32-sized slots of one packed (64-cout block, 8-channel chunk): 18 point pairs x 2 cout halves x 3 KB

template <int TC>
struct Wino5Shape {
  static constexpr int CC = 8, NTILE = 32, TRW = NTILE / TC, NT = 768;   // NT: 16-byte slots per channel quad of a raw chunk
  static constexpr int OH = 4 * TRW, OW = 4 * TC;        // output pixels of the workgroup tile
  static constexpr int IH = OH + 2, RP = OW + 8, GR = RP / 4;
  static constexpr int NG = CC * IH * GR;                // 16-byte groups of one chunk's raw halo image
  static constexpr int NI = (NG + NT - 1) / NT;
  static constexpr int RAWPAD = NI * NT * 4;             // floats of one raw buffer (every lane of every DMA has a slot)
  static constexpr int VIMG = 36 * 3 * 32 * 4;           // 32-bit words of one V image
  static constexpr int XCH = 6 * 16 * 64 * 4;            // floats of the epilogue's exchange image (96 KB)
  static constexpr int LOOP = 2 * RAWPAD + 2 * VIMG;
  static constexpr size_t LDS_BYTES = (size_t)(LOOP > XCH ? LOOP : XCH) * sizeof(float);
};

constexpr int w5_waitcnt(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

__device__ __forceinline__ unsigned w5_cvt_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(w5f2{a, b}, w5bf2));
}
__device__ __forceinline__ void w5_dma16(__amdgpu_buffer_rsrc_t rs, float* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w5_rsrc(const float* base, int num_records) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, num_records, 0x00020000);
}

// P16[cb][k][xi][q][mh][frag 3][lane half 2][cout 32][8 ch]: pieces of U = G g G^T, G of F(4x4, 3x3) (computed in fp64, rounded
// once to fp32, split exactly), for the point pair (xi, nu = 2 q) / (xi, 2 q + 1): frag 0 = X of the even point (hi | mid),
// frag 1 = X' of the odd point (mid | hi), frag 2 = L (lo of the odd point | lo of the even point).  One thread = one
// (cout, cin) pair.
__global__ void pack_weights_wino5_kernel(PackTable t) {
  const PackEntry& e = t.e[blockIdx.y];
  if (e.perm != 5) return;
  __bf16* const P16 = reinterpret_cast<__bf16*>(e.P);
  const size_t total = (size_t)e.ncb * e.nchunks * 512;   // (cout, cin) pairs incl. padding
  constexpr double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                              {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i & 7), col = (int)((i >> 3) & 63);
    const size_t ck = i >> 9;
    const int k = (int)(ck % e.nchunks), cb = (int)(ck / e.nchunks);
    const int co = cb * 64 + col, ci = k * 8 + c8;
    double g[3][3];
    const bool ok = co < e.Cout && ci < e.Ctot;
    const float* src = !e.wt ? e.w + ((size_t)co * e.Ctot + ci) * 9 : e.w + ((size_t)ci * e.w_ctot + e.w_coff + co) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) g[tap / 3][tap % 3] = ok ? (double)src[e.wt ? 8 - tap : tap] : 0.0;
    double c[6][3];
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int b = 0; b < 3; ++b) c[xi][b] = G[xi][0] * g[0][b] + G[xi][1] * g[1][b] + G[xi][2] * g[2][b];
    const int mh = col >> 5, c32 = col & 31;
    __bf16* dst = P16 + ck * (size_t)(2 * W5_PCH) + (size_t)mh * 1536 + (size_t)c32 * 8 + c8;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) {
        const float u = (float)(c[xi][0] * G[nu][0] + c[xi][1] * G[nu][1] + c[xi][2] * G[nu][2]);
        const __bf16 h = (__bf16)u;
        const float r1 = u - (float)h;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        __bf16* d = dst + (size_t)(xi * 3 + (nu >> 1)) * 3072;
        if (nu & 1) { d[512] = m; d[768] = h; d[1024] = l; }   // X' = (mid | hi); L lower lanes
        else { d[0] = h; d[256] = m; d[1280] = l; }           // X = (hi | mid); L upper lanes
      }
  }
}

int pack_weights_wino5_run(const PackTable& t, hipStream_t st) {
  hipLaunchKernelGGL(pack_weights_wino5_kernel, dim3(64, t.n), dim3(256), 0, st, t);
  return check_launch("pack_weights_wino5_kernel");
}

// RES: the epilogue adds a residual tensor (a.res != nullptr; plain stores only).  R16 (instantiated with RES): all sixteen
// waves read the exchange image, one cout each; otherwise eight consumer waves read a cout PAIR each -- what PixelShuffle(2)
// needs for 16-byte rows of the shuffled image, and what measures faster without a residual -- and the producers leave after
// the loop.
template <int TC, bool RES, bool R16>
__global__ __launch_bounds__(1024) void conv2d_wino5_kernel(ConvK2 a) {
  using Sh = Wino5Shape<TC>;
  constexpr int IH = Sh::IH, RP = Sh::RP, GR = Sh::GR;
  constexpr int CHF = IH * RP;              // floats between two channels of one quad of the raw image
  constexpr int QF = 768 * 4;               // floats between the two channel quads of a raw chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const raw0 = smem;                                         // two raw chunks
  unsigned* const vimg0 = reinterpret_cast<unsigned*>(smem + 2 * Sh::RAWPAD);   // two V images

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
  const size_t HW = (size_t)a.H * a.W;
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  W5_STAMP(0);
#ifdef DVSR_CONV_TRACE
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memrealtime();
#endif

  // R16 epilogue, one round: every wave reads the exchange image [xi 6][j 4][rq 4][lane 64] x 16 B -- lane' = the writer's lane
  // (tile, hi), register quad rq' = wave & 3, cout wave >> 2 of the quad --, applies the row transform, bias, activation and
  // residual and stores four 16-byte output rows.  (Eight readers with a cout pair each took 3.1-3.7 k cycles per round,
  // sixteen take 1.8 k.)
  auto read16 = [&](int m) __attribute__((always_inline)) {
    int le = lane;
    asm volatile("" : "+v"(le));   // (nothing of this addressing may be hoisted above the chunk loop)
    const int rq_r = wave & 3, cp_r = wave >> 2;
    const float* const xr = smem + (rq_r * 64 + le) * 4 + cp_r;
    const int lo_e = le & 31, hi_e = le >> 5;
    const int trow_e = lo_e / TC, tcol_e = lo_e - trow_e * TC;
    const int oy = oy0 + 4 * trow_e, ox = ox0 + 4 * tcol_e;
    const size_t HWo = (size_t)a.Ho * a.Wo;
    const float slope = a.act == ACT_LRELU ? 0.1f : (a.act == ACT_RELU ? 0.f : 1.f);
    const float* bias = wset_ptr(a.bias, a.b_gs, n, a.wdiv);
    const int co0 = cbi * 64 + m * 32 + 8 * rq_r + 4 * hi_e + cp_r;   // this thread's cout
    float y[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float z[6];
#pragma unroll
      for (int x = 0; x < 6; ++x) z[x] = xr[(x * 16 + j * 4) * 256];
      const float sa = z[1] + z[2], da = z[1] - z[2];
      const float sb = z[3] + z[4], db = z[3] - z[4];
      y[0][j] = z[0] + sa + sb;
      y[1][j] = __builtin_fmaf(2.f, db, da);
      y[2][j] = __builtin_fmaf(4.f, sb, sa);
      y[3][j] = __builtin_fmaf(8.f, db, da) + z[5];
    }
    const bool px_ok = oy < a.Ho && ox < a.Wo;   // (Wo is a multiple of 4: whole tile columns; the last tile row may be cut)
    const int nrow = a.Ho - oy < 4 ? a.Ho - oy : 4;
    if (px_ok && co0 < a.Cout) {
      const float b0 = bias ? bias[co0] : 0.f;
      const size_t base = ((size_t)n * a.Cout + co0) * HWo + (size_t)oy * a.Wo + ox;
      f32x4 rr[4];
      if constexpr (RES) {   // (the four residual rows together, ahead of the stores: see the pair reader below)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          rr[i] = i < nrow ? *reinterpret_cast<const f32x4*>(a.res + base + (size_t)i * a.Wo) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i >= nrow) break;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0 = y[i][j] + b0;
          v[j] = fmaxf(v0, v0 * slope);
        }
        if constexpr (RES) v += rr[i];
        *reinterpret_cast<f32x4*>(a.y + base + (size_t)i * a.Wo) = v;
      }
    }
  };

  if (wave >= 12) {
    // =================================================================================================================
    // PRODUCER waves 12..15: wave pw moves the raw halo (its quarter of the DMA) and transforms all 8 channels of the tiles
    // 8 pw .. 8 pw + 7 -- lane (tile, channel pair, parity) = one channel of one tile, ALL six rows xi: the six raw rows are read once, the row pass shares
    // e +- o between (1, 2) and (3, 4), then per row pair (1, 2), (3, 4), (0, 5): column pass, the lane-half exchange that
    // pairs the even and the odd channel, the exact 3-way split and 18 stores.  ~430 instructions per chunk, none of them
    // between a consumer and its MFMAs.
    const int pw = wave - 12, ptid = tid - 768;   // 0..255
    const float* x0n = a.x0 + (size_t)n * a.x0_bs;
    const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : x0n;
    // raw halo groups this lane moves: a chunk = 2 channel quads x 720 groups (channel 4, row, column group) = 3 DMA
    // instructions of the 256 producer lanes per quad; the second quad differs from the first by a SCALAR offset.  LDS float
    // index of (channel c, row iy, column x): (c >> 2) * QF + (c & 3) * CHF + iy * RP + x.
    static_assert(IH * GR * 4 == 720, "a channel quad is 720 groups");
    unsigned hoff[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int L = jj * 256 + ptid;
      const int cq = L / (IH * GR), rr = L - cq * (IH * GR);
      const int iy = rr / GR, g = rr - iy * GR;
      const int gy = oy0 - 1 + iy, gx = ox0 - 4 + 4 * g;
      const bool live = L < 4 * IH * GR;
      const bool ok = live && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      hoff[jj] = ok ? (unsigned)(((size_t)cq * HW + (size_t)gy * a.W + gx) * 4) : 0x80000000u;
      if (live && !ok) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) *reinterpret_cast<f32x4*>(raw0 + (bb >> 1) * Sh::RAWPAD + (bb & 1) * QF + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const unsigned chunk_bytes = (unsigned)(Sh::CC * HW * 4), quad_bytes = (unsigned)(4 * HW * 4);
    auto issue_raw = [&](int k) __attribute__((always_inline)) {   // raw chunk k -> buffer k & 1 (past the last chunk: nothing)
      const bool live = k < a.nchunks;
      const bool second = k * Sh::CC >= a.c0;  // only possible when c1 > 0; a chunk never straddles the two inputs
      const unsigned soff = live ? (unsigned)(second ? k - a.c0 / Sh::CC : k) * chunk_bytes : 0u;
      const __amdgpu_buffer_rsrc_t rs = w5_rsrc(second ? x1n : x0n, live ? 0x7fffffff : 0);
      float* dst = raw0 + (k & 1) * Sh::RAWPAD + 256 * pw;
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        w5_dma16(rs, dst + jj * 1024, hoff[jj], soff);
        w5_dma16(rs, dst + QF + jj * 1024, hoff[jj], soff + quad_bytes);
      }
    };
    // lane = (tile 8 pw + (lane & 7), channel pair (lane >> 3) & 3, parity lane >> 5): the 32 lanes of a store group write 32
    // consecutive words of the V image (tile x pair) -- no bank conflict; a wave transforms all 8 channels of 8 tiles
    const int ptile = 8 * pw + (lane & 7), ppair = (lane >> 3) & 3;
    const int trow_t = ptile / TC, tcol_t = ptile - trow_t * TC;
    const int pch = 2 * ppair + hi;
    // the patch's first aligned group in the raw buffer read next; the V words written next (lower lanes: the first row of a
    // row pair, upper lanes: the second), image of the chunk being built
    const float* prd = raw0 + (pch >> 2) * QF + (pch & 3) * CHF + 4 * trow_t * RP + 4 * tcol_t;
    unsigned* vw12 = vimg0 + (hi ? 2 : 1) * 6 * 384 + ptile * 4 + ppair;
    unsigned* vw34 = vimg0 + (hi ? 4 : 3) * 6 * 384 + ptile * 4 + ppair;
    unsigned* vw05 = vimg0 + (hi ? 5 : 0) * 6 * 384 + ptile * 4 + ppair;
    auto ld_row = [&](float (&d)[6], int row) __attribute__((always_inline)) {
      const float* p = prd + row * RP;
      const w5f2 l2 = *reinterpret_cast<const w5f2*>(p + 2);
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(p + 4);
      const w5f2 h2 = *reinterpret_cast<const w5f2*>(p + 8);
      d[0] = l2[1]; d[1] = m4[0]; d[2] = m4[1]; d[3] = m4[2]; d[4] = m4[3]; d[5] = h2[0];
    };
    // one row of B^T applied along a 6-vector: nu 0 and 5 single, (1, 2) and (3, 4) as e +- o
    auto colpass = [&](const float (&r)[6], float (&v)[6]) __attribute__((always_inline)) {
      v[0] = __builtin_fmaf(4.f, r[0], __builtin_fmaf(-5.f, r[2], r[4]));
      v[5] = __builtin_fmaf(4.f, r[1], __builtin_fmaf(-5.f, r[3], r[5]));
      const float e1 = __builtin_fmaf(-4.f, r[2], r[4]), o1 = __builtin_fmaf(-4.f, r[1], r[3]);
      v[1] = e1 + o1; v[2] = e1 - o1;
      const float e2 = r[4] - r[2], o2 = r[3] - r[1];
      v[3] = __builtin_fmaf(2.f, o2, e2); v[4] = __builtin_fmaf(-2.f, o2, e2);
    };
    // rows ra (lower lanes' row) and rb (upper lanes' row) of B^T d for this lane's channel -> column pass, pairing, split, stores
    auto finish_pair = [&](const float (&ra)[6], const float (&rb)[6], unsigned* vw) __attribute__((always_inline)) {
      float va[6], vb[6];
      colpass(ra, va);
      colpass(rb, vb);
#pragma unroll
      for (int v = 0; v < 6; ++v) {
        // lower lanes: (even, odd channel) of the first row; upper lanes: of the second row
        const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, va[v]), __builtin_bit_cast(unsigned, vb[v]), false, false);
        // (through unsigned temporaries: __builtin_bit_cast(float, s0[1]) on the vector ELEMENT reads element 0 -- hipcc 7.2)
        const unsigned u0 = s0[0], u1 = s0[1];
        const float v0 = __builtin_bit_cast(float, u0), v1 = __builtin_bit_cast(float, u1);
        const unsigned h = w5_cvt_pk(v0, v1);
        const float r0 = v0 - __builtin_bit_cast(float, h << 16), r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
        const unsigned m = w5_cvt_pk(r0, r1);
        const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
        unsigned* d = vw + v * 384;
#ifdef W5_NOVW   // (probe build, results wrong: one V word per point instead of three)
        d[0] = h ^ m ^ w5_cvt_pk(q0, q1);
#else
        d[0] = h; d[128] = m; d[256] = w5_cvt_pk(q0, q1);
#endif
      }
    };
    auto produce = [&]() __attribute__((always_inline)) {
      float d0[6], d1[6], d2[6], d3[6], d4[6], d5[6];
      ld_row(d1, 1); ld_row(d2, 2); ld_row(d3, 3); ld_row(d4, 4);
      ld_row(d0, 0); ld_row(d5, 5);
      float ra[6], rb[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float e = __builtin_fmaf(-4.f, d2[j], d4[j]), o = __builtin_fmaf(-4.f, d1[j], d3[j]);
        ra[j] = e + o; rb[j] = e - o;
      }
      finish_pair(ra, rb, vw12);
      fence();
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float e = d4[j] - d2[j], o = d3[j] - d1[j];
        ra[j] = __builtin_fmaf(2.f, o, e); rb[j] = __builtin_fmaf(-2.f, o, e);
      }
      finish_pair(ra, rb, vw34);
      fence();
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        ra[j] = __builtin_fmaf(4.f, d0[j], __builtin_fmaf(-5.f, d2[j], d4[j]));
        rb[j] = __builtin_fmaf(4.f, d1[j], __builtin_fmaf(-5.f, d3[j], d5[j]));
      }
      finish_pair(ra, rb, vw05);
    };
    // ---- prologue: two raw chunks in flight, V(0) built
    issue_raw(0);
    issue_raw(1);
    __builtin_amdgcn_s_waitcnt(w5_waitcnt(6, 0));   // raw(0) (and the zero fill) of this wave
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    produce();
    __builtin_amdgcn_s_waitcnt(w5_waitcnt(0, 0));   // raw(1)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int vstep = Sh::VIMG, rstep = Sh::RAWPAD;
    vw12 += vstep; vw34 += vstep; vw05 += vstep; prd += rstep;   // on to raw(1) -> V(1)
    // (the four producer waves are the last dispatched of the workgroup and lose every issue arbitration to the twelve consumers,
    // which have slack: above them, 90.7 -> 86.5 us on the 8-chunk layer)
    __builtin_amdgcn_s_setprio(3);
    // ---- chunk k: raw(k + 2) into the buffer raw(k) has left, V(k + 1) from raw(k + 1)
    for (int k = 0; k < a.nchunks; ++k) {

#ifndef W5_NODMA   // (probe build, results wrong: the chunk loop fetches no raw halo)
      issue_raw(k + 2);
#endif
#ifndef W5_NOPROD  // (probe build, results wrong: no transform work inside the chunk loop)
      if (k + 1 < a.nchunks) produce();
#endif
      __builtin_amdgcn_s_waitcnt(w5_waitcnt(0, 0));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      vw12 -= vstep; vw34 -= vstep; vw05 -= vstep; prd -= rstep;
      vstep = -vstep; rstep = -rstep;
    }
    if constexpr (R16) {
      // (the producers hold no results: they are four of the epilogue's sixteen reader waves)
      __builtin_amdgcn_s_setprio(0);
      lds_barrier();                 // every wave is past its last V read
      lds_barrier(); read16(0);      // round 0 written / read
      lds_barrier();                 // ... its reads done
      lds_barrier(); read16(1);      // round 1
    } else {
      // (the epilogue's four barriers; the producers hold no results)
      lds_barrier(); lds_barrier(); lds_barrier(); lds_barrier();
    }
    return;
  }
  {
    // =================================================================================================================
    // CONSUMER waves 0..11: wave (xi, mh) multiplies the six points (xi, nu) of one 32-cout half: 18 MFMAs per chunk, their
    // weight fragments straight from global memory (a ring of three: one in use, two in flight), their V fragments from the
    // LDS image (two buffers).  Nothing else: ~100 instructions per chunk, so a late fragment stalls nothing but its MFMA.
    const int cxi = wave % 6, cmh = wave / 6;
    f32x16 acc[6];
#pragma unroll
    for (int v = 0; v < 6; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
    const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + (size_t)cbi * a.nchunks * W5_PCH;
    const __amdgpu_buffer_rsrc_t wrsrc = w5_rsrc(wp_cb, -1);
    const unsigned avoff = (unsigned)(lane * 16);
    const int asb = (cxi * 3 * 2 + cmh) * 3072;      // byte offset of (point pair (xi, 0), mh) inside a chunk of the pack
    // Three weight fragments per point PAIR (2 q, 2 q + 1): X = (Uh | Um) of the even point, X' = (Um | Uh) of the odd one and
    // L = (Ul odd | Ul even).  After the two MFMAs that read X its upper lanes take L's (W = (Uh | Ul)); after the two that
    // read X' its lower lanes do (W' = (Ul | Uh), multiplied with the MIRRORED fragment (Vh | Vl)): 3 KB instead of 4 KB per
    // pair and cout half cross the vector memory path, which is what bounds the chunk loop (profiles/r06_wino5_*.txt).
    // Fragment g = 3 q + (0: X, 1: X', 2: L) of a chunk lives in slot g % 3: nine per chunk, one in use, two in flight.
    f32x4 AF[3];
    auto gldA = [&](auto g_, int k) __attribute__((always_inline)) {   // fragment G (0..8) of chunk k
      constexpr int G = decltype(g_)::value;
#ifdef W5_NOA   // (probe build, results wrong: only the first fragments are loaded -- what do the weight loads cost?)
      if (k > 0 || G > 2) return;
#endif
      const int soff = k * (W5_PCH * 4) + asb + (G / 3) * 6144 + (G % 3) * 1024;
      AF[G % 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)avoff, soff, 0));
    };
    // x <- (x in the lanes where keep, l elsewhere)
    auto blend = [&](f32x4& x, const f32x4& l, bool keep) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = keep ? x[r] : l[r];
    };
    // V fragments: (point (xi, 0), piece 0, tile lo) of the image being read; the (Vl | Vh) fragment: lower lanes piece 2,
    // upper lanes piece 0.  The pointers move to the other image at the end of every chunk (+- VIMG).
    const unsigned* vrd = vimg0 + cxi * 6 * 384 + lo * 4;
    const unsigned* vrd_lh = vrd + (hi ? 0 : 256);   // even points: (Vl | Vh)
    const unsigned* vrd_hl = vrd + (hi ? 256 : 0);   // odd points:  (Vh | Vl)
    // two buffers, fragment f = 3 nu + (0: Vh | Vh, 1: Vm | Vm, 2: Vl | Vh) in buffer f & 1 -- one in use, one landing
    w5u4 BF[2];
    auto ldB = [&](auto f_) __attribute__((always_inline)) {
      constexpr int F = decltype(f_)::value, NU = F / 3, W = F % 3;
      if constexpr (W == 0) BF[F & 1] = *reinterpret_cast<const w5u4*>(vrd + NU * 384);
      else if constexpr (W == 1) BF[F & 1] = *reinterpret_cast<const w5u4*>(vrd + NU * 384 + 128);
      else if constexpr (NU & 1) BF[F & 1] = *reinterpret_cast<const w5u4*>(vrd_hl + NU * 384);
      else BF[F & 1] = *reinterpret_cast<const w5u4*>(vrd_lh + NU * 384);
    };
    auto mma = [&](f32x16& c, const f32x4& av, const w5u4& bv) __attribute__((always_inline)) {
#ifdef W5_NOMMA   // (probe build, results wrong: the operands are consumed by a cheap VALU op instead of the MFMA)
      c[0] += av[0] * __builtin_bit_cast(float, bv[0]);
      return;
#endif
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(w5bf8, av), __builtin_bit_cast(w5bf8, bv), c, 0, 0, 0);
    };
    static_for<0, 3>([&](auto g_) __attribute__((always_inline)) { gldA(g_, 0); });
    __builtin_amdgcn_s_barrier();   // raw(0) landed (the producers' business)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();   // V(0) built
    asm volatile("" ::: "memory");
    int vstep = Sh::VIMG;
    for (int k = 0; k < a.nchunks; ++k) {
      ldB(std::integral_constant<int, 0>{}); ldB(std::integral_constant<int, 1>{});
      const int knext = k + 1 < a.nchunks ? k + 1 : k;   // (past the last chunk: a harmless reload, the same loads on every path)
      static_for<0, 18>([&](auto f_) __attribute__((always_inline)) {
        constexpr int F = decltype(f_)::value, NU = F / 3, I = F % 3;
        constexpr int XS = NU & 1;   // the slot of this point's X / X' (fragments 3 q and 3 q + 1; L in slot 2)
        mma(acc[NU], AF[XS], BF[F & 1]);
#ifdef W5_NOBR   // (probe build, results wrong: the two V fragments read at the chunk top serve all 18 MFMAs)
        if constexpr (false)
#else
        if constexpr (F + 2 < 18)
#endif
          ldB(std::integral_constant<int, (F + 2 < 18 ? F + 2 : 0)>{});
        if constexpr (I == 1) {
          // X has been read twice: the lane half that held the mid pieces takes the lo pieces (even point: the upper half)
          blend(AF[XS], AF[2], (NU & 1) ? hi != 0 : hi == 0);
          // an odd point was L's last reader: its slot takes the next pair's L
          if constexpr ((NU & 1) && NU < 5) gldA(std::integral_constant<int, 3 * ((NU < 5 ? NU : 1) / 2) + 5>{}, k);
          if constexpr (NU == 5) gldA(std::integral_constant<int, 2>{}, knext);
        }
        if constexpr (I == 2) {
          // the X slot takes the X of the next pair (same parity)
          if constexpr (NU < 4) gldA(std::integral_constant<int, 3 * ((NU < 4 ? NU : 0) / 2) + 3 + (NU & 1)>{}, k);
          else gldA(std::integral_constant<int, (NU & 1)>{}, knext);
        }
        fence();
      });
      // every V fragment of the image has been read; the barrier frees it for V(k + 2) and hands V(k + 1) over
      lds_barrier();
      if (k < 16) W5_STAMP(3 + k);
      vrd += vstep; vrd_lh += vstep; vrd_hl += vstep;
      vstep = -vstep;
    }
  W5_STAMP(40);

  // ---- epilogue.  A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]].  Column transform in registers:
  // Z_j = sum_nu M[nu] A^T[j][nu]
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));   // (nothing of the epilogue's addressing may be hoisted above the chunk loop)
  f32x16 Z[4];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float sa = acc[1][r] + acc[2][r], da = acc[1][r] - acc[2][r];
    const float sb = acc[3][r] + acc[4][r], db = acc[3][r] - acc[4][r];
    Z[0][r] = acc[0][r] + sa + sb;
    Z[1][r] = __builtin_fmaf(2.f, db, da);
    Z[2][r] = __builtin_fmaf(4.f, sb, sa);
    Z[3][r] = __builtin_fmaf(8.f, db, da) + acc[5][r];
  }
  // exchange image: [xi 6][j 4][rq 4][lane 64] x 16 B; reader (waves 0-7): lane' = writer lane (tile, hi), rq' = wave & 3,
  // cout pair cp = wave >> 2 of the register quad
  float* const xw = smem + (cxi * 16 * 64 + lane_e) * 4;
  const int rq_r = wave & 3, cp_r = wave >> 2;
  const float* const xr = smem + (rq_r * 64 + lane_e) * 4 + cp_r * 2;
  const int lo_e = lane_e & 31, hi_e = lane_e >> 5;
  const int trow_e = lo_e / TC, tcol_e = lo_e - trow_e * TC;
  const int oy = oy0 + 4 * trow_e, ox = ox0 + 4 * tcol_e;
  const size_t HWo = (size_t)a.Ho * a.Wo;
  const float slope = a.act == ACT_LRELU ? 0.1f : (a.act == ACT_RELU ? 0.f : 1.f);
  const float* bias = wset_ptr(a.bias, a.b_gs, n, a.wdiv);
  W5_STAMP(50);
  lds_barrier();   // every wave is past its last V read
  W5_STAMP(51);
#pragma unroll 1
  for (int m = 0; m < 2; ++m) {
    if (cmh == m) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
          *reinterpret_cast<f32x4*>(xw + (j * 4 + rq) * 256) = f32x4{Z[j][4 * rq], Z[j][4 * rq + 1], Z[j][4 * rq + 2], Z[j][4 * rq + 3]};
    }
    W5_STAMP(52 + 4 * m);
    lds_barrier();
    W5_STAMP(53 + 4 * m);
    if constexpr (R16) read16(m);
    else if (wave < 8) {
      const int co0 = cbi * 64 + m * 32 + 8 * rq_r + 4 * hi_e + 2 * cp_r;   // this thread's two couts: co0, co0 + 1
      float y[2][4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w5f2 z[6];
#pragma unroll
        for (int x = 0; x < 6; ++x) z[x] = *reinterpret_cast<const w5f2*>(xr + (x * 16 + j * 4) * 256);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float sa = z[1][c] + z[2][c], da = z[1][c] - z[2][c];
          const float sb = z[3][c] + z[4][c], db = z[3][c] - z[4][c];
          y[c][0][j] = z[0][c] + sa + sb;
          y[c][1][j] = __builtin_fmaf(2.f, db, da);
          y[c][2][j] = __builtin_fmaf(4.f, sb, sa);
          y[c][3][j] = __builtin_fmaf(8.f, db, da) + z[5][c];
        }
      }
      const bool px_ok = oy < a.Ho && ox < a.Wo;   // (Wo is a multiple of 4: whole tile columns; the last tile row may be cut)
      const int nrow = a.Ho - oy < 4 ? a.Ho - oy : 4;
      if (px_ok && co0 < a.Cout) {
        const bool two = co0 + 1 < a.Cout;
        const float b0 = bias ? bias[co0] : 0.f, b1 = (bias && two) ? bias[co0 + 1] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v0 = y[0][i][j] + b0, v1 = y[1][i][j] + b1;
            y[0][i][j] = fmaxf(v0, v0 * slope);
            y[1][i][j] = fmaxf(v1, v1 * slope);
          }
        if (a.ps == 0) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if (c == 1 && !two) break;
            const size_t base = ((size_t)n * a.Cout + co0 + c) * HWo + (size_t)oy * a.Wo + ox;
            // (the residual rows of a cout are fetched TOGETHER, ahead of its stores: res may be the output buffer itself, so
            // the compiler keeps every load behind the previous row's store -- eight dependent round trips per thread and
            // round, 15-20 us of the 5 x 64 x 180 x 320 layers with a residual.  Every element is read and written by this
            // thread only, so reading a cout's four rows before writing them is the same computation.  A separate
            // instantiation: in one kernel with the plain form the allocator spilled three more registers and every layer
            // WITHOUT a residual lost 4 %.)
            if constexpr (RES) {
              f32x4 rr[4];
#pragma unroll
              for (int i = 0; i < 4; ++i)
                rr[i] = i < nrow ? *reinterpret_cast<const f32x4*>(a.res + base + (size_t)i * a.Wo) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (i >= nrow) break;
                const f32x4 v = f32x4{y[c][i][0], y[c][i][1], y[c][i][2], y[c][i][3]} + rr[i];
                *reinterpret_cast<f32x4*>(a.y + base + (size_t)i * a.Wo) = v;
              }
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (i >= nrow) break;
                *reinterpret_cast<f32x4*>(a.y + base + (size_t)i * a.Wo) = f32x4{y[c][i][0], y[c][i][1], y[c][i][2], y[c][i][3]};
              }
            }
          }
        } else {
          // PixelShuffle(2): couts 4 q + 2 dy + dx; this thread holds dy = cp_r, dx = 0, 1 of channel q = co0 >> 2
          const int cq = co0 >> 2, dy = (co0 >> 1) & 1;
          float* dst = a.y + (((size_t)n * (a.Cout >> 2) + cq) * (2 * a.Ho) + (2 * oy + dy)) * (size_t)(2 * a.Wo) + 2 * ox;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i >= nrow) break;
            float* d = dst + (size_t)(2 * i) * (2 * a.Wo);
            *reinterpret_cast<f32x4*>(d) = f32x4{y[0][i][0], y[1][i][0], y[0][i][1], y[1][i][1]};
            *reinterpret_cast<f32x4*>(d + 4) = f32x4{y[0][i][2], y[1][i][2], y[0][i][3], y[1][i][3]};
          }
        }
      }
    }
    W5_STAMP(54 + 4 * m);
    if (m == 0) lds_barrier();   // the reads of the first round are done
    W5_STAMP(55 + 4 * m);
  }
#ifdef DVSR_CONV_TRACE
  W5_STAMP(41);
  __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
  W5_STAMP(42);
  if (a.trace && threadIdx.x == 0) {
    a.trace[(size_t)blockIdx.x * 64 + 61] = __builtin_amdgcn_s_memrealtime();
    a.trace[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_ID
    a.trace[(size_t)blockIdx.x * 64 + 62] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // XCC_ID
  }
#endif
  }
}

template <int TC>
static int launch_wino5(ConvK2 k, hipStream_t st) {
  using Sh = Wino5Shape<TC>;
  // Sixteen readers only where the epilogue also fetches a residual (fe_rb_b 96-99 -> 91-92 us on one box); without one the
  // eight pair readers measure 2-3 % FASTER per launch (L1_om 238 against 243 us, HRconv 222-226 against 229-233), and a
  // PixelShuffle(2) launch (never with a residual: conv2_prepare) needs the pairs for its 16-byte rows.
  auto kern = k.res ? conv2d_wino5_kernel<TC, true, true> : conv2d_wino5_kernel<TC, false, false>;
  static PerDeviceOnce attr_once, attr_once_r;
  set_dyn_lds_once(attr_once, (const void*)conv2d_wino5_kernel<TC, false, false>, Sh::LDS_BYTES);
  set_dyn_lds_once(attr_once_r, (const void*)conv2d_wino5_kernel<TC, true, true>, Sh::LDS_BYTES);
  k.tiles_x = ceil_div(k.Wo, Sh::OW); k.tiles_y = ceil_div(k.Ho, Sh::OH); k.ntiles = k.tiles_x * k.tiles_y * k.N;
  k.ncb = ceil_div(k.Cout, 64);
  k.tiles_per_xcd = ceil_div(k.ntiles, 8);
  k.nitems = k.tiles_per_xcd * 8 * k.ncb;
  hipLaunchKernelGGL(kern, dim3(k.nitems), dim3(1024), Sh::LDS_BYTES, st, k);
  return check_launch("conv2d_wino5_kernel");
}

// th = 8: 8 x 64-pixel workgroup tiles (2 x 16 tiles of 4 x 4), th = 16: 16 x 32 (4 x 8 tiles)
int conv2d_wino5_launch(const ConvK2& k, int th, hipStream_t st) {
  return th == 16 ? launch_wino5<8>(k, st) : launch_wino5<16>(k, st);
}

}  // namespace dvsr
