// 3x3 / stride 1 / pad 1 convolution by Winograd's minimal filtering F(2x2, 3x3), fp32, on v_mfma_f32_32x32x2_f32.
//
// The dense 3x3 convolutions are ~90 % of EDVR's FLOPs (EDVR_arch.py:254-313) and the exact-fp32 matrix pipe tops
// out at 157 TFLOP/s; conv2d_dma_kernel sits at 0.75-0.79 of that on the big layers and two rounds of work on its
// pipeline moved it by a few percent.  This kernel spends 2.25x fewer multiplies instead: every 2x2 block of output
// pixels ("tile") of a (cout, cin) pair costs 16 multiplies in the transformed domain instead of 36,
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A,      g: 3x3 filter, d: 4x4 input patch, Y: 2x2 outputs,
// which turns the layer into SIXTEEN independent GEMMs (one per position xn = (xi, nu) of the 4x4 transformed patch),
//     M[xn][cout][tile] = sum_cin U[xn][cout][cin] V[xn][cin][tile],
// with U = G g G^T computed once per weight update (pack_weights_wino_kernel) and V = B^T d B computed here, per
// 8-channel chunk, from the raw halo tile.  Same arithmetic class as the fp32 path of the vendor libraries the
// reference runs on (cuDNN's WINOGRAD algorithms for the nn.Conv2d calls of EDVR_arch.py); the transforms of F(2x2, 3x3)
// only add/subtract and halve, the result differs from the direct sum by a few fp32 ulps (tests: rel-L2 <= 2e-6 against
// the direct kernel, the network-level goldens unchanged).
//
// Work split (one workgroup = 8 waves = one CU; the accumulators are the whole register file: 16 xn x 64 couts x 64 tiles):
//   * workgroup tile = 64 couts x 64 tiles (TC tile columns x 64/TC tile rows: 4x64 or 8x32 output pixels);
//   * wave (mh, tr, xh) = 32 couts x 32 tiles x EIGHT xn (rows xi in {0,1} or {2,3} of the transformed patch): acc[8] x
//     f32x16, two waves per SIMD.  (Four waves x 16 xn was built first: with ONE wave on a SIMD every dependent VALU chain
//     and LDS round trip of the transforms stalls the MFMA stream -- 174 us on a layer this design does in 132.)
//     Per chunk and pair of xn: one ds_read_b128 of U per xn (4 k-steps) and four 4-byte reads of V (ds_read2st64_b32),
//     two independent accumulator chains alternating, operands one pair ahead;
//   * per chunk the workgroup moves U(k+1) (32 KB, the packed LDS image) and the raw halo of chunk k+2 global -> LDS by
//     buffer-load DMA (16 B per lane; per-lane offset fixed, chunk offsets scalar), transforms the raw halo of chunk k+1
//     into V (wave = channel, lane = tile: three aligned 8-byte reads per patch row, 32 adds, sixteen conflict-free 4-byte
//     stores into V[xn][channel][tile]) and runs its 32 MFMAs; ONE barrier per chunk, in front of the last MFMA pair, whose
//     MFMAs cover the LDS round trip of the next chunk's first operands.
// LDS: 2 x (U 32 KB + V 32 KB + raw <= 13.5 KB) = 155 KB of the CU's 160 KB.
// Epilogue: Y = A^T M A is linear in the rows of M: each wave reduces its two rows to a partial 2x2 output per (cout,
// tile) in registers (packed over cout pairs), the two waves of a pair swap halves through the idle LDS buffers and each
// finishes 8 of the 16 cout registers: bias / activation / residual / accumulate / gradient mask / PixelShuffle(2) exactly
// as store_mfma_tile does for the direct kernels.  Design notes and measurements: DESIGN.md 3.1e.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

// measurement aid of the debug build (DVSR_CONV_ABLATE, results are WRONG when set): bit 0 no epilogue, bit 1 no input
// transform, bit 2 no DMA, bit 3 operands read once, bit 4 no chunk barriers
// (and tools/wino_trace.py: thread 0 of every workgroup stamps s_memtime at the phase boundaries)
#ifdef DVSR_CONV_TRACE
#define WINO_ABLATE(a) ((a).ablate)
#define WINO_STAMP(i)                                                                                     \
  do {                                                                                                    \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define WINO_ABLATE(a) 0
#define WINO_STAMP(i) \
  do {                \
  } while (0)
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));

// P[cb][k][((xn*2 + hi)*64 + co_l)*4 + j] = (G g G^T)[xi][nu] of (cout = cb*64 + co_l, cin = k*8 + 2j + hi), xn = 4 xi + nu.
// One thread = one (cout, cin) pair: nine contiguous weights in, sixteen transformed values out (a wave = 8 couts x 8
// channels writes two 128-byte runs per xn).
__global__ void pack_weights_wino_kernel(PackTable t) {
  const PackEntry& e = t.e[blockIdx.y];
  if (e.perm != 3) return;
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
    // rows of G on the columns of g (c[xi][b]), then on the rows of the result: same operation order as before
    float c[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      c[0][b] = g[0][b];
      c[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      c[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      c[3][b] = g[2][b];
    }
    float* dst = e.P + ck * 8192 + ((size_t)(c8 & 1) * 64 + col) * 4 + (c8 >> 1);
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      dst[(xi * 4 + 0) * 512] = c[xi][0];
      dst[(xi * 4 + 1) * 512] = 0.5f * (c[xi][0] + c[xi][1] + c[xi][2]);
      dst[(xi * 4 + 2) * 512] = 0.5f * (c[xi][0] - c[xi][1] + c[xi][2]);
      dst[(xi * 4 + 3) * 512] = c[xi][2];
    }
  }
}

int pack_weights_wino_run(const PackTable& t, hipStream_t st) {
  hipLaunchKernelGGL(pack_weights_wino_kernel, dim3(64, t.n), dim3(256), 0, st, t);
  return check_launch("pack_weights_wino_kernel");
}

// 16 bytes per lane global -> LDS: buffer load with the per-lane byte offset in a VGPR and everything that changes per chunk in
// the scalar offset (no vector instruction per transfer).  A plain function: the builtin inside the kernel TEMPLATE makes the
// host pass drop the kernel's stub.
__device__ __forceinline__ void wino_dma16(const float* base, float* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000),
                                           (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

template <int TC>
struct WinoShape {
  static constexpr int CC = 8, NTILE = 64, TRW = NTILE / TC;
  static constexpr int OH = 2 * TRW, OW = 2 * TC;      // output pixels of the workgroup tile
  static constexpr int IH = OH + 2, RP = OW + 8, GR = RP / 4;
  static constexpr int NG = CC * IH * GR;              // 16-byte groups of one chunk's raw halo image
  static constexpr int NI = (NG + 511) / 512;
  static constexpr int RAW_FLOATS = NG * 4;
  static constexpr int UV = 16 * 2 * 64 * 4;           // floats of one U (or V) image
  static constexpr size_t LDS_BYTES = (size_t)(4 * UV + 2 * RAW_FLOATS) * sizeof(float);
};

template <int TC>
__global__ __launch_bounds__(512, 2) void conv2d_wino_kernel(ConvK2 a) {
  using Sh = WinoShape<TC>;
  constexpr int IH = Sh::IH, RP = Sh::RP, GR = Sh::GR, NI = Sh::NI, UV = Sh::UV;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_u0 = smem;
  float* const s_v0 = smem + 2 * UV;
  float* const s_r0 = smem + 4 * UV;

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
  const int mh = wave & 1, tr = (wave >> 1) & 1, xh = wave >> 2;
  const size_t HW = (size_t)a.H * a.W;
  const float* x0n = a.x0 + (size_t)n * a.x0_bs;
  const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : x0n;

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

  f32x16 acc[8];   // xn = 8 xh + i (first written by the MFMAs of chunk 0)

  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + (size_t)cbi * a.nchunks * UV;

  // DMA addressing: buffer loads take (resource, per-lane byte offset, scalar byte offset) -- the per-lane part is fixed
  // for the life of the workgroup, the chunk / piece part is scalar arithmetic: no vector instruction per transfer.
  const unsigned uoff = (unsigned)(lane * 16 + wave * 1024);
  const unsigned chunk_bytes = (unsigned)(Sh::CC * HW * 4);
  auto issue_raw_piece = [&](int k, int buf, int jj) {
    const int cbase = k * Sh::CC;
    const bool second = cbase >= a.c0;  // only possible when c1 > 0; a chunk never straddles the two inputs
    const unsigned soff = (unsigned)((second ? k - a.c0 / Sh::CC : k)) * chunk_bytes;
    float* dst = s_r0 + buf * Sh::RAW_FLOATS;
    if (hval[jj]) {
      if (second)
        wino_dma16(x1n, dst + 256 * (wave + 8 * jj), hoff[jj], soff);
      else
        wino_dma16(x0n, dst + 256 * (wave + 8 * jj), hoff[jj], soff);
    }
  };
  auto issue_raw = [&](int k, int buf) {
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) issue_raw_piece(k, buf, jj);
  };
  auto issue_u_piece = [&](int k, int buf, int j) {   // j = 0..3: 8 KB each
    float* wdst = s_u0 + buf * UV;
    wino_dma16(wp_cb, wdst + (j * 8 + wave) * 256, uoff,
                                             (unsigned)(k * (UV * 4) + j * 8192));
  };
  auto issue_u = [&](int k, int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_u_piece(k, buf, j);
  };

  // input transform: wave = channel of the chunk, lane = tile.  A patch row is read as three aligned 8-byte pairs (columns
  // 2 tc + 2 .. 2 tc + 7 of the raw image, conflict-free; the patch is columns + 3 .. + 6), the row pass B^T d runs on the
  // pairs, the column pass is scalar; V[xn][channel][tile] takes 4-byte stores, consecutive lanes consecutive words.
  const int trow_t = lane / TC, tcol_t = lane - trow_t * TC;
  const int roff = (wave * IH + 2 * trow_t) * RP + 2 * tcol_t + 2;
  const int voff = wave * 64 + lane;
  f32x2 td[4][3];
  auto tf_load = [&](int rbuf) {
    const float* ra = s_r0 + rbuf * Sh::RAW_FLOATS + roff;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p2 = 0; p2 < 3; ++p2) td[i][p2] = *reinterpret_cast<const f32x2*>(ra + i * RP + 2 * p2);
  };
  auto tf_rows = [&]() {
#pragma unroll
    for (int p2 = 0; p2 < 3; ++p2) {
      const f32x2 d0 = td[0][p2], d1 = td[1][p2], d2 = td[2][p2], d3 = td[3][p2];
      td[0][p2] = d0 - d2;
      td[1][p2] = d1 + d2;
      td[2][p2] = d2 - d1;
      td[3][p2] = d1 - d3;
    }
  };
  auto tf_cols = [&](int i, int vbuf) {   // row i of (B^T d) B -> V[4 i .. 4 i + 3]
    float* v = s_v0 + vbuf * UV + voff;
    const float c0 = td[i][0][1], c1 = td[i][1][0], c2 = td[i][1][1], c3 = td[i][2][0];
    v[(i * 4 + 0) * 512] = c0 - c2;
    v[(i * 4 + 1) * 512] = c1 + c2;
    v[(i * 4 + 2) * 512] = c2 - c1;
    v[(i * 4 + 3) * 512] = c1 - c3;
  };
  auto transform = [&](int rbuf, int vbuf) {
    tf_load(rbuf);
    tf_rows();
    tf_cols(0, vbuf); tf_cols(1, vbuf); tf_cols(2, vbuf); tf_cols(3, vbuf);
  };

  // bias: every output of a tile receives M[1][1] with weight one (column 1 of A^T is (1, 1)), so the bias is the INITIAL value
  // of xn = 5 (a wave of the xh = 0 half); the other accumulators start from the MFMA's inline zero
  const int co_block = cbi * 64 + mh * 32;
  f32x16 cinit5;
  {
    const float* bias = wset_ptr(a.bias, a.b_gs, n, a.wdiv);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_block + (r & 3) + 8 * (r >> 2) + 4 * hi;
      cinit5[r] = (bias && xh == 0) ? bias[co < a.Cout ? co : a.Cout - 1] : 0.f;
    }
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // One chunk = 4 steps of two xn (two independent accumulator chains alternate), the operands of the next pair in flight.
  // The next chunk's weight / halo DMA and its input transform are issued in steps 0..2; the chunk barrier sits BEFORE the
  // last step, whose MFMAs then cover the LDS round trip of the next chunk's first operands (after the barrier no wave reads
  // this chunk's buffers any more: the last pair's operands are already in registers).
  const int abase = xh * 8 * 512 + (hi * 64 + mh * 32 + lo) * 4;       // A: U[xn][hi][cout][j], one 16-byte read = 4 k-steps
  const int bbase = xh * 8 * 512 + hi * 64 + tr * 32 + lo;              // B: V[xn][2 j + hi][tile], four 4-byte reads
  f32x4 A[2][2];
  float B[2][2][4];
  auto load_pair = [&](const float* s_u, const float* s_v, int pp, int rb) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      A[rb][e] = *reinterpret_cast<const f32x4*>(s_u + (2 * pp + e) * 512);
#pragma unroll
      for (int j = 0; j < 4; ++j) B[rb][e][j] = s_v[(2 * pp + e) * 512 + j * 128];
    }
  };
  auto block = [&](int k, auto has_next_tag, auto first_tag) {
    constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    const int buf = k & 1;
    const float* s_u = s_u0 + buf * UV + abase;
    const float* s_v = s_v0 + buf * UV + bbase;
    const bool raw2 = HAS_NEXT && k + 2 < a.nchunks;
    static_for<0, 4>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      constexpr int rb = p & 1;
      if (p == 3 && HAS_NEXT) {
        if (!(WINO_ABLATE(a) & 16)) __syncthreads();  // next U / V / raw complete (the barrier's vmcnt(0) covers the DMAs)
        if (!(WINO_ABLATE(a) & 8)) load_pair(s_u0 + (buf ^ 1) * UV + abase, s_v0 + (buf ^ 1) * UV + bbase, 0, 0);
      }
      if (p + 1 < 4 && !(WINO_ABLATE(a) & 8)) load_pair(s_u, s_v, p + 1, rb ^ 1);
      if (HAS_NEXT && !(WINO_ABLATE(a) & 2)) {
        if (p == 0) tf_load(buf ^ 1);
      }
      if (HAS_NEXT && !(WINO_ABLATE(a) & 4)) {
        if (p < 2) {
          issue_u_piece(k + 1, buf ^ 1, 2 * p);
          issue_u_piece(k + 1, buf ^ 1, 2 * p + 1);
          if (raw2 && p < NI) issue_raw_piece(k + 2, buf, p);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int i = 2 * p + e;
          if (FIRST && j == 0)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rb][e][j], B[rb][e][j], i == 5 ? cinit5 : zero16, 0, 0, 0);
          else
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rb][e][j], B[rb][e][j], acc[i], 0, 0, 0);
        }
      if (HAS_NEXT && !(WINO_ABLATE(a) & 2)) {
        if (p == 0) tf_rows();
        if (p == 1) { tf_cols(0, buf ^ 1); tf_cols(1, buf ^ 1); }
        if (p == 2) { tf_cols(2, buf ^ 1); tf_cols(3, buf ^ 1); }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  WINO_STAMP(0);
#ifdef DVSR_CONV_TRACE
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memrealtime();
#endif
  issue_u(0, 0);
  issue_raw(0, 0);
  issue_raw(1, 1);   // (at least two chunks: conv2d_packed_prepare)
  __syncthreads();
  WINO_STAMP(1);
  transform(0, 0);
  __syncthreads();
  load_pair(s_u0 + abase, s_v0 + bbase, 0, 0);
  WINO_STAMP(2);
  block(0, std::true_type{}, std::true_type{});
  WINO_STAMP(3);
  for (int k = 1; k + 1 < a.nchunks; ++k) {
    block(k, std::true_type{}, std::false_type{});
    if (k < 30) WINO_STAMP(3 + k);
  }
  block(a.nchunks - 1, std::false_type{}, std::false_type{});
  WINO_STAMP(40);

  // ---- epilogue.  Y = A^T M A is linear in the rows of M: this wave reduces ITS two rows (xi = 2 xh, 2 xh + 1) to a partial
  // 2x2 output per (cout, tile), the two waves of a pair swap halves through LDS (the idle U / V buffers of the other
  // parity: the last block reads buffer (nchunks - 1) & 1 only) and each finishes 8 of the 16 cout registers:
  // activation / residual / accumulate / gradient mask / PixelShuffle(2) as store_mfma_tile does for the direct kernels.
  const int ttw = tr * 32 + lo;                        // this lane's tile
  const int orow = oy0 + 2 * (ttw / TC), ocol = ox0 + 2 * (ttw % TC);
  const size_t HWo = (size_t)a.Ho * a.Wo;
  const float slope = a.act == ACT_LRELU ? 0.1f : (a.act == ACT_RELU ? 0.f : 1.f);
  const float neg = a.gmask_act == ACT_LRELU ? 0.1f : (a.gmask_act == ACT_RELU ? 0.f : 1.f);
  const bool full = oy0 + Sh::OH <= a.Ho && ox0 + Sh::OW <= a.Wo && cbi * 64 + 64 <= a.Cout;
  const int ob = ((a.nchunks - 1) & 1) ^ 1;
  const int q = mh + 2 * tr;   // the pair
  float* const xch = (q < 2 ? s_u0 + ob * UV : s_v0 + ob * UV) + (q & 1) * 4096 + lane * 4;   // slot [receiving half][rr][lane]
  // full tiles: address = scalar base of (image, cout) + one per-lane byte offset
  const char* const ybase = reinterpret_cast<const char*>(a.y + ((size_t)n * a.Cout + co_block) * HWo);
  const unsigned lane_off = (unsigned)(((size_t)(4 * hi) * HWo + (size_t)orow * a.Wo + ocol) * 4);
  auto finish = [&](auto xh_) {   // (one instantiation per half: register indices stay compile-time constants)
    constexpr int XH = decltype(xh_)::value;
    // pp[P] = (y00, y01, y10, y11) of the register PAIR (2P, 2P + 1) = output channels (co, co + 1), as packed pairs
    f32x2 pp[8][4];
#pragma unroll
    for (int P = 0; P < 8; ++P) {
      f32x2 s0[4], s1[4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        const f32x2 ma = {acc[nu][2 * P], acc[nu][2 * P + 1]}, mb = {acc[4 + nu][2 * P], acc[4 + nu][2 * P + 1]};
        // XH = 0: rows 0, 1 of M: s0 = M0 + M1, s1 = M1;  XH = 1: rows 2, 3: s0 = M2, s1 = -(M2 + M3) (sign applied below)
        s0[nu] = XH == 0 ? ma + mb : ma;
        s1[nu] = XH == 0 ? mb : ma + mb;
      }
      pp[P][0] = s0[0] + s0[1] + s0[2];
      pp[P][1] = s0[1] - s0[2] - s0[3];
      if (XH == 0) {
        pp[P][2] = s1[0] + s1[1] + s1[2];
        pp[P][3] = s1[1] - s1[2] - s1[3];
      } else {
        pp[P][2] = -s1[0] - s1[1] - s1[2];
        pp[P][3] = s1[2] + s1[3] - s1[1];
      }
    }
    // swap: the pairs of the OTHER half's registers go out (two 16-byte slots per pair), this half's come in
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int P = (XH ^ 1) * 4 + q4;
      *reinterpret_cast<f32x4*>(xch + (((XH ^ 1) * 4 + q4) * 2 + 0) * 256) = f32x4{pp[P][0][0], pp[P][0][1], pp[P][1][0], pp[P][1][1]};
      *reinterpret_cast<f32x4*>(xch + (((XH ^ 1) * 4 + q4) * 2 + 1) * 256) = f32x4{pp[P][2][0], pp[P][2][1], pp[P][3][0], pp[P][3][1]};
    }
    // full tiles with a residual / accumulate / gradient mask: their loads go out before the swap's barrier
    // (ex[q4][c][i] = value added after the activation, gm = multiplier of the result)
    const bool plain = !a.res && !a.accum && !a.gmask;
    f32x2 ex[4][2][2];
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
            f32x2 e = {0.f, 0.f};
            if (a.res) e = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(a.res) + sb + lane_off);
            if (a.accum) e += *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(a.y) + sb + lane_off);
            ex[q4][c][i] = e;
          }
      }
    }
    __syncthreads();
    f32x2 o[4][4];   // own pairs, activated
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int P = XH * 4 + q4;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(xch + ((XH * 4 + q4) * 2 + 0) * 256);
      const f32x4 r1 = *reinterpret_cast<const f32x4*>(xch + ((XH * 4 + q4) * 2 + 1) * 256);
      const f32x2 in[4] = {f32x2{r0[0], r0[1]}, f32x2{r0[2], r0[3]}, f32x2{r1[0], r1[1]}, f32x2{r1[2], r1[3]}};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 v = pp[P][e] + in[e];
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
            f32x2 v = {o[q4][2 * i][c], o[q4][2 * i + 1][c]};
            if (full) {
              const size_t sb = ((size_t)(rc + c) * HWo + (size_t)i * a.Wo) * 4;   // scalar
              float* dst = reinterpret_cast<float*>(const_cast<char*>(ybase) + sb + lane_off);
              if (!plain) {
                v += ex[q4][c][i];
                if (a.gmask) {   // (data-gradient launches: the activation mask of the producer, read late -- registers)
                  const size_t sg = (((size_t)n * a.Cout + co_block + rc + c) * HWo + (size_t)i * a.Wo) * 4;
                  const f32x2 m = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(a.gmask) + sg + lane_off);
                  v = f32x2{v[0] * (m[0] > 0.f ? 1.f : neg), v[1] * (m[1] > 0.f ? 1.f : neg)};
                }
              }
              *reinterpret_cast<f32x2*>(dst) = v;
              continue;
            }
            const int co = co_block + rc + c + 4 * hi;
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
      // PixelShuffle(2): channels 4 cq .. 4 cq + 3 (registers 4 g .. 4 g + 3) are the 2x2 sub-pixels (dy, dx) of channel cq;
      // one output row of a tile is 4 consecutive floats (x = 2 ocol .. 2 ocol + 3, dx interleaved): the register pair
      // (dx = 0, 1) of y_i0 followed by the pair of y_i1 -> 16-byte stores straight from the packed pairs
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int co = co_block + 8 * (2 * XH + gg) + 4 * hi;
        const int cq = co >> 2;
        if (co >= a.Cout) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int oy = orow + i;
          if (!full && (oy >= a.Ho || ocol >= a.Wo)) continue;
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            const f32x2 e0 = o[2 * gg + dy][2 * i], e1 = o[2 * gg + dy][2 * i + 1];
            const f32x4 v = f32x4{e0[0], e0[1], e1[0], e1[1]};
            float* dst = a.y + (((size_t)n * (a.Cout >> 2) + cq) * (2 * a.Ho) + (2 * oy + dy)) * (size_t)(2 * a.Wo) + 2 * ocol;
            if (full || ocol + 1 < a.Wo) *reinterpret_cast<f32x4*>(dst) = v;
            else *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
          }
        }
      }
    }
  };
  if (xh == 0) finish(std::integral_constant<int, 0>{});
  else finish(std::integral_constant<int, 1>{});
#ifdef DVSR_CONV_TRACE
  WINO_STAMP(41);
  __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
  WINO_STAMP(42);
  if (a.trace && threadIdx.x == 0) {
    a.trace[(size_t)blockIdx.x * 64 + 61] = __builtin_amdgcn_s_memrealtime();
    a.trace[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_ID
  }
#endif
}

template <int TC>
static int launch_wino(ConvK2 k, hipStream_t st) {
  using Sh = WinoShape<TC>;
  auto kern = conv2d_wino_kernel<TC>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, Sh::LDS_BYTES);
  k.tiles_x = ceil_div(k.Wo, Sh::OW); k.tiles_y = ceil_div(k.Ho, Sh::OH); k.ntiles = k.tiles_x * k.tiles_y * k.N;
  k.ncb = ceil_div(k.Cout, 64);
  k.tiles_per_xcd = ceil_div(k.ntiles, 8);
  k.nitems = k.tiles_per_xcd * 8 * k.ncb;
  hipLaunchKernelGGL(kern, dim3(k.nitems), dim3(512), Sh::LDS_BYTES, st, k);
  return check_launch("conv2d_wino_kernel");
}

// th = 4: 4 x 64-pixel workgroup tiles (TC = 32), th = 8: 8 x 32 (TC = 16)
int conv2d_wino_launch(const ConvK2& k, int th, hipStream_t st) {
  return th == 8 ? launch_wino<16>(k, st) : launch_wino<32>(k, st);
}

}  // namespace dvsr
