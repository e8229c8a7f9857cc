// Pipelined MFMA convolution used by the EDVR engine (forward and data-gradient).
//
// Same math as conv2d.hip (implicit GEMM on v_mfma_f32_32x32x2_f32, D rows = cout / columns =
// pixels, exact fp32) with the K loop software-pipelined and the launch geometry chosen per layer:
//   * weights are PRE-PACKED once per forward by pack_weights_kernel into the exact LDS image of
//     every (64-cout block, channel chunk): [mt][tap][q][hi][lo] x float4(kk = 4q..4q+3), channel
//     c = 2kk + hi of the chunk.  Staging them is a straight 16-byte-per-lane LDS-DMA
//     (global_load_lds_dwordx4): no index math, no VGPRs, no ds_write;
//   * the input halo tile of chunk k+1 is fetched into registers and the weight DMA of chunk k+1
//     is issued BEFORE the MFMAs of chunk k; both LDS images are double buffered -> one barrier
//     per chunk, HBM/L2 latency hides under the MFMAs.  Loaded halo values are only consumed
//     (masked, transposed) when they are written to LDS one iteration later, so no s_waitcnt sits
//     behind the prefetch loads;
//   * the halo tile is stored as [q][row][hi][x] x float4(kk): every MFMA operand set (4 k-steps)
//     is ONE conflict-free ds_read_b128 per lane for A and for B;
//   * tile geometry is a template: TH x 32 pixels (TH = 8: two pixel rows per wave, TH = 4: one)
//     by 32*MT output channels (MT = 2 or 1).  The host picks (TH, MT) per launch to minimise
//     max(MFMA-pipe time of the busiest CU, latency of the serial chunk loop): 8x32x64 for the big
//     layers, 4x32x64 / 4x32x32 when the grid would not fill 256 CUs x 2-3 workgroups.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

// Debug timeline: thread 0 of every workgroup stamps s_memtime at the phase boundaries of the
// pipeline.  Compiled in only with -DDVSR_CONV_TRACE (a separate library; the product build has none).
#ifdef DVSR_CONV_TRACE
#define DVSR_ABLATE(a) ((a).ablate)
#define DVSR_STAMP(i)                                                                              \
  do {                                                                                             \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define DVSR_ABLATE(a) 0
#define DVSR_STAMP(i) \
  do {                \
  } while (0)
#endif

// BF = bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): a 16-byte LDS operand then holds 8
// consecutive channels (k = 8*hi .. 8*hi+7 of a 16-channel block) instead of the 4 of the fp32 layout
// (k = 2kk + hi), everything else -- tiles, double buffering, DMA pieces, epilogue -- is shared.
// BF = 2 ("split3", experimental): both operands are split into three bf16 pieces (hi / mid / lo) and the six
// largest partial products are issued -- as accurate as the exact-fp32 MFMA against a double reference
// (profiles/r01_bf16_split_probe.txt) for 6 x 32 instead of 8 x 64 pipe cycles per tap and 16 channels.
// LDS holds the three halo pieces only (double-buffered 65 KB at 8x32 pixels: two workgroups per CU); the
// weight pieces are read per k-step straight from the packed image in global memory (Sh::AG).
template <int KS, int S, int CC, int TH, int MT, int BF = 0>
struct Conv2Shape {
  static constexpr int TW = 32, KK = KS * KS, NT = TH / 4;  // NT pixel rows per wave
  static constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
  static constexpr int PLANE = IH * IW, E = (PLANE + 255) / 256;
  static constexpr int KQ4 = BF ? CC / 16 : CC / 8;   // 16-byte operand groups (x 2 lane halves) per chunk
  static constexpr int PIECES = BF == 2 ? 3 : 1;      // bf16 pieces per operand
  static constexpr int IN1 = KQ4 * PLANE * 8;         // one piece of the halo image = KQ4 * IH * 2 * IW float4
  static constexpr int IN_FLOATS = PIECES * IN1;
  static constexpr int HALF = KK * KQ4 * 2 * 32 * 4;  // packed floats of one 32-cout half (of one piece)
  static constexpr int W1 = MT * HALF;                // one piece of the weight image
  static constexpr bool AG = BF == 2;                 // A operands straight from global memory (L1/L2), not via LDS
  static constexpr int W_FLOATS = AG ? 0 : PIECES * W1;
  static constexpr int NPIECE = W_FLOATS / 256;       // 1-KiB DMA pieces (one wave-instruction each)
  static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF_FLOATS * sizeof(float);
};

// channels per chunk: enough k-steps per barrier (72 MFMAs per wave for 3x3, 64 for 2x2 and 1x1)
int conv2_cc(int ks, int stride) { (void)stride; return ks == 1 ? 32 : (ks == 2 ? 16 : 8); }
int conv2_pch(int ks, int stride) {  // packed floats per (64-cout block, chunk): two halves
  return 2 * ks * ks * (conv2_cc(ks, stride) / 8) * 2 * 32 * 4;
}

// ---- weight packing ---------------------------------------------------------------------------
// P[cb][k][((((mt*KK + tap)*KQ4 + q)*2 + hi)*32 + lo)*4 + j] = W(cout = cb*64 + mt*32 + lo,
//   cin = k*CC + 2*(4q + j) + hi, tap), zero outside.
// wt = 1 (data gradient): this conv's (cin, cout, tap) = original (cout, cin slice, mirrored tap).
__global__ void pack_weights_kernel(PackTable t) {
  const PackEntry& e = t.e[blockIdx.y];
  if (e.bf || e.perm >= 3) return;  // packed by pack_weights_bf16_kernel / pack_weights_wino(3)_kernel / pack_weights_dcn3_kernel
  const int kq4 = e.CC / 8;
  const size_t per_chunk = (size_t)e.pch;
  const size_t total = (size_t)e.ncb * e.nchunks * per_chunk;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int r = (int)(i % per_chunk);
    const size_t ck = i / per_chunk;
    const int k = (int)(ck % e.nchunks), cb = (int)(ck / e.nchunks);
    const int j = r & 3; r >>= 2;
    const int lo = r & 31; r >>= 5;
    const int hi = r & 1; r >>= 1;
    const int q = r % kq4; r /= kq4;
    const int tap = r % e.KK;
    const int mt = r / e.KK;
    // channel of (operand group q, lane half hi, k-step j): interleaved for the register-staged halo image,
    // 4 consecutive channels per lane half for the DMA-staged planar one (conv2d_dma_item)
    const int co = cb * 64 + mt * 32 + lo, ci = k * e.CC + (e.perm ? 8 * q + 4 * hi + j : 2 * (4 * q + j) + hi);
    float v = 0.f;
    if (co < e.Cout && ci < e.Ctot) {
      if (!e.wt) v = e.w[((size_t)co * e.Ctot + ci) * e.KK + tap];
      else v = e.w[((size_t)ci * e.w_ctot + e.w_coff + co) * e.KK + (e.KK - 1 - tap)];
    }
    e.P[i] = v;
  }
}

// bf16 image: P16[cb][k][((((mt*KK + tap)*KB + q)*2 + hi)*32 + lo)*8 + i] = bf16(W(cout = cb*64 + mt*32 + lo,
//   cin = k*CC + 16q + 8hi + i, tap)); e.pch counts fp32-sized slots, i.e. 2 bf16 each.
__global__ void pack_weights_bf16_kernel(PackTable t) {
  const PackEntry& e = t.e[blockIdx.y];
  if (!e.bf) return;
  const int kb = e.CC / 16;
  const size_t per_chunk = (size_t)e.pch * 2;
  const size_t total = (size_t)e.ncb * e.nchunks * per_chunk;
  __bf16* P16 = reinterpret_cast<__bf16*>(e.P);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int r = (int)(i % per_chunk);
    const size_t ck = i / per_chunk;
    const int k = (int)(ck % e.nchunks), cb = (int)(ck / e.nchunks);
    const int j = r & 7; r >>= 3;
    const int lo = r & 31; r >>= 5;
    const int hi = r & 1; r >>= 1;
    const int q = r % kb; r /= kb;
    const int tap = r % e.KK; r /= e.KK;
    const int mt = r & 1;
    const int piece = r >> 1;  // 0 (the only one for bf = 1) | hi / mid / lo of the 3-way split (bf = 2)
    const int co = cb * 64 + mt * 32 + lo, ci = k * e.CC + 16 * q + 8 * hi + j;
    float v = 0.f;
    if (co < e.Cout && ci < e.Ctot) {
      if (!e.wt) v = e.w[((size_t)co * e.Ctot + ci) * e.KK + tap];
      else v = e.w[((size_t)ci * e.w_ctot + e.w_coff + co) * e.KK + (e.KK - 1 - tap)];
    }
    __bf16 out = (__bf16)v;
    if (piece > 0) {
      const float r1 = v - (float)out;
      const __bf16 m = (__bf16)r1;
      out = piece == 1 ? m : (__bf16)(r1 - (float)m);
    }
    P16[i] = out;
  }
}

int pack_weights_run(const PackTable& t, hipStream_t st) {
  if (t.n <= 0) return DVSR_OK;
  bool any_f32 = false, any_bf = false, any_wino = false, any_wino3 = false, any_wino5 = false, any_dcn3 = false;
  for (int i = 0; i < t.n; ++i)
    (t.e[i].bf ? any_bf : (t.e[i].perm == 3 ? any_wino : (t.e[i].perm == 4 ? any_wino3 : (t.e[i].perm == 5 ? any_wino5 : (t.e[i].perm == 6 ? any_dcn3 : any_f32))))) = true;
  if (any_dcn3) {
    int rc = pack_weights_dcn3_run(t, st);
    if (rc) return rc;
  }
  if (any_wino) {
    int rc = pack_weights_wino_run(t, st);
    if (rc) return rc;
  }
  if (any_wino3) {
    int rc = pack_weights_wino3_run(t, st);
    if (rc) return rc;
  }
  if (any_wino5) {
    int rc = pack_weights_wino5_run(t, st);
    if (rc) return rc;
  }
  if (any_f32) {
    hipLaunchKernelGGL(pack_weights_kernel, dim3(48, t.n), dim3(256), 0, st, t);
    int rc = check_launch("pack_weights_kernel");
    if (rc) return rc;
  }
  if (any_bf) {
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(48, t.n), dim3(256), 0, st, t);
    return check_launch("pack_weights_bf16_kernel");
  }
  return DVSR_OK;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One work item = one (pixel tile, 32*MT-cout block).
template <int KS, int S, int CC, int TH, int MT, int BF>
__device__ __forceinline__ void conv2d_pipe_item(const ConvK2& a, const int id, float* const smem) {
  using Sh = Conv2Shape<KS, S, CC, TH, MT, BF>;
  constexpr int KK = Sh::KK, IH = Sh::IH, IW = Sh::IW, PLANE = Sh::PLANE, E = Sh::E, KQ4 = Sh::KQ4,
                NT = Sh::NT;
  float* const s_in0 = smem;
  float* const s_w0 = smem + Sh::IN_FLOATS;

  // XCD-aware order.  Workgroup ids are dealt round-robin to the 8 XCDs (id & 7), each with its own L2.  An XCD
  // owns a contiguous range of tiles in (frame, row, column) order, i.e. a band of tile rows, walked top-down:
  // the two halo rows a tile shares with its vertical neighbours are then L2 hits (with tiles dealt round-robin,
  // vertical neighbours sat on different XCDs and every halo row crossed the fabric twice: 87 MB fetched per
  // launch against 51 MB of input).  The ncb cout blocks of a tile follow each other on the same XCD.
  const int q_ = id >> 3;
  const int cbi = q_ % a.ncb;  // block of 32*MT output channels
  const int j_ = q_ / a.ncb;
  const int tile = (id & 7) * a.tiles_per_xcd + j_;
  if (j_ >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * Sh::TW;
  const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: DMA addressing stays scalar
  const int lo = lane & 31, hi = lane >> 5;
  const int Ctot = a.c0 + a.c1;
  const size_t HW = (size_t)a.H * a.W;
  const float* x0n = a.x0 + (size_t)n * a.x0_bs;
  const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : x0n;
  const size_t cstride0 = a.in_dil ? (size_t)a.Hs * a.Ws : HW;  // x0 channel stride

  // per-thread halo elements: byte offset inside a channel plane + validity, fixed for all chunks
  unsigned eoffb[E];
  int elds[E];
  bool evalid[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = tid + 256 * e;
    const int iy = idx / IW, ix = idx - iy * IW;
    elds[e] = (iy * 2) * IW + ix;  // float4 index of (row iy, hi 0, x ix) inside one q-slab
    const int gy = iy0 + iy, gx = ix0 + ix;
    bool ok = idx < PLANE && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    int off = gy * a.W + gx;
    if (a.in_ps) off = (2 * gy) * (2 * a.W) + 2 * gx;
    if (a.in_dil) {
      ok = ok && !((gy | gx) & 1) && (gy >> 1) < a.Hs && (gx >> 1) < a.Ws;
      off = (gy >> 1) * a.Ws + (gx >> 1);
    }
    eoffb[e] = ok ? (unsigned)off * 4u : 0u;
    evalid[e] = ok;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float rin[CC][E];
  // packed weights of this workgroup's cout block: 64-cout block (cbi*MT)/2, starting half (cbi*MT)%2
  // global image: [64-cout block][chunk][piece][half][HALF]
  constexpr int PIECES = Sh::PIECES;
  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + ((size_t)((cbi * MT) >> 1) * a.nchunks * PIECES * 2 + ((cbi * MT) & 1)) * Sh::HALF;

  // Halo loads of chunk k into registers (raw; masked when they are written to LDS).  All address
  // math is wave-uniform scalar work: one base pointer per chunk (a chunk never straddles the two
  // inputs: the host checks c0 % CC == 0), then a pointer increment per channel -- the channel
  // stride, or the (+1, +2W-1, +1, +4HW-2W-1) walk of a pixel-shuffled input.  The lane part is the
  // fixed 32-bit byte offset eoffb: global_load_dword v, v_off, s[base:base+1], no VALU at all.
  auto issue_halo = [&](int k) {
    const int cbase = k * CC;
    const bool second = cbase >= a.c0;  // only possible when c1 > 0
    const float* b = second ? x1n : x0n;
    const int ci = second ? cbase - a.c0 : cbase;
    const size_t cs = second ? HW : cstride0;
    const char* p = reinterpret_cast<const char*>(b + (size_t)ci * cs);
    const size_t inc0 = a.in_ps ? 4 : cs * 4;                                    // c even -> c + 1
    const size_t inc1 = a.in_ps ? ((size_t)2 * a.W - 1) * 4 : cs * 4;            // c % 4 == 1
    const size_t inc3 = a.in_ps ? ((size_t)4 * HW - 2 * a.W - 1) * 4 : cs * 4;   // c % 4 == 3
#pragma unroll
    for (int c = 0; c < CC; ++c) {
#pragma unroll
      for (int e = 0; e < E; ++e) rin[c][e] = *reinterpret_cast<const float*>(p + eoffb[e]);
      if (c + 1 < CC) {
        const size_t inc = (c & 1) == 0 ? inc0 : ((c & 3) == 1 ? inc1 : inc3);
        p += (cbase + c + 1 < Ctot) ? inc : 0;  // channels past the end re-read the last one (masked later)
      }
    }
  };
  // Weights of chunk k: LDS-DMA, 16 B per lane, destination = wave-uniform base + lane*16.  Every wave
  // issues the same number of DMAs (the last piece is re-sent when NPIECE % 4 != 0), so the loop body
  // stays one basic block and the compiler's vmcnt bookkeeping stays exact.
  auto issue_dma = [&](int k, int buf) {
    if constexpr (Sh::AG) return;
    const float* wsrc = wp_cb + (size_t)k * (PIECES * 2 * Sh::HALF);
    float* wdst = s_w0 + buf * Sh::BUF_FLOATS;
    constexpr int NP1 = Sh::W1 / 256;  // 1-KiB DMA pieces of one operand piece
#pragma unroll
    for (int j = 0; j < (Sh::NPIECE + 3) / 4; ++j) {
      int piece = j * 4 + wave;
      piece = piece < Sh::NPIECE ? piece : Sh::NPIECE - 1;
      const int op = piece / (NP1 > 0 ? NP1 : 1), within = piece - op * NP1;  // operand piece (hi / mid / lo), 1-KiB block inside it
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(wsrc + op * (2 * Sh::HALF) + within * 256 + lane * 4),
          (__attribute__((address_space(3))) void*)(wdst + op * Sh::W1 + within * 256), 16, 0, 0);
    }
  };
  // halo registers of chunk k -> LDS buffer, transposed to [q][row][hi][x] x float4(kk): one
  // ds_write_b128 per (q, hi)
  auto write_halo = [&](int k, int buf) {
    float* s_in = s_in0 + buf * Sh::BUF_FLOATS;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (tid + 256 * e < PLANE) {
#pragma unroll
        for (int q = 0; q < KQ4; ++q)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const bool ev = evalid[e];
            f32x4 v;
            if (BF == 2) {  // 8 consecutive channels, each split into three bf16 pieces: x = hi + mid + lo (+ 2^-24)
              const int cb0 = k * CC + 16 * q + 8 * h2;
              bf16x8 p0, p1, p2;
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float x = (ev && cb0 + i < Ctot) ? rin[16 * q + 8 * h2 + i][e] : 0.f;
                const __bf16 h = (__bf16)x;
                const float r1 = x - (float)h;
                const __bf16 m = (__bf16)r1;
                p0[i] = h; p1[i] = m; p2[i] = (__bf16)(r1 - (float)m);
              }
              float* dst = s_in + ((size_t)(q * IH * 2 * IW) + elds[e] + h2 * IW) * 4;
              *reinterpret_cast<f32x4*>(dst + Sh::IN1) = __builtin_bit_cast(f32x4, p1);
              *reinterpret_cast<f32x4*>(dst + 2 * Sh::IN1) = __builtin_bit_cast(f32x4, p2);
              v = __builtin_bit_cast(f32x4, p0);
            } else if (BF) {  // 8 consecutive channels, rounded to bf16 (RNE)
              const int cb0 = k * CC + 16 * q + 8 * h2;
              bf16x8 hv;
#pragma unroll
              for (int i = 0; i < 8; ++i) hv[i] = (__bf16)((ev && cb0 + i < Ctot) ? rin[16 * q + 8 * h2 + i][e] : 0.f);
              v = __builtin_bit_cast(f32x4, hv);
            } else {
              const int cb0 = k * CC + 8 * q + h2;
              v = f32x4{(ev && cb0 < Ctot) ? rin[8 * q + h2][e] : 0.f,
                        (ev && cb0 + 2 < Ctot) ? rin[8 * q + 2 + h2][e] : 0.f,
                        (ev && cb0 + 4 < Ctot) ? rin[8 * q + 4 + h2][e] : 0.f,
                        (ev && cb0 + 6 < Ctot) ? rin[8 * q + 6 + h2][e] : 0.f};
            }
            *reinterpret_cast<f32x4*>(s_in + ((size_t)(q * IH * 2 * IW) + elds[e] + h2 * IW) * 4) = v;
          }
      }
    }
  };

  // One chunk.  The non-MFMA work of the pipeline lives INSIDE the wave's MFMA stream: a wave that
  // streams MFMAs keeps the SIMD's vector issue port, so work parked in another wave (or in a
  // separate phase of this one) is not overlapped by the hardware -- measured with
  // tools/conv_trace.py, a separate prefetch phase cost as much as the MFMA block itself.
  //   top of the block : weight DMA + halo loads of chunk k+1 are issued (scalar address math only)
  //   after 3/4 of it  : the halo registers (long since arrived) are masked and written to LDS
  //   end              : one barrier
  constexpr int NSTEP = KK * KQ4;
  constexpr int SPLIT = (3 * NSTEP) / 4;
  // (register ring of the global-A path: three sets where the step count divides by three -- 3x3: nine steps --, two
  // otherwise -- 2x2: four steps --; the prefetch distance is one step either way, the slot index stays static across chunks)
  constexpr int RING = NSTEP % 3 == 0 ? 3 : 2;
  static_assert(!Sh::AG || (NSTEP % RING == 0 && KQ4 == 1), "the register ring of the global-A path needs NSTEP % RING == 0");
  // BF == 2: the weight operands never touch LDS.  Every lane reads its 16 B of each (piece, 32-cout half) of
  // one tap straight from the packed image (1 KiB per wave instruction, the four waves of a workgroup read the
  // same lines: L1 hits), two steps ahead of their use.  LDS then holds only the three halo pieces, which
  // leaves room for two workgroups per CU, and its read port only serves the B operands.
  constexpr int AD = 1;  // prefetch distance in steps
  f32x4 Ag[Sh::AG ? RING : 1][MT][PIECES];
  auto load_a = [&](int k, int tap, int slot) {
    const float* b = wp_cb + (size_t)k * (PIECES * 2 * Sh::HALF) + tap * 256 + lane * 4;
#pragma unroll
    for (int op = 0; op < PIECES; ++op)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        Ag[slot][mt][op] = *reinterpret_cast<const f32x4*>(b + (op * 2 + mt) * Sh::HALF);
  };
  auto block = [&](int k, auto has_next_tag) {
    constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
    const int buf = k & 1;
    const float* s_in = s_in0 + buf * Sh::BUF_FLOATS;
    const float* s_w = s_w0 + buf * Sh::BUF_FLOATS;
    if (k < 8) DVSR_STAMP(2 + 4 * k);
    if (HAS_NEXT && !Sh::AG) {
      issue_dma(k + 1, buf ^ 1);
      issue_halo(k + 1);
      // keep the loads up here: the scheduler would sink them to their use (a sched_group_barrier pattern
      // "one prefetch load behind each of the first MFMAs" was tried: it also let them sink)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (k < 8) DVSR_STAMP(3 + 4 * k);
    // MFMA over (tap, q): operands of step i+1 are read before the MFMAs of step i
    f32x4 A[2][Sh::AG ? 1 : MT][PIECES], Bv[Sh::AG ? 1 : 2][NT][PIECES];
    auto load_ops = [&](int step, int rb) {
      const int tap = step / KQ4, q = step - tap * KQ4;
      const int ty = tap / KS, tx = tap - ty * KS;
#pragma unroll
      for (int op = 0; op < PIECES; ++op) {
        if constexpr (!Sh::AG) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            A[rb][mt][op] = *reinterpret_cast<const f32x4*>(
                s_w + op * Sh::W1 + ((size_t)((((mt * KK + tap) * KQ4 + q) * 2 + hi) * 32 + lo)) * 4);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          Bv[rb][nt][op] = *reinterpret_cast<const f32x4*>(
              s_in + op * Sh::IN1 + ((size_t)(((q * IH + (NT * wave + nt) * S + ty) * 2 + hi) * IW + lo * S + tx)) * 4);
      }
    };
    auto load_b = [&](int step, int op) {  // one piece of the B operands of a step (BF == 2)
      const int ty = step / KS, tx = step - ty * KS;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        Bv[0][nt][op] = *reinterpret_cast<const f32x4*>(
            s_in + op * Sh::IN1 + ((size_t)((((NT * wave + nt) * S + ty) * 2 + hi) * IW + lo * S + tx)) * 4);
    };
    if constexpr (Sh::AG) { load_b(0, 2); load_b(0, 1); load_b(0, 0); }
    else load_ops(0, 0);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int rb = step & 1;
      if constexpr (Sh::AG) {
        // weights of step + 2 (ring of three register sets; NSTEP % 3 == 0 keeps the slot index static across
        // chunks).  The halo loads of the next chunk go out right behind the first one: vmcnt retires in
        // order, so the weights of steps 1 and 2 (issued earlier) never wait for them, and by step 3 the halo
        // has had three steps of MFMAs to arrive.
        if (step + AD < NSTEP) load_a(k, step + AD, (step + AD) % RING);
        else if (HAS_NEXT) load_a(k + 1, step + AD - NSTEP, (step + AD) % RING);
        if (HAS_NEXT && step == 0) issue_halo(k + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!Sh::AG && step + 1 < NSTEP) load_ops(step + 1, rb ^ 1);
      if (HAS_NEXT && step == SPLIT) {
        if (k < 8) DVSR_STAMP(4 + 4 * k);
        __builtin_amdgcn_sched_barrier(0);
        write_halo(k + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (k < 8) DVSR_STAMP(5 + 4 * k);
      }
      if constexpr (BF == 2) {
        // Six partial products per k-step (the three below 2^-24 are dropped).  The B pieces live in ONE
        // register set: the order retires the lo piece after the first product, the mid piece after the third,
        // and each is re-read from LDS for the next step right behind its last use -- the 4..12 MFMAs that
        // follow cover the LDS latency.  (Two sets plus the weight ring spilled at two pixel rows per wave.)
        auto mm = [&](int pa, int pb) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, Ag[step % RING][mt][pa]), __builtin_bit_cast(bf16x8, Bv[0][nt][pb]),
                  acc[mt][nt], 0, 0, 0);
        };
        mm(0, 2);
        if (step + 1 < NSTEP) { load_b(step + 1, 2); __builtin_amdgcn_sched_barrier(0); }
        mm(1, 1);
        mm(0, 1);
        if (step + 1 < NSTEP) { load_b(step + 1, 1); __builtin_amdgcn_sched_barrier(0); }
        mm(2, 0);
        mm(1, 0);
        mm(0, 0);
        if (step + 1 < NSTEP) { load_b(step + 1, 0); __builtin_amdgcn_sched_barrier(0); }
      } else if (BF) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[rb][mt][0]),
                                                                  __builtin_bit_cast(bf16x8, Bv[rb][nt][0]), acc[mt][nt],
                                                                  0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rb][mt][0][j], Bv[rb][nt][0][j], acc[mt][nt], 0, 0, 0);
      }
    }
    if (HAS_NEXT) __syncthreads();  // next buffers complete (the barrier's vmcnt(0) covers the DMA)
  };

  DVSR_STAMP(0);
#ifdef DVSR_CONV_TRACE
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + 60] = __builtin_amdgcn_s_memrealtime();
#endif
  issue_dma(0, 0);
  if constexpr (Sh::AG) {
#pragma unroll
    for (int i = 0; i < AD; ++i) load_a(0, i, i);
  }
  issue_halo(0);
  write_halo(0, 0);
  DVSR_STAMP(1);
  __syncthreads();
  for (int k = 0; k + 1 < a.nchunks; ++k) block(k, std::true_type{});
  block(a.nchunks - 1, std::false_type{});

  // ---- epilogue: bias, activation, residual / accumulate, (pixel-shuffled) store
  DVSR_STAMP(40);
  const TileOut t{a.y, wset_ptr(a.bias, a.b_gs, n, a.wdiv), a.res, a.act, a.ps, a.accum, a.Cout, a.Ho, a.Wo, a.gmask, a.gmask_act};
  store_mfma_tile<MT, NT>(acc, t, n, cbi * MT * 32, oy0, TH, ox0, oy0 + NT * wave, lo, hi);
#ifdef DVSR_CONV_TRACE
  DVSR_STAMP(41);
  __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
  DVSR_STAMP(42);
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + 61] = __builtin_amdgcn_s_memrealtime();
  if (a.trace && threadIdx.x == 0)
    a.trace[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_ID
#endif
}

// tools/conv_trace.py reports the per-CU occupancy of a launch: 2.6-2.7 of 3 workgroup slots on average, ~13 k
// cycles of slot turnover.
template <int KS, int S, int CC, int TH, int MT, int BF = 0>
__global__ __launch_bounds__(256, (BF == 2 && TH == 4) ? 3 : 2) void conv2d_pipe_kernel(ConvK2 a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv2d_pipe_item<KS, S, CC, TH, MT, BF>(a, blockIdx.x, smem);
}

template <int KS, int S, int CC, int TH, int MT, int BF = 0>
static int launch_conv2(ConvK2 k, hipStream_t st) {
  using Sh = Conv2Shape<KS, S, CC, TH, MT, BF>;
  auto kern = conv2d_pipe_kernel<KS, S, CC, TH, MT, BF>;
  static PerDeviceOnce attr_once;
  size_t lds = Sh::LDS_BYTES;
#ifdef DVSR_CONV_TRACE
  // debug: DVSR_CONV_LDS=<bytes> inflates the LDS request to force fewer workgroups per CU
  if (const char* e = getenv("DVSR_CONV_LDS")) lds = std::max(lds, (size_t)atol(e));
#endif
  set_dyn_lds_once(attr_once, (const void*)kern, lds);
  k.tiles_x = ceil_div(k.Wo, 32); k.tiles_y = ceil_div(k.Ho, TH); k.ntiles = k.tiles_x * k.tiles_y * k.N;
  k.ncb = ceil_div(k.Cout, 32 * MT);
  k.tiles_per_xcd = ceil_div(k.ntiles, 8);
  k.nitems = k.tiles_per_xcd * 8 * k.ncb;
  // One workgroup per item.  A persistent launch (256 CUs x resident workgroups walking the items with a grid
  // stride) was measured 2-3 % slower on every big layer, and the grid-stride loop alone costs 19 VGPRs, i.e.
  // the third workgroup per CU (152 -> 171); a chunk stream across a workgroup's items with cross-item
  // prefetch and deferred stores was no faster either.
  const int grid = k.nitems;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, k);
  return check_launch("conv2d_pipe_kernel");
}


// -------------------------------------------------------------------------------------------------
// Halo-by-DMA variant of the pipelined kernel (fp32, 3x3 / stride 1 / pad 1, plain inputs, W % 4 == 0,
// channel counts % 8 == 0): the big layers of the 180x320 forward.
//
// conv2d_pipe_item spends ~75 cycles of its MFMA stream on every vector-memory instruction it issues (ablations in
// profiles/r01_mfma_ceiling.txt: the 8 halo loads per wave and chunk cost 12 % of the kernel, the weight DMA
// 7 %), then masks and transposes the halo through VGPRs.  Here the halo goes global -> LDS by DMA as well, in
// 16-byte pieces: the tile's input window is widened to the aligned columns [ox0 - 4, ox0 + 36), ten float4
// groups per (channel, row), so one global_load_lds_dwordx4 moves 64 groups = 1 KiB and a chunk
// (8 channels x 6 rows x 10 groups = 480 groups) is TWO instructions per wave instead of eight, with no staging
// registers, no masking and no ds_write.  The LDS image is planar, [channel][row][40 columns]; a B operand is four
// ds_read_b32 (channels 4*hi + j, the order pack_weights_kernel uses with perm = 1), conflict-free: the 32 lanes
// of a half read 32 consecutive dwords.
// Groups outside the image are never written by the DMA (their lanes are masked off); the prologue zeroes them
// once in both buffers -- validity depends on the position only, not on the chunk.
// -------------------------------------------------------------------------------------------------
template <int TH, int MT>
struct DmaShape {
  static constexpr int KK = 9, CC = 8, NT = TH / 4, IH = TH + 2, RP = 40, GR = RP / 4;
  static constexpr int NG = CC * IH * GR;                 // 16-byte groups of one chunk's halo image
  static constexpr int NI = (NG + 255) / 256;             // halo DMA instructions per wave and chunk
  static constexpr int IN_FLOATS = NG * 4;
  static constexpr int HALF = KK * 2 * 32 * 4;            // packed floats of one 32-cout half
  static constexpr int W_FLOATS = MT * HALF;
  static constexpr int NPIECE = W_FLOATS / 256;
  static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF_FLOATS * sizeof(float);
};

template <int TH, int MT>
__device__ __forceinline__ void conv2d_dma_item(const ConvK2& a, const int id, float* const smem) {
  using Sh = DmaShape<TH, MT>;
  constexpr int KK = Sh::KK, CC = Sh::CC, IH = Sh::IH, NT = Sh::NT, RP = Sh::RP, GR = Sh::GR, NI = Sh::NI;
  float* const s_in0 = smem;
  float* const s_w0 = smem + Sh::IN_FLOATS;

  const int q_ = id >> 3;  // XCD-aware order, as conv2d_pipe_item
  const int cbi = q_ % a.ncb;
  const int j_ = q_ / a.ncb;
  const int tile = (id & 7) * a.tiles_per_xcd + j_;
  if (j_ >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * 32;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const size_t HW = (size_t)a.H * a.W;
  const float* x0n = a.x0 + (size_t)n * a.x0_bs;
  const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : x0n;

  // the groups this lane moves: group L = 64 * (wave + 4 jj) + lane = (channel, row, column group)
  unsigned hoff[NI];
  bool hval[NI];
#pragma unroll
  for (int jj = 0; jj < NI; ++jj) {
    const int L = 64 * (wave + 4 * jj) + lane;
    const int c = L / (IH * GR), r = L - c * (IH * GR);
    const int iy = r / GR, g = r - iy * GR;
    const int gy = oy0 - 1 + iy, gx = ox0 - 4 + 4 * g;
    const bool ok = L < Sh::NG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    hoff[jj] = ok ? (unsigned)(((size_t)c * HW + (size_t)gy * a.W + gx) * 4) : 0u;
    hval[jj] = ok;
    if (L < Sh::NG && !ok) {
      *reinterpret_cast<f32x4*>(s_in0 + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(s_in0 + Sh::BUF_FLOATS + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + ((size_t)((cbi * MT) >> 1) * a.nchunks * 2 + ((cbi * MT) & 1)) * Sh::HALF;

  auto issue_halo = [&](int k, int buf) {
    const int cbase = k * CC;
    const bool second = cbase >= a.c0;  // only possible when c1 > 0; a chunk never straddles the two inputs
    const float* b = second ? x1n : x0n;
    const int ci = second ? cbase - a.c0 : cbase;
    const char* p = reinterpret_cast<const char*>(b + (size_t)ci * HW);
    float* dst = s_in0 + buf * Sh::BUF_FLOATS;
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
      if (hval[jj])
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + hoff[jj]),
                                         (__attribute__((address_space(3))) void*)(dst + 256 * (wave + 4 * jj)), 16, 0, 0);
    }
  };
  auto issue_w = [&](int k, int buf) {
    const float* wsrc = wp_cb + (size_t)k * (2 * Sh::HALF);
    float* wdst = s_w0 + buf * Sh::BUF_FLOATS;
#pragma unroll
    for (int j = 0; j < (Sh::NPIECE + 3) / 4; ++j) {
      int piece = j * 4 + wave;
      piece = piece < Sh::NPIECE ? piece : Sh::NPIECE - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + piece * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(wdst + piece * 256), 16, 0, 0);
    }
  };

  // B operand of (row nt of this wave, tap (ty, tx), k-step j): channel 4 hi + j, row NT wave + nt + ty, column lo + tx + 3
  const int bbase = (4 * hi * IH + NT * wave) * RP + lo + 3;
  auto block = [&](int k, auto has_next_tag) {
    constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
    const int buf = k & 1;
    const float* s_in = s_in0 + buf * Sh::BUF_FLOATS + bbase;
    const float* s_w = s_w0 + buf * Sh::BUF_FLOATS;
    if (HAS_NEXT) {
      if (!(DVSR_ABLATE(a) & 4)) issue_w(k + 1, buf ^ 1);
      if (!(DVSR_ABLATE(a) & 2)) issue_halo(k + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 A[2][MT];
    float Bv[2][NT][4];
    auto load_ops = [&](int tap, int rb) {
      const int ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        A[rb][mt] = *reinterpret_cast<const f32x4*>(s_w + ((size_t)(((mt * KK + tap) * 2 + hi) * 32 + lo)) * 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) Bv[rb][nt][j] = s_in[(j * IH + nt + ty) * RP + tx];
    };
    load_ops(0, 0);
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      const int rb = tap & 1;
      if (tap + 1 < KK) load_ops(tap + 1, rb ^ 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rb][mt][j], Bv[rb][nt][j], acc[mt][nt], 0, 0, 0);
    }
    if (HAS_NEXT && !(DVSR_ABLATE(a) & 8)) __syncthreads();  // next buffers complete (the barrier's vmcnt(0) covers both DMAs)
  };

  issue_w(0, 0);
  issue_halo(0, 0);
  __syncthreads();
  for (int k = 0; k + 1 < a.nchunks; ++k) block(k, std::true_type{});
  block(a.nchunks - 1, std::false_type{});

  if ((DVSR_ABLATE(a) & 1) && acc[0][0][0] != 12345.f) return;
  const TileOut t{a.y, wset_ptr(a.bias, a.b_gs, n, a.wdiv), a.res, a.act, a.ps, a.accum, a.Cout, a.Ho, a.Wo, a.gmask, a.gmask_act};
  store_mfma_tile<MT, NT>(acc, t, n, cbi * MT * 32, oy0, TH, ox0, oy0 + NT * wave, lo, hi);
}

template <int TH, int MT>
__global__ __launch_bounds__(256, 2) void conv2d_dma_kernel(ConvK2 a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv2d_dma_item<TH, MT>(a, blockIdx.x, smem);
}

template <int TH, int MT>
static int launch_dma(ConvK2 k, hipStream_t st) {
  using Sh = DmaShape<TH, MT>;
  auto kern = conv2d_dma_kernel<TH, MT>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, Sh::LDS_BYTES);
  k.tiles_x = ceil_div(k.Wo, 32); k.tiles_y = ceil_div(k.Ho, TH); k.ntiles = k.tiles_x * k.tiles_y * k.N;
  k.ncb = ceil_div(k.Cout, 32 * MT);
  k.tiles_per_xcd = ceil_div(k.ntiles, 8);
  k.nitems = k.tiles_per_xcd * 8 * k.ncb;
  hipLaunchKernelGGL(kern, dim3(k.nitems), dim3(256), Sh::LDS_BYTES, st, k);
  return check_launch("conv2d_dma_kernel");
}


// -------------------------------------------------------------------------------------------------
// Large kernels (7x7 of SpyNet, 9x9 of the TOFlow head; stride 1, pad KS/2, W % 4 == 0, 16-byte aligned plain input):
// the DMA-halo kernel with the K loop split by KERNEL ROW.  A 64-cout x 8-channel x 49-tap weight image is 100 KB --
// it cannot be double buffered -- so a step of the loop is (8-channel chunk, kernel row ky): KS taps, 7-9 KiB of
// weights per 32-cout half by LDS-DMA into one of two buffers, while the chunk's halo tile ((TH + KS - 1) rows x the
// aligned columns [ox0 - 4, ox0 + 36), which cover a pad of up to 4) stays in LDS for its KS steps and the next chunk's
// arrives in the other halo buffer.  One barrier per step, KS x 4 x MT x NT MFMAs between barriers.  The packed image is
// the ordinary one (KK = KS * KS taps, `perm` channel order): the taps of a kernel row are contiguous in it.
// Channel counts that are not a multiple of 8 (the 21-channel 9x9 conv) multiply stale-but-finite LDS contents by the
// pack's zero weights; both halo buffers are cleared once for that.
// (TOFlow spends 96 % of its forward in these convolutions: 17.7 ms on the single-buffered conv2d_mfma_kernel.)
// -------------------------------------------------------------------------------------------------
template <int KS, int TH, int MT>
struct RowShape {
  static constexpr int CC = 8, NT = TH / 4, PADK = KS / 2, IH = TH + KS - 1, RP = 40, GR = RP / 4;
  static_assert(PADK <= 4, "the 40-column window covers a pad of at most 4");
  static constexpr int NG = CC * IH * GR, NI = (NG + 255) / 256;
  static constexpr int IN_FLOATS = NG * 4;
  static constexpr int WROW = KS * 2 * 32 * 4;          // one kernel row of one 32-cout half: [kx][hi][lo] x float4
  static constexpr int W_FLOATS = MT * WROW;
  static constexpr int NPIECE = W_FLOATS / 256;         // = MT * KS
  static constexpr int HALF = KS * KS * 2 * 32 * 4;     // packed floats of one 32-cout half of a chunk
  static constexpr size_t LDS_BYTES = (size_t)2 * (IN_FLOATS + W_FLOATS) * sizeof(float);
};

template <int KS, int TH, int MT>
__global__ __launch_bounds__(256, 2) void conv2d_dmarow_kernel(ConvK2 a) {
  using Sh = RowShape<KS, TH, MT>;
  constexpr int CC = Sh::CC, IH = Sh::IH, NT = Sh::NT, RP = Sh::RP, GR = Sh::GR, NI = Sh::NI;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const s_in0 = smem;                       // two halo buffers
  float* const s_w0 = smem + 2 * Sh::IN_FLOATS;    // two weight-row buffers

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
  const int oy0 = ty_ * TH, ox0 = tx_ * 32;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int Ctot = a.c0 + a.c1;
  const size_t HW = (size_t)a.H * a.W;
  const float* x0n = a.x0 + (size_t)n * a.x0_bs;
  const float* x1n = a.c1 ? a.x1 + (size_t)(n / a.x1_bdiv) * a.x1_bs : x0n;

  if (Ctot % CC != 0) {  // a partial last chunk reads LDS the DMA never wrote: make it finite (x zero weights)
    for (int i = tid; i < 2 * Sh::IN_FLOATS / 4; i += 256) reinterpret_cast<f32x4*>(s_in0)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
  }
  unsigned hoff[NI];
  bool hval[NI];
  int hch[NI];
#pragma unroll
  for (int jj = 0; jj < NI; ++jj) {
    const int L = 64 * (wave + 4 * jj) + lane;
    const int c = L / (IH * GR), r = L - c * (IH * GR);
    const int iy = r / GR, g = r - iy * GR;
    const int gy = oy0 - Sh::PADK + iy, gx = ox0 - 4 + 4 * g;
    const bool ok = L < Sh::NG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    hoff[jj] = ok ? (unsigned)(((size_t)c * HW + (size_t)gy * a.W + gx) * 4) : 0u;
    hval[jj] = ok;
    hch[jj] = c;
    if (L < Sh::NG && !ok) {
      *reinterpret_cast<f32x4*>(s_in0 + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(s_in0 + Sh::IN_FLOATS + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + ((size_t)((cbi * MT) >> 1) * a.nchunks * 2 + ((cbi * MT) & 1)) * Sh::HALF;

  auto issue_halo = [&](int k, int buf) {
    const int cbase = k * CC;
    const bool second = cbase >= a.c0;  // a chunk never straddles the two inputs (host: c0 % 8 == 0 with two inputs)
    const float* b = second ? x1n : x0n;
    const int ci = second ? cbase - a.c0 : cbase;
    const int nci = (second ? a.c1 : a.c0) - ci;  // channels of this input left from the chunk's first one
    const char* p = reinterpret_cast<const char*>(b + (size_t)ci * HW);
    float* dst = s_in0 + buf * Sh::IN_FLOATS;
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
      if (hval[jj] && hch[jj] < nci)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + hoff[jj]),
                                         (__attribute__((address_space(3))) void*)(dst + 256 * (wave + 4 * jj)), 16, 0, 0);
    }
  };
  auto issue_w = [&](int k, int ky, int buf) {
    const float* wsrc = wp_cb + (size_t)k * (2 * Sh::HALF) + ky * (KS * 256);
    float* wdst = s_w0 + buf * Sh::W_FLOATS;
#pragma unroll
    for (int j = 0; j < (Sh::NPIECE + 3) / 4; ++j) {
      int piece = j * 4 + wave;
      piece = piece < Sh::NPIECE ? piece : Sh::NPIECE - 1;
      const int mt = piece / KS, kx = piece - mt * KS;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(wsrc + (size_t)mt * Sh::HALF + kx * 256 + lane * 4),
          (__attribute__((address_space(3))) void*)(wdst + piece * 256), 16, 0, 0);
    }
  };

  const int nsteps = a.nchunks * KS;
  issue_w(0, 0, 0);
  issue_halo(0, 0);
  __syncthreads();
  // B operand of (row nt of this wave, kernel row ky, tap kx, k-step j): channel 4 hi + j, row NT wave + nt + ky,
  // column lo + kx + 4 - pad of the window
  const int bbase = (4 * hi * IH + NT * wave) * RP + lo + 4 - Sh::PADK;
  int k = 0, ky = 0;
  for (int s = 0; s < nsteps; ++s) {
    const bool has_next = s + 1 < nsteps;
    if (has_next) {
      const int ky1 = ky + 1 == KS ? 0 : ky + 1;
      issue_w(ky1 == 0 ? k + 1 : k, ky1, (s + 1) & 1);
      if (ky == 0 && k + 1 < a.nchunks) issue_halo(k + 1, (k + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float* s_in = s_in0 + (k & 1) * Sh::IN_FLOATS + bbase + ky * RP;
    const float* s_w = s_w0 + (s & 1) * Sh::W_FLOATS;
    f32x4 A[2][MT];
    float Bv[2][NT][4];
    auto load_ops = [&](int kx, int rb) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        A[rb][mt] = *reinterpret_cast<const f32x4*>(s_w + ((size_t)(((mt * KS + kx) * 2 + hi) * 32 + lo)) * 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) Bv[rb][nt][j] = s_in[(j * IH + nt) * RP + kx];
    };
    load_ops(0, 0);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const int rb = kx & 1;
      if (kx + 1 < KS) load_ops(kx + 1, rb ^ 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rb][mt][j], Bv[rb][nt][j], acc[mt][nt], 0, 0, 0);
    }
    if (has_next) __syncthreads();  // next weight row (and, at a chunk boundary, the next halo tile) complete
    if (++ky == KS) { ky = 0; ++k; }
  }

  const TileOut t{a.y, wset_ptr(a.bias, a.b_gs, n, a.wdiv), a.res, a.act, a.ps, a.accum, a.Cout, a.Ho, a.Wo, a.gmask, a.gmask_act};
  store_mfma_tile<MT, NT>(acc, t, n, cbi * MT * 32, oy0, TH, ox0, oy0 + NT * wave, lo, hi);
}

template <int KS, int TH, int MT>
static int launch_dmarow(ConvK2 k, hipStream_t st) {
  using Sh = RowShape<KS, TH, MT>;
  auto kern = conv2d_dmarow_kernel<KS, TH, MT>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, Sh::LDS_BYTES);
  k.tiles_x = ceil_div(k.Wo, 32); k.tiles_y = ceil_div(k.Ho, TH); k.ntiles = k.tiles_x * k.tiles_y * k.N;
  k.ncb = ceil_div(k.Cout, 32 * MT);
  k.tiles_per_xcd = ceil_div(k.ntiles, 8);
  k.nitems = k.tiles_per_xcd * 8 * k.ncb;
  hipLaunchKernelGGL(kern, dim3(k.nitems), dim3(256), Sh::LDS_BYTES, st, k);
  return check_launch("conv2d_dmarow_kernel");
}

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv2d_ksplit_kernel(ConvK2 a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv2d_ksplit_item<MT, NT>(a, blockIdx.x, smem);
}

template <int MT, int NT>
static int launch_ksplit(ConvK2 k, hipStream_t st) {
  using Sh = KsShape<MT, NT>;
  auto kern = conv2d_ksplit_kernel<MT, NT>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, Sh::LDS_BYTES);
  hipLaunchKernelGGL(kern, dim3(k.nitems), dim3(256), Sh::LDS_BYTES, st, k);  // tiles: conv2d_packed_prepare
  return check_launch("conv2d_ksplit_kernel");
}

// Launch geometry.  Calibrated on MI355X per-layer timings of all five candidate geometries
// (profiles/r01_conv_geometry_sweep.txt): the 4x32-pixel tile beats 8x32 on every EDVR layer, 16-channel
// chunks never beat 8, and between 64 (MT=2) and 32 (MT=1) output channels per workgroup the winner is
// the one with the smaller MFMA time on the busiest CU,
//     ceil(workgroups / 256 CUs) x MFMA cycles of one workgroup,
// i.e. pure tile quantisation (e.g. Cout = 216 wastes 18 % of a 64-wide block but 4 % of 32-wide ones;
// 575 tiles are 3 rounds of 64-wide but 5 half-size rounds of 32-wide blocks).  Ties go to MT=2 (half
// the workgroups, half the weight traffic).
// cycles of a conv2d_wino5_kernel workgroup (32 tiles of 4x4 outputs x 64 couts): per 8-channel chunk and fixed (prologue +
// epilogue), from the op-level times of the 8- and 16-chunk layers at 5x180x320 (profiles/r06_wino5_layers.txt)
static constexpr double W5_CHUNK_CYC = 4300.0, W5_FIXED_CYC = 17500.0;
static double conv2_pipe_cost(int TH, int MT, int KK, int CC, int N, int Ho, int Wo, int Cout) {
  const double wgs = (double)ceil_div(Wo, 32) * ceil_div(Ho, TH) * N * ceil_div(Cout, 32 * MT);
  return ceil(wgs / 256.0) * 64.0 * KK * (CC / 2) * (TH / 4) * MT;
}

ConvGeo conv2_choose(int ks, int stride, int N, int Ho, int Wo, int Cout, int Ctot, int allow_ksplit) {
  if (ks == 7 || ks == 9) {  // row-split DMA kernel (conv2d_dmarow_kernel) or nothing: dma = 2 when it applies
    ConvGeo g{8, 4, 2};
    const double c42 = conv2_pipe_cost(4, 2, ks * ks, 8, N, Ho, Wo, Cout);
    const double c41 = conv2_pipe_cost(4, 1, ks * ks, 8, N, Ho, Wo, Cout);
    if (c41 < 0.97 * c42) g.mt = 1;
    static int row_on = -1;  // DVSR_CONV_DMAROW=0: the caller falls back to the single-buffered kernel (A/B aid)
    if (row_on < 0) {
      const char* v = getenv("DVSR_CONV_DMAROW");
      row_on = v ? atoi(v) : 1;
    }
    // (any Cout: even the 16 -> 2 flow head, 2 of 32 tile rows used, gains over the single-buffered kernel:
    // TOFlow forward 6.8 -> 6.1 ms, forward+backward 48.3 -> 45.0 ms with it)
    if (row_on && (allow_ksplit & 2) && stride == 1 && Wo % 4 == 0) g.dma = 2;
    return g;
  }
  // Small grids: the K-split kernel (geo {32, NT, MT} with ks == 3).  DVSR_CONV_KSPLIT_BELOW=<workgroups of the
  // 4x32x32 geometry> moves the threshold (0 disables); DVSR_CONV_KSPLIT_NT=1|2 pins the tile shape.
  // Threshold from profiles/r02_small_grid_ab.txt: below ~700 such workgroups (the 44x80 levels: 66..330) the K-split
  // kernel wins (rc_rb 19.2 -> 14.8 us, fe_rb 30 -> 26 us); at 900 (the 180x320 trunk) and 1155 (L1_om at 5x44x80) the
  // pipelined 4-row kernel is faster again (47.5 vs 52.6 us, 65 vs 72 us): three times the halo and unshared weights.
  if (ks == 3 && stride == 1 && (allow_ksplit & 1) && Ctot >= 32 && Cout >= 32) {
    static int below = -1, pin_nt = -1;
    if (below < 0) {
      const char* v = getenv("DVSR_CONV_KSPLIT_BELOW");
      below = v ? atoi(v) : 700;
      const char* n = getenv("DVSR_CONV_KSPLIT_NT");
      pin_nt = n ? atoi(n) : 0;
    }
    const long long wg41 = (long long)ceil_div(Wo, 32) * ceil_div(Ho, 4) * N * ceil_div(Cout, 32);
    if (wg41 < below) {
      // 1 row x 64 couts (even row count not needed) or 2 rows x 32 couts: the latter halves the weight traffic
      // per workgroup but needs an even split of the rows and Cout in 32-blocks; 64-wide blocks waste less when
      // Cout % 64 == 0
      const bool nt2 = pin_nt ? pin_nt == 2 : (Cout % 64 != 0 && Ho % 2 == 0);
      return nt2 ? ConvGeo{32, 2, 1} : ConvGeo{32, 1, 2};
    }
  }
  static int force = -2;  // DVSR_CONV_TILE=0|1|2 pins (8,2)/(4,2)/(4,1) tiles (A/B aid); default: model
  if (force == -2) {
    const char* v = getenv("DVSR_CONV_TILE");
    force = (v && v[0] >= '0' && v[0] <= '2') ? v[0] - '0' : -1;
  }
  const int cc = conv2_cc(ks, stride);
  if (ks == 3 && stride == 2) {  // the two pyramid convs: few tiles, pick by tile quantisation alone
    const double c82 = conv2_pipe_cost(8, 2, 9, cc, N, Ho, Wo, Cout);
    const double c42 = conv2_pipe_cost(4, 2, 9, cc, N, Ho, Wo, Cout);
    const double c41 = conv2_pipe_cost(4, 1, 9, cc, N, Ho, Wo, Cout);
    if (force == 0 || (force < 0 && c82 <= c42 && c82 <= c41)) return ConvGeo{cc, 8, 2};
    if (force == 2 || (force < 0 && c41 < 0.97 * c42)) return ConvGeo{cc, 4, 1};
    return ConvGeo{cc, 4, 2};
  }
  const double c42 = conv2_pipe_cost(4, 2, ks * ks, cc, N, Ho, Wo, Cout);
  const double c41 = conv2_pipe_cost(4, 1, ks * ks, cc, N, Ho, Wo, Cout);
  ConvGeo g = c41 < 0.97 * c42 ? ConvGeo{cc, 4, 1} : ConvGeo{cc, 4, 2};
  if (force == 0 && ks != 2) g = ConvGeo{cc, 8, 2};
  if (force == 1) g = ConvGeo{cc, 4, 2};
  if (force == 2) g = ConvGeo{cc, 4, 1};
  // halo by DMA (conv2d_dma_item): plain pad-1 inputs on a 16-byte column grid, whole 8-channel chunks.
  // DVSR_CONV_DMA=0 keeps the register-staged kernel (A/B aid).
  static int dma_on = -1;
  if (dma_on < 0) {
    const char* v = getenv("DVSR_CONV_DMA");
    dma_on = v ? atoi(v) : 1;
  }
  if (dma_on && (allow_ksplit & 2) && ks == 3 && stride == 1 && Wo % 4 == 0 && Ctot % 8 == 0) g.dma = 1;
  if (force >= 0) return g;
  // Winograd F(2x2, 3x3) (conv2d_wino.hip): one workgroup per CU, 64 couts x 64 2x2-pixel tiles, 4/9 of the MFMAs plus the
  // transforms.  Model: rounds over the 256 CUs x (MFMA cycles of a workgroup's chunks + transform / epilogue overhead)
  // against the direct kernel's rounds x MFMA cycles at its measured 0.78 efficiency.  DVSR_CONV_WINO=0 disables,
  // =2 takes it wherever it is eligible (A/B aid).
  int wino_on = 1;   // (read per call: plans are built once, and the tests switch it at run time)
  if (const char* v = getenv("DVSR_CONV_WINO")) wino_on = atoi(v);
  if (wino_on && g.dma == 1 && (allow_ksplit & 4) && Cout >= 32 && Ctot >= 16) {
    const int nch = Ctot / 8;
    // (one workgroup per CU: rounds over the device's CUs; cycle figures measured on MI355X: tools/wino_trace.py, 2.07 GHz
    // under this kernel)
    const double cus = (double)device_cus();
    auto wino_cost = [&](int oh, int ow) {
      const double wgs = (double)ceil_div(Wo, ow) * ceil_div(Ho, oh) * N * ceil_div(Cout, 64);
      return ceil(wgs / cus) * (nch * 5100.0 + 12500.0) / 2.07;
    };
    const double w4 = wino_cost(4, 64), w8 = wino_cost(8, 32);
    // (the direct kernels reach 0.78 of their MFMA time only over many rounds of workgroups; on grids of a few rounds their
    // prologue / epilogue is exposed: L2_fea 5x64->64 @90x160 measures 0.60 -- profiles/r03_c_per_launch_fwd180x320.txt)
    const double dcyc = std::min(c42, c41);
    const double drounds = dcyc / (64.0 * 9 * 4 * (c41 < 0.97 * c42 ? 1 : 2));
    const double direct = dcyc * nch / (drounds <= 6.0 ? 0.60 : 0.78) / 2.4;
    // DVSR_CONV_WINO3 (default 1): the same GEMMs on the bf16 pipe with the exact 3-way operand split (conv2d_wino3.hip)
    int wino3_on = 1;
    if (const char* v = getenv("DVSR_CONV_WINO3")) wino3_on = atoi(v);
    // ... which also has a 16 x 16-pixel tile (th = 16): the 44x80 levels of the batched inner step fill 92 % of its tiles
    // against 72 % of the 8 x 32 ones (DVSR_CONV_WINO_T16=0: off)
    static const bool t16_on = [] { const char* v = getenv("DVSR_CONV_WINO_T16"); return !(v && v[0] == '0'); }();
    const double w16 = (wino3_on && t16_on) ? wino_cost(16, 16) : 1e300;
    const double best = std::min(std::min(w4, w8), w16);
    // (both Winograd kernels hold a workgroup's whole working set in ~150 KB of LDS: gfx950's 160 KB, checked, not assumed)
    // DVSR_CONV_WINO5 (read per call): Winograd F(4x4, 3x3) on the bf16 pipe (conv2d_wino5.hip: 32 tiles of 4x4 outputs x 64
    // couts per workgroup, 1024 threads, 156 KB of LDS).  Forward launches only (allow bit 3: plain / residual / PixelShuffle(2)
    // stores of 4-pixel tile rows -- Wo % 4 == 0 --, no accumulate / gradient mask).  0: off, 1: where the model says it is faster (default),
    // 2: wherever it is eligible (A/B aid), 3: eligible and 16x32-pixel workgroup tiles (A/B aid).
    int wino5_on = 1;
    if (const char* v = getenv("DVSR_CONV_WINO5")) wino5_on = atoi(v);
    if (wino5_on && wino3_on && (allow_ksplit & 8) && Wo % 4 == 0 && device_lds_optin() >= (size_t)156 * 1024) {
      auto w5_cost = [&](int oh, int ow) {
        const double wgs = (double)ceil_div(Wo, ow) * ceil_div(Ho, oh) * N * ceil_div(Cout, 64);
        return ceil(wgs / cus) * (nch * W5_CHUNK_CYC + W5_FIXED_CYC) / 2.07;
      };
      const double f8 = w5_cost(8, 64), f16 = w5_cost(16, 32);
      const double best5 = std::min(f8, f16);
      // (the model flatters this kernel on grids of one round -- the N = 1 trunk at 180x320 measures 31 us against 26 --: it has
      // to win by 15 %)
      if (wino5_on >= 2 || (1.15 * best5 < best && best5 < direct)) return ConvGeo{8, wino5_on == 3 ? 16 : (f16 < f8 ? 16 : 8), 2, 0, 5};
    }
    if ((wino_on == 2 || best < direct) && device_lds_optin() >= (size_t)155 * 1024)
      return ConvGeo{8, (w16 < w4 && w16 < w8) ? 16 : (w8 < w4 ? 8 : 4), 2, 0, wino3_on ? 4 : 3};
  }
  // Small grids (every workgroup resident at once) are bound by one memory latency per chunk, not by the
  // matrix pipe: 16-channel chunks halve the number of exposed latencies.  DVSR_CONV_CC16_BELOW=<workgroups>
  // moves the threshold (0 disables).
  static int cc16_below = -1;
  if (cc16_below < 0) {
    const char* v = getenv("DVSR_CONV_CC16_BELOW");
    cc16_below = v ? atoi(v) : 0;
  }
  if (ks == 3 && stride == 1 && Ctot >= 32) {
    const long long wgs = (long long)ceil_div(Wo, 32) * ceil_div(Ho, 4) * N * ceil_div(Cout, 32 * g.mt);
    if (wgs <= cc16_below) g.cc = 16;
  }
  return g;
}

int conv2_pch_cc(int ks, int cc, int bf, int dma) {
  if (dma == 3) return 16 * 2 * 64 * 4;   // Winograd image: 16 transformed taps x 8 channels x 64 couts
  if (dma == 4) return 2 * 6144;          // the same as three bf16 pieces: 2 phases x 24 KB
  if (dma == 5) return 18 * 2 * 3 * 256;  // F(4x4, 3x3): 18 point pairs x 2 cout halves x 3 fragments of 1 KB (X, X', L)
  return (bf == 2 ? 3 : 1) * 2 * ks * ks * (bf ? cc / 16 : cc / 8) * 2 * 32 * 4;
}

// `wp` = weights packed by pack_weights_kernel for this (ks, wt, geo.cc) combination.
#ifdef DVSR_CONV_TRACE
static long long* g_trace_buf = nullptr;
static int g_trace_countdown = -1;
// the launch_index-th conv2d_packed_run call from now on writes its timeline to buf
extern "C" int dvsr_debug_conv_trace(void* buf, int launch_index) {
  g_trace_buf = (long long*)buf;
  g_trace_countdown = launch_index;
  return 0;
}
#endif

int conv2d_packed_prepare(const dvsr_conv2d_desc& d, const float* wp, const ConvExtra& ex, const ConvGeo& geo, ConvK2* out) {
  DVSR_REQUIRE(d.x0 && wp && d.y, DVSR_ERR_INVALID, "conv2d_packed: null x0/wp/y");
  DVSR_REQUIRE(((d.ks == 1 || d.ks == 2) && d.stride == 1) || (d.ks == 3 && (d.stride == 1 || d.stride == 2)) ||
                   ((d.ks == 7 || d.ks == 9) && d.stride == 1 && geo.dma == 2),
               DVSR_ERR_UNSUPPORTED, "conv2d_packed: ks=%d stride=%d", d.ks, d.stride);
  DVSR_REQUIRE(d.c1 == 0 || (d.c0 % geo.cc == 0 && !ex.in_ps && !ex.in_dil), DVSR_ERR_UNSUPPORTED,
               "conv2d_packed: two inputs need c0 %% %d == 0 and a plain first input (c0=%d)", geo.cc, d.c0);
  ConvK2& k = *out;
  k.x0 = d.x0; k.x1 = d.x1; k.wp = wp; k.bias = d.bias; k.res = d.res; k.y = d.y;
  k.N = d.N; k.c0 = d.c0; k.c1 = d.c1; k.H = d.H; k.W = d.W; k.Cout = d.Cout;
  k.pad = d.pad; k.act = d.act; k.ps = d.pixel_shuffle; k.x1_bdiv = d.x1_bdiv > 0 ? d.x1_bdiv : 1;
  k.x0_bs = d.x0_bstride > 0 ? d.x0_bstride : (long long)d.c0 * d.H * d.W;
  k.x1_bs = d.x1_bstride > 0 ? d.x1_bstride : (long long)d.c1 * d.H * d.W;
  k.Ho = (d.H + 2 * d.pad - d.ks) / d.stride + 1;
  k.Wo = (d.W + 2 * d.pad - d.ks) / d.stride + 1;
  k.nchunks = ceil_div(d.c0 + d.c1, geo.cc);
  k.in_ps = ex.in_ps; k.in_dil = ex.in_dil; k.Hs = ex.Hs; k.Ws = ex.Ws; k.accum = ex.accum;
  k.gmask = ex.gmask; k.gmask_act = ex.gmask_act;
  k.wdiv = ex.wdiv > 0 ? ex.wdiv : 1; k.w_gs = ex.w_gs; k.b_gs = ex.b_gs;
#ifdef DVSR_CONV_TRACE
  {
    // measurement aid of the debug build, results are WRONG when set (profiles/r02_z_conv_dma_ablation.txt): bit 0 no
    // stores, bit 1 no halo DMA, bit 2 no weight DMA, bit 3 no chunk barriers (conv2d_dma_kernel only)
    static int ablate = -1;
    if (ablate < 0) {
      const char* v = getenv("DVSR_CONV_ABLATE");
      ablate = v ? atoi(v) : 0;
    }
    k.ablate = ablate;
  }
#endif
  if (k.in_ps) k.x0_bs = (long long)d.c0 * d.H * d.W;
  if (k.in_dil) k.x0_bs = (long long)d.c0 * ex.Hs * ex.Ws;
#ifdef DVSR_CONV_TRACE
  k.trace = (g_trace_countdown == 0) ? g_trace_buf : nullptr;
  if (g_trace_countdown >= 0) --g_trace_countdown;
#endif
  if (geo.dma == 2) {
    DVSR_REQUIRE((d.ks == 7 || d.ks == 9) && d.stride == 1 && d.pad == d.ks / 2 && !ex.in_ps && !ex.in_dil && geo.cc == 8 &&
                     geo.th == 4 && d.W % 4 == 0 && (d.c1 == 0 || d.c0 % 8 == 0) && k.x0_bs % 4 == 0 && k.x1_bs % 4 == 0 &&
                     ((uintptr_t)d.x0 & 15) == 0 && ((uintptr_t)d.x1 & 15) == 0,
                 DVSR_ERR_UNSUPPORTED, "conv2d_packed: the row-split DMA kernel needs 7x7 / 9x9, stride 1, pad ks/2, plain "
                 "16-byte aligned inputs and W %% 4 == 0 (W=%d c0=%d c1=%d)", d.W, d.c0, d.c1);
  } else if (geo.dma) {
    DVSR_REQUIRE(geo.dma < 3 || d.c0 + d.c1 >= 16, DVSR_ERR_UNSUPPORTED, "conv2d_packed: the Winograd kernel needs two 8-channel chunks");
    DVSR_REQUIRE(geo.dma != 5 || (!ex.accum && !ex.gmask && d.W % 4 == 0 && (geo.th == 8 || geo.th == 16)), DVSR_ERR_UNSUPPORTED,
                 "conv2d_packed: the F(4x4, 3x3) kernel stores whole tile columns of forward launches (W=%d th=%d)", d.W, geo.th);
    DVSR_REQUIRE(geo.dma < 3 || d.pixel_shuffle == 0 || (d.pixel_shuffle == 2 && d.Cout % 4 == 0 && !d.res && !ex.accum && !ex.gmask),
                 DVSR_ERR_UNSUPPORTED, "conv2d_packed: the Winograd kernel stores plain or PixelShuffle(2) tiles (ps=%d)", d.pixel_shuffle);
    DVSR_REQUIRE(d.ks == 3 && d.stride == 1 && d.pad == 1 && !ex.in_ps && !ex.in_dil && geo.cc == 8 && (geo.th == 4 || geo.th == 8 || (geo.th == 16 && geo.dma >= 4)) &&
                     d.W % 4 == 0 && d.c0 % 8 == 0 && d.c1 % 8 == 0 && k.x0_bs % 4 == 0 && k.x1_bs % 4 == 0 &&
                     ((uintptr_t)d.x0 & 15) == 0 && ((uintptr_t)d.x1 & 15) == 0,
                 DVSR_ERR_UNSUPPORTED, "conv2d_packed: the DMA-halo kernel needs 3x3/s1/pad 1, plain 16-byte aligned inputs, "
                 "W %% 4 == 0 and channel counts %% 8 == 0 (W=%d c0=%d c1=%d)", d.W, d.c0, d.c1);
  }
  if (d.ks == 3 && geo.cc == 32) {  // K-split small-grid kernel: tile bookkeeping of its launch
    DVSR_REQUIRE(d.stride == 1 && (d.c1 == 0 || d.c0 % 32 == 0) && !ex.in_ps && !ex.in_dil && d.pad == 1 &&
                     ((geo.th == 1 && geo.mt == 2) || (geo.th == 2 && geo.mt == 1)),
                 DVSR_ERR_UNSUPPORTED, "conv2d_packed: the K-split kernel needs 3x3/s1/pad 1, plain inputs, c0 %% 32 == 0 "
                 "with two inputs (th=%d mt=%d)", geo.th, geo.mt);
    k.tiles_x = ceil_div(k.Wo, 32); k.tiles_y = ceil_div(k.Ho, geo.th); k.ntiles = k.tiles_x * k.tiles_y * k.N;
    k.ncb = ceil_div(k.Cout, 32 * geo.mt);
    k.tiles_per_xcd = ceil_div(k.ntiles, 8);
    k.nitems = k.tiles_per_xcd * 8 * k.ncb;
  }
  return DVSR_OK;
}

int conv2d_packed_run(const dvsr_conv2d_desc& d, const float* wp, const ConvExtra& ex, const ConvGeo& geo,
                      hipStream_t st) {
  ConvK2 k;
  int rc = conv2d_packed_prepare(d, wp, ex, geo, &k);
  if (rc) return rc;
  const int code = geo.cc * 100 + geo.th * 10 + geo.mt;
  if (geo.bf) {
    DVSR_REQUIRE((d.ks == 3 || (d.ks == 2 && geo.bf == 2)) && d.stride == 1 && geo.cc == 16 && (geo.th == 4 || (geo.bf == 2 && geo.th == 8)),
                 DVSR_ERR_UNSUPPORTED,
                 "conv2d_packed: the bf16 kernel exists for 3x3 (split: also 2x2) stride-1 convs with 16-channel chunks");
    if (geo.bf == 2 && d.ks == 2) {   // the estimators' 4x4 stride-2 convolutions in their 2x2 space-to-depth form
      if (geo.th == 8) {
        if (geo.mt == 2) return launch_conv2<2, 1, 16, 8, 2, 2>(k, st);
        return launch_conv2<2, 1, 16, 8, 1, 2>(k, st);
      }
      if (geo.mt == 2) return launch_conv2<2, 1, 16, 4, 2, 2>(k, st);
      return launch_conv2<2, 1, 16, 4, 1, 2>(k, st);
    }
    if (geo.bf == 2) {
      if (geo.th == 8) {
        if (geo.mt == 2) return launch_conv2<3, 1, 16, 8, 2, 2>(k, st);
        return launch_conv2<3, 1, 16, 8, 1, 2>(k, st);
      }
      if (geo.mt == 2) return launch_conv2<3, 1, 16, 4, 2, 2>(k, st);
      return launch_conv2<3, 1, 16, 4, 1, 2>(k, st);
    }
    if (geo.mt == 2) return launch_conv2<3, 1, 16, 4, 2, 1>(k, st);
    return launch_conv2<3, 1, 16, 4, 1, 1>(k, st);
  }
  if (geo.dma == 3) return conv2d_wino_launch(k, geo.th, st);
  if (geo.dma == 4) return conv2d_wino3_launch(k, geo.th, st);
  if (geo.dma == 5) return conv2d_wino5_launch(k, geo.th, st);
  if (geo.dma == 2) {
    if (d.ks == 7) return geo.mt == 2 ? launch_dmarow<7, 4, 2>(k, st) : launch_dmarow<7, 4, 1>(k, st);
    return geo.mt == 2 ? launch_dmarow<9, 4, 2>(k, st) : launch_dmarow<9, 4, 1>(k, st);
  }
  if (geo.dma) {
    if (geo.th == 8) return geo.mt == 2 ? launch_dma<8, 2>(k, st) : launch_dma<8, 1>(k, st);
    return geo.mt == 2 ? launch_dma<4, 2>(k, st) : launch_dma<4, 1>(k, st);
  }
  if (d.ks == 3 && geo.cc == 32) {  // K-split small-grid kernel
    if (geo.mt == 2) return launch_ksplit<2, 1>(k, st);
    return launch_ksplit<1, 2>(k, st);
  }
  if (d.ks == 3 && d.stride == 2) {
    switch (code) {
      case 882: return launch_conv2<3, 2, 8, 8, 2>(k, st);
      case 842: return launch_conv2<3, 2, 8, 4, 2>(k, st);
      case 841: return launch_conv2<3, 2, 8, 4, 1>(k, st);
    }
  } else
  if (d.ks == 3) {
    switch (code) {
      case 882: return launch_conv2<3, 1, 8, 8, 2>(k, st);
      case 842: return launch_conv2<3, 1, 8, 4, 2>(k, st);
      case 841: return launch_conv2<3, 1, 8, 4, 1>(k, st);
      case 1642: return launch_conv2<3, 1, 16, 4, 2>(k, st);
      case 1641: return launch_conv2<3, 1, 16, 4, 1>(k, st);
    }
  } else if (d.ks == 2) {  // the estimator's 4x4 stride-2 convs, re-expressed over a space-to-depth input
    switch (code) {
      case 1642: return launch_conv2<2, 1, 16, 4, 2>(k, st);
      case 1641: return launch_conv2<2, 1, 16, 4, 1>(k, st);
    }
  } else {
    switch (code) {
      case 3282: return launch_conv2<1, 1, 32, 8, 2>(k, st);
      case 3242: return launch_conv2<1, 1, 32, 4, 2>(k, st);
      case 3241: return launch_conv2<1, 1, 32, 4, 1>(k, st);
    }
  }
  DVSR_REQUIRE(false, DVSR_ERR_INVALID, "conv2d_packed: no kernel for ks=%d cc=%d th=%d mt=%d", d.ks, geo.cc, geo.th,
               geo.mt);
}

}  // namespace dvsr

// ---- op-level entries to the pipelined / K-split kernels ---------------------------------------------------
// dvsr_conv2d_forward runs the un-packed kernel (no workspace).  These take a caller workspace for the packed weight
// image, pack it on the stream and run the geometry the engine would choose -- used by the op-composed backbones
// (TOFlow head, DUF) whose 3x3 / 1x1 convolutions otherwise ran on the single-buffered kernel.
namespace {
struct OpPack {
  dvsr::ConvGeo geo;
  size_t floats;
};
OpPack op_pack(int ks, int stride, int pad, int N, int Ho, int Wo, int Cout, int Ctot, bool plain, bool aligned = false) {
  using namespace dvsr;
  OpPack o;
  const bool k3 = ks == 3 && stride == 1 && pad == 1, kbig = (ks == 7 || ks == 9) && stride == 1 && pad == ks / 2;
  o.geo = conv2_choose(ks, stride, N, Ho, Wo, Cout, Ctot, (k3 && plain ? 1 : 0) | ((k3 || kbig) && aligned ? 2 : 0) | (k3 && aligned ? 4 | 8 : 0));
  o.floats = (size_t)ceil_div(Cout, 64) * ceil_div(Ctot, o.geo.cc) * conv2_pch_cc(ks, o.geo.cc, 0, o.geo.dma);
  return o;
}
OpPack op_pack_for(const dvsr_conv2d_desc& d, int Cout, int Ctot) {
  // (whole 8-channel chunks for the 3x3 DMA kernel; the row-split 7x7 / 9x9 kernel takes a partial last chunk)
  const bool big = d.ks == 7 || d.ks == 9;
  const bool aligned = (((uintptr_t)d.x0 | (uintptr_t)d.x1) & 15) == 0 && (big ? (d.c1 == 0 || d.c0 % 8 == 0) : (d.c0 % 8 == 0 && d.c1 % 8 == 0)) &&
                       d.x0_bstride % 4 == 0 && d.x1_bstride % 4 == 0 && (d.H * d.W) % 4 == 0;
  return op_pack(d.ks, 1, d.pad, d.N, d.H, d.W, Cout, Ctot, d.c1 == 0 || d.c0 % 32 == 0, aligned);
}
int op_run(const dvsr_conv2d_desc& d, const dvsr::ConvExtra& ex, int Cout, int Ctot, void* ws, size_t bytes, hipStream_t st) {
  using namespace dvsr;
  DVSR_REQUIRE((d.ks == 1 || d.ks == 3 || d.ks == 7 || d.ks == 9) && d.stride == 1 && d.pad == d.ks / 2, DVSR_ERR_UNSUPPORTED,
               "conv2d (packed): ks=%d stride=%d pad=%d (1x1 / 3x3 / 7x7 / 9x9, stride 1, pad ks/2)", d.ks, d.stride, d.pad);
  const OpPack o = op_pack_for(d, Cout, Ctot);
  DVSR_REQUIRE(d.ks <= 3 || o.geo.dma == 2, DVSR_ERR_UNSUPPORTED, "conv2d (packed): this %dx%d convolution is not eligible for the "
               "row-split kernel (dvsr_conv2d_packed_geometry): use dvsr_conv2d_forward / _backward", d.ks, d.ks);
  DVSR_REQUIRE(ws && bytes >= o.floats * sizeof(float), DVSR_ERR_WORKSPACE, "conv2d (packed): workspace %zu < %zu bytes", bytes,
               o.floats * sizeof(float));
  PackTable t;
  t.n = 1;
  PackEntry& e = t.e[0];
  e.w = d.w; e.P = (float*)ws; e.Cout = Cout; e.Ctot = Ctot; e.KK = d.ks * d.ks; e.CC = o.geo.cc; e.wt = ex.wt;
  e.w_ctot = ex.w_ctot; e.w_coff = ex.w_coff; e.ncb = ceil_div(Cout, 64); e.nchunks = ceil_div(Ctot, e.CC); e.bf = 0;
  e.perm = o.geo.dma;
  e.pch = conv2_pch_cc(d.ks, e.CC, 0, o.geo.dma);
  int rc = pack_weights_run(t, st);
  if (rc) return rc;
  return conv2d_packed_run(d, (const float*)ws, ex, o.geo, st);
}
}  // namespace

extern "C" size_t dvsr_conv2d_packed_workspace_bytes(const dvsr_conv2d_desc* d) {
  if (!d || (d->ks != 1 && d->ks != 3 && d->ks != 7 && d->ks != 9)) return 0;
  // the larger of the forward pack and the data-gradient pack (roles of Cout and Ctot swapped)
  const int ctot = d->c0 + d->c1;
  const size_t a = op_pack(d->ks, 1, d->ks / 2, d->N, d->H, d->W, d->Cout, ctot, true).floats;
  const size_t b = op_pack(d->ks, 1, d->ks / 2, d->N, d->H, d->W, ctot, d->Cout, true).floats;
  const size_t c = op_pack(d->ks, 1, d->ks / 2, d->N, d->H, d->W, d->Cout, ctot, false).floats;
  // (the Winograd images of a 3x3 layer: 16 transformed taps instead of 9 -- F(4x4, 3x3): 36, as two fragments each)
  const size_t w = d->ks == 3 ? (size_t)std::max(dvsr::ceil_div(d->Cout, 64) * dvsr::ceil_div(ctot, 8), dvsr::ceil_div(ctot, 64) * dvsr::ceil_div(d->Cout, 8)) *
                                    std::max(dvsr::conv2_pch_cc(3, 8, 0, 4), dvsr::conv2_pch_cc(3, 8, 0, 5)) : 0;
  return std::max(std::max(a, b), std::max(c, w)) * sizeof(float);
}

extern "C" int dvsr_conv2d_packed_geometry(const dvsr_conv2d_desc* d, int geo[4]) {
  DVSR_REQUIRE(d && geo && (d->ks == 1 || d->ks == 3 || d->ks == 7 || d->ks == 9) && d->stride == 1 && d->pad == d->ks / 2,
               DVSR_ERR_INVALID, "conv2d_packed_geometry: 1x1 / 3x3 / 7x7 / 9x9 stride-1 descriptors only");
  const OpPack o = op_pack_for(*d, d->Cout, d->c0 + d->c1);
  geo[0] = o.geo.cc; geo[1] = o.geo.th; geo[2] = o.geo.mt; geo[3] = o.geo.dma;
  return DVSR_OK;
}

extern "C" int dvsr_conv2d_forward_packed(const dvsr_conv2d_desc* d, void* workspace, size_t workspace_bytes,
                                          dvsr_stream_t stream) {
  DVSR_REQUIRE(d && d->x0 && d->w && d->y, DVSR_ERR_INVALID, "conv2d_forward_packed: null argument");
  DVSR_REQUIRE(d->pixel_shuffle == 0 || (d->pixel_shuffle == 2 && d->Cout % 4 == 0 && !d->res), DVSR_ERR_INVALID,
               "conv2d_forward_packed: pixel_shuffle needs Cout %% 4 == 0 and no residual");
  return op_run(*d, dvsr::ConvExtra(), d->Cout, d->c0 + d->c1, workspace, workspace_bytes, (hipStream_t)stream);
}

// gx0 = data gradient of a single-input conv (d->c1 == 0): a stride-1 conv of gy with the transposed, tap-mirrored
// weights.  gy: gradient w.r.t. the pre-activation output.
extern "C" int dvsr_conv2d_dgrad_packed(const dvsr_conv2d_desc* d, const float* gy, float* gx0, void* workspace,
                                        size_t workspace_bytes, dvsr_stream_t stream) {
  using namespace dvsr;
  DVSR_REQUIRE(d && gy && gx0 && d->w, DVSR_ERR_INVALID, "conv2d_dgrad_packed: null argument");
  DVSR_REQUIRE(d->c1 == 0 && d->pixel_shuffle == 0, DVSR_ERR_UNSUPPORTED, "conv2d_dgrad_packed: single plain input only");
  dvsr_conv2d_desc g = {};
  g.x0 = gy; g.w = d->w; g.y = gx0; g.N = d->N; g.c0 = d->Cout; g.Cout = d->c0; g.H = d->H; g.W = d->W; g.ks = d->ks;
  g.stride = 1; g.pad = d->ks / 2; g.act = ACT_NONE; g.x1_bdiv = 1;
  ConvExtra ex;
  ex.wt = 1; ex.w_ctot = d->c0; ex.w_coff = 0;
  return op_run(g, ex, d->c0, d->Cout, workspace, workspace_bytes, (hipStream_t)stream);
}

