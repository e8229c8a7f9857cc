// Modulated deformable convolution forward (deform_conv.py:258-291 -> deform_conv_cuda.cpp:486-590 ->
// deform_conv_cuda_kernel.cu:466-633), round 5: the DMA-staged sampler kernel of mdcn.hip with its 64 x 576 contraction on the
// bf16 matrix pipe under the exact 3-way operand split (conv2d_wino4.hip's arithmetic: six bf16 products per fp32 product,
// fp32 accumulate -- fp32 results).
//
// Why.  mdcn_fwd_dma_kernel contracts on v_mfma_f32_32x32x2_f32, which IS the fp32 vector datapath: the ~114 vector
// instructions a tap's sampler needs (geometry of two pixel rows, 32 blend FMAs, two mask sigmoids) do not overlap its sixteen
// MFMAs (1024 cycles), a tap took 1770 cycles against 1044 without the sampler (profiles/r02_z_dcn_dma_trace.txt), and the
// matrix pipe stayed 40 % busy.  A bf16 MFMA runs beside vector instructions of either wave of its SIMD
// (profiles/r04_mfma_overlap.txt), and a tap's six products are twelve v_mfma_f32_32x32x16_bf16 = 384 cycles.
//
// What changes against the fp32 kernel:
//   * B operand: the lane halves are PIXEL ROWS now -- lane (lo, hi) samples all EIGHT channels of pixel (row 2 wave + hi,
//     column lo), so the geometry of a pixel (mask sigmoid, floor, window address, four corner weights) is computed once
//     instead of once per lane half (~55 of a tap's vector instructions) --; the eight blended values are split into three
//     bf16 pieces as four channel pairs (11 vector instructions per pair) and v_permlane32_swap turns (hi, mid) / (hi copy,
//     lo) of the two rows into the fragments B1[row] = (hi | mid), B2[row] = (hi | lo): K = 16 = 8 channels x 2 pieces, lanes
//     32-63 carry K 8..15 (conv2d_wino4.hip's exchange);
//   * A operand: pack_weights_dcn3_kernel lays the weights out as [tap][piece][cout 64][8 channels] bf16 (16-byte records:
//     the fragment layout), read straight from global memory one tap ahead (A1 = (hi | hi), A2 = (mid | hi), A3 = (lo | mid)
//     by lane half) -- no weight image in the LDS (54 KB per workgroup instead of 72), no weight DMA;
//   * per tap and wave: acc[mt][nt] += A1 B1 + A3 B1 + A2 B2 on the 2 x 2 blocks = 12 MFMAs.
// Window / offset / mask staging, the sampler's pieces, the exact global fix-up for samples outside the window and the
// epilogue are those of mdcn_fwd_dma_kernel.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace dvsr {

int mdcn_fwd_variant() {
  static const int variant = [] {
    const char* v = getenv("DVSR_DCN_FWD");
    if (!v || !v[0] || v[0] == 's') return 3;
    return v[0] == 'r' ? 2 : 0;   // (the LDS-column-tile kernel, 'lds', was retired in round 6)
  }();
  return variant;
}
int mdcn_pack_floats() { return 6912; }   // 9 taps x 3 pieces x 64 couts x 8 channels bf16 (the fp32 pack needs 4608)
int mdcn_pack_perm(int W) { return (mdcn_fwd_variant() == 3 && W % 4 == 0) ? 6 : 0; }

typedef float ds2f __attribute__((ext_vector_type(2)));
typedef __bf16 dsbf2 __attribute__((ext_vector_type(2)));
typedef __bf16 dsbf8 __attribute__((ext_vector_type(8)));
typedef unsigned dsu4 __attribute__((ext_vector_type(4)));

// P16[cb][k][tap][piece][cout 64][slot 8] = piece of W(cout = cb*64 + col, cin = k*8 + slot, tap)
__global__ void pack_weights_dcn3_kernel(PackTable t) {
  const PackEntry& e = t.e[blockIdx.y];
  if (e.perm != 6) return;
  __bf16* const P16 = reinterpret_cast<__bf16*>(e.P);
  const size_t total = (size_t)e.ncb * e.nchunks * 9 * 512;   // (tap, cout, slot) triples incl. padding
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(i & 7), col = (int)((i >> 3) & 63);
    const size_t r = i >> 9;
    const int tap = (int)(r % 9);
    const size_t ck = r / 9;
    const int k = (int)(ck % e.nchunks), cb = (int)(ck / e.nchunks);
    const int co = cb * 64 + col, ci = k * 8 + s;
    float v = 0.f;
    if (co < e.Cout && ci < e.Ctot) v = e.w[((size_t)co * e.Ctot + ci) * 9 + tap];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    __bf16* d = P16 + ck * 13824 + (size_t)tap * 1536 + (size_t)col * 8 + s;
    d[0] = h;
    d[512] = m;
    d[1024] = l;
  }
}

int pack_weights_dcn3_run(const PackTable& t, hipStream_t st) {
  hipLaunchKernelGGL(pack_weights_dcn3_kernel, dim3(32, t.n), dim3(256), 0, st, t);
  return check_launch("pack_weights_dcn3_kernel");
}

#ifdef DVSR_CONV_TRACE
#define DCS_STAMP(i)                                                                               \
  do {                                                                                             \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define DCS_STAMP(i) \
  do {               \
  } while (0)
#endif

struct DcnSplitShape {
  static constexpr int CPG = 8, KK = 9, TH = 8, TW = 32, HALO = 4, X0 = 8;
  static constexpr int XH = TH + 2 + 2 * HALO, XW = 48, XG = XW / 4, XCH = XH * XW;
  static constexpr int NXG = CPG * XH * XG, NXI = (NXG + 255) / 256;
  static constexpr int OMP = 27, NOI = (OMP + 3) / 4;
  static constexpr int X_FLOATS = CPG * XCH, OM_FLOATS = OMP * TH * TW;
  static constexpr int PACK_BYTES = 9 * 3 * 64 * 8 * 2;   // one (cout block, chunk) of the weight pack
  static constexpr size_t LDS_BYTES = (size_t)(X_FLOATS + OM_FLOATS) * sizeof(float);
};

__device__ __forceinline__ unsigned dcs_cvt_pk(float a, float b) {   // {bf16(a) in bits 15:0, bf16(b) in bits 31:16}, RNE
  return __builtin_bit_cast(unsigned, __builtin_convertvector(ds2f{a, b}, dsbf2));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dcs_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
}

template <bool MASK_LOGIT>
__global__ __launch_bounds__(256, 2) void mdcn_fwd_split_kernel(DcnK2 a) {
  using Sh = DcnSplitShape;
  constexpr int KK = Sh::KK, TH = Sh::TH, TW = Sh::TW, XH = Sh::XH, XW = Sh::XW, XG = Sh::XG, XCH = Sh::XCH;
  extern __shared__ __attribute__((aligned(16))) float smem_dcs[];
  float* const s_x = smem_dcs;
  float* const s_om = smem_dcs + Sh::X_FLOATS;

  const int id = blockIdx.x;
  const int tpx = (a.ntiles + 7) >> 3;
  const int q_ = id >> 3;
  const int cb = q_ % a.ncb;
  const int tile = (id & 7) * tpx + q_ / a.ncb;
  if (q_ / a.ncb >= tpx || tile >= a.ntiles) return;
  const int tx_ = tile % a.tiles_x;
  const int t2 = tile / a.tiles_x;
  const int ty_ = t2 % a.tiles_y;
  const int n = t2 / a.tiles_y;
  const int oy0 = ty_ * TH, ox0 = tx_ * TW;
  const int wy0 = oy0 - 1 - Sh::HALO, wx0 = ox0 - Sh::X0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const size_t HW = (size_t)a.H * a.W;
  const float* offn = a.off + (size_t)n * a.off_bstride;
  const float* mskn = a.msk + (size_t)n * a.msk_bstride;
  const int px = ox0 + lo;
  const int py = oy0 + 2 * wave + hi;          // this lane's pixel row (lane halves = the wave's two rows)
  const bool pv = py < a.H && px < a.W;
  const int prow = (2 * wave + hi) * TW + lo;

  // window groups of this lane: L = 64 (wave + 4 jj) + lane = (row, channel, column group) -- the window is laid out
  // [row][channel][column]: the two corners of a row are neighbours and the four channels x two corners of a row lie within
  // 255 dwords of one base address, i.e. ONE ds_read2_b32 per (channel, row) instead of two ds_read_b32
  unsigned xo[Sh::NXI];
  bool xv[Sh::NXI];
#pragma unroll
  for (int jj = 0; jj < Sh::NXI; ++jj) {
    const int L = 64 * (wave + 4 * jj) + lane;
    const int y = L / (Sh::CPG * XG), r = L - y * (Sh::CPG * XG);
    const int c = r / XG, g4 = r - c * XG;
    const int gy = wy0 + y, gx = wx0 + 4 * g4;
    const bool ok = L < Sh::NXG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    xo[jj] = ok ? (unsigned)(((size_t)c * HW + (size_t)gy * a.W + gx) * 4) : 0u;
    xv[jj] = ok;
    if (L < Sh::NXG && !ok) *reinterpret_cast<f32x4*>(s_x + L * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // offset / mask planes: plane p = wave + 4 jj is ONE instruction; this lane moves (row lane / 8, columns 4 (lane % 8) ..)
  const int omy = oy0 + (lane >> 3), omx = ox0 + 4 * (lane & 7);
  const bool omv = omy < a.H && omx < a.W;
  const unsigned omo = omv ? (unsigned)(((size_t)omy * a.W + omx) * 4) : 0u;
  if (!omv) {
#pragma unroll
    for (int jj = 0; jj < Sh::NOI; ++jj)
      if (wave + 4 * jj < Sh::OMP) *reinterpret_cast<f32x4*>(s_om + (wave + 4 * jj) * 256 + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS byte addresses of this lane's operands (inline-asm reads of the tap loop)
  auto lds_addr = [](const float* p) { return (unsigned)(size_t)((__attribute__((address_space(3))) const float*)p); };
  const unsigned a_x = lds_addr(s_x);          // channel c: + c XCH floats
  const unsigned a_om = lds_addr(s_om) + (unsigned)prow * 4u;

  // A fragments: [tap][piece][cout][8 ch] bf16; lane halves read the pieces A1: hi|hi, A2: mid|hi, A3: lo|mid
  const float* wp_cb = wset_ptr(a.wp, a.w_gs, n, a.wdiv) + (size_t)cb * a.nchunks * (Sh::PACK_BYTES / 4);
  const __amdgpu_buffer_rsrc_t wrsrc = dcs_rsrc(wp_cb);
  const unsigned av0 = (unsigned)(lo * 16);
  const unsigned av1 = av0 + (hi ? 0u : 1024u), av2 = av0 + (hi ? 1024u : 2048u);
  f32x4 Aop[2][2][3];   // [tap parity][cout half][A1 / A2 / A3]
  auto a_load = [&](int par, int kc, int tap) __attribute__((always_inline)) {
    const int soff = (kc * KK + tap) * 3072;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        Aop[par][mt][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            wrsrc, (int)((j == 0 ? av0 : (j == 1 ? av1 : av2)) + mt * 512), soff, 0));
  };

  const int sub = a.nchunks / a.dg;  // 8-channel chunks per deformable group (they share its offsets and masks)
  DCS_STAMP(0);
  for (int kc = 0; kc < a.nchunks; ++kc) {
    const int g = kc / sub;
    if (kc < 2) DCS_STAMP(1 + 30 * kc);
    __syncthreads();  // the previous chunk's taps are done with s_x / s_om
    const float* xg = a.x + ((size_t)n * a.C + kc * Sh::CPG) * HW;
    {
      const char* xb = reinterpret_cast<const char*>(xg);
#pragma unroll
      for (int jj = 0; jj < Sh::NXI; ++jj)
        if (xv[jj])
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + xo[jj]),
                                           (__attribute__((address_space(3))) void*)(s_x + 256 * (wave + 4 * jj)), 16, 0, 0);
    }
    if (kc % sub == 0) {
#pragma unroll
      for (int jj = 0; jj < Sh::NOI; ++jj) {
        const int p = wave + 4 * jj;
        if (p < Sh::OMP) {
          const float* pb = p < 18 ? offn + (size_t)(g * 18 + p) * HW : mskn + (size_t)(g * 9 + p - 18) * HW;
          if (omv)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(pb) + omo),
                                             (__attribute__((address_space(3))) void*)(s_om + 256 * p), 16, 0, 0);
        }
      }
    }
    a_load(0, kc, 0);
    if (kc < 2) DCS_STAMP(2 + 30 * kc);
    __syncthreads();  // everything landed (the barrier's vmcnt(0) covers the DMAs)
    if (kc < 2) DCS_STAMP(3 + 30 * kc);

    // ---- the nine taps, software-pipelined by hand as in mdcn_fwd_dma_kernel: the sampler of tap t + 1 is cut into pieces
    // that sit between the twelve MFMAs of tap t; LDS reads are inline asm, their results pass through one s_waitcnt asm
    // before the blends; operands ping-pong between two register sets by tap parity.
    struct Geo { ds2f w12, w34; float h_im, w_im, m, lh, lw; int ry, rx; unsigned addr; bool inwin; };
    auto om_issue = [&](auto TAP, float (&o)[3]) {
      constexpr int tap = decltype(TAP)::value;
      const unsigned ad = a_om;  // (a local: asm operands of a generic lambda do not capture)
      f32x2 hw2;  // (dh, dw): planes 2 tap and 2 tap + 1, 1 KiB = 4 x 64 dwords apart
      asm volatile("ds_read2st64_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read_b32 %1, %2 offset:%5"
                   : "=&v"(hw2), "=&v"(o[2])
                   : "v"(ad), "i"(2 * tap * 4), "i"((2 * tap + 1) * 4), "i"((18 + tap) * 1024));
      o[0] = hw2[0]; o[1] = hw2[1];
    };
    // cp[2 ch] = ((y,x), (y,x+1)), cp[2 ch + 1] = ((y+1,x), (y+1,x+1)) of channel ch; half h = channels 4 h .. 4 h + 3.
    // Bases: addr (row y, channels 0-3), + 768 B (channels 4-7), + 1536 B (row y + 1), + 2304 B
    auto corners_issue = [&](unsigned addr, ds2f (&cp)[16], int h) {
      const unsigned b0 = addr + (h ? 768u : 0u), b1 = b0 + 1536u;
      if (h == 0)
        asm volatile(
            "ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %9 offset1:1\n\t"
            "ds_read2_b32 %2, %8 offset0:48 offset1:49\n\tds_read2_b32 %3, %9 offset0:48 offset1:49\n\t"
            "ds_read2_b32 %4, %8 offset0:96 offset1:97\n\tds_read2_b32 %5, %9 offset0:96 offset1:97\n\t"
            "ds_read2_b32 %6, %8 offset0:144 offset1:145\n\tds_read2_b32 %7, %9 offset0:144 offset1:145"
            : "=&v"(cp[0]), "=&v"(cp[1]), "=&v"(cp[2]), "=&v"(cp[3]), "=&v"(cp[4]), "=&v"(cp[5]), "=&v"(cp[6]), "=&v"(cp[7])
            : "v"(b0), "v"(b1));
      else
        asm volatile(
            "ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %9 offset1:1\n\t"
            "ds_read2_b32 %2, %8 offset0:48 offset1:49\n\tds_read2_b32 %3, %9 offset0:48 offset1:49\n\t"
            "ds_read2_b32 %4, %8 offset0:96 offset1:97\n\tds_read2_b32 %5, %9 offset0:96 offset1:97\n\t"
            "ds_read2_b32 %6, %8 offset0:144 offset1:145\n\tds_read2_b32 %7, %9 offset0:144 offset1:145"
            : "=&v"(cp[8]), "=&v"(cp[9]), "=&v"(cp[10]), "=&v"(cp[11]), "=&v"(cp[12]), "=&v"(cp[13]), "=&v"(cp[14]), "=&v"(cp[15])
            : "v"(b0), "v"(b1));
    };
    static_assert(XW == 48 && Sh::CPG == 8, "corners_issue hard-codes the window pitch");
#define DCS_PIN8(c, o) asm volatile("" : "+v"(c[o]), "+v"(c[o + 1]), "+v"(c[o + 2]), "+v"(c[o + 3]), "+v"(c[o + 4]), "+v"(c[o + 5]), "+v"(c[o + 6]), "+v"(c[o + 7]))
    // the first half: everything but the 10 newest LDS reads has returned (the other 8 corner pairs and the offsets / masks
    // follow them)
    auto landed_a = [&](ds2f (&cp)[16]) {
      asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(cp[0]), "+v"(cp[1]), "+v"(cp[2]), "+v"(cp[3]), "+v"(cp[4]), "+v"(cp[5]), "+v"(cp[6]), "+v"(cp[7]));
    };
    auto landed_b = [&](ds2f (&cp)[16]) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cp[8]), "+v"(cp[9]), "+v"(cp[10]), "+v"(cp[11]), "+v"(cp[12]), "+v"(cp[13]), "+v"(cp[14]), "+v"(cp[15]));
    };
    auto landed_om = [&](float (&o)[3]) { asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2])); };
    auto geom_a = [&](auto TAP, const float (&o)[3], Geo& q) {
      constexpr int tap = decltype(TAP)::value;
      constexpr int ki = tap / 3, kj = tap - ki * 3;
      float m = o[2];
      if (MASK_LOGIT) m = __builtin_amdgcn_rcpf(1.f + __expf(-m));
      q.m = m;
      q.h_im = (float)(py - 1 + ki) + o[0];
      q.w_im = (float)(px - 1 + kj) + o[1];
      asm volatile("" : "+v"(q.m), "+v"(q.h_im), "+v"(q.w_im));
    };
    auto geom_b1 = [&](Geo& q) {
      const float hf = floorf(q.h_im), wf = floorf(q.w_im);
      q.lh = q.h_im - hf; q.lw = q.w_im - wf;
      q.ry = (int)hf - wy0; q.rx = (int)wf - wx0;
      const int ryc = min(max(q.ry, 0), XH - 2), rxc = min(max(q.rx, 0), XW - 2);
      q.addr = a_x + (unsigned)(ryc * (Sh::CPG * XW) + rxc) * 4u;
      asm volatile("" : "+v"(q.lh), "+v"(q.lw), "+v"(q.ry), "+v"(q.rx), "+v"(q.addr));
    };
    auto geom_b2 = [&](Geo& q, int& fix) {
      // (bitwise on purpose: a short-circuit splits the tap into basic blocks)
      const int inwin = ((unsigned)q.ry <= (unsigned)(XH - 2)) & ((unsigned)q.rx <= (unsigned)(XW - 2));
      q.inwin = inwin;
      const float hh = 1.f - q.lh, hw = 1.f - q.lw;
      const float ms = ((int)pv & inwin) ? q.m : 0.f;
      q.w12 = ds2f{hh * hw * ms, hh * q.lw * ms}; q.w34 = ds2f{q.lh * hw * ms, q.lh * q.lw * ms};
      fix |= (int)pv & (inwin ^ 1);  // outside the window: the exact path decides (it applies the image gate itself)
      asm volatile("" : "+v"(q.w12), "+v"(q.w34), "+v"(fix));
    };
    // channels k0, k0 + 1: the two corners of a row are one packed pair -- v_pk_mul_f32, v_pk_fma_f32 and one add per channel
    auto blend2 = [&](const Geo& q, const ds2f (&cp)[16], int k0, float (&B)[8]) {
#pragma unroll
      for (int k = k0; k < k0 + 2; ++k) {
        const ds2f t = __builtin_elementwise_fma(cp[2 * k + 1], q.w34, cp[2 * k] * q.w12);
        B[k] = t[0] + t[1];
      }
    };
    auto fixup = [&](const Geo& q, float (&B)[8]) {
      if (!pv || q.inwin) return;
      DcnTap tp;  // sample leaves the staged window: exact clamped global gathers
      const bool in = make_tap(q.h_im, q.w_im, a.H, a.W, tp);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float b = 0.f;
        if (in) {
          const float* pl = xg + (size_t)c * HW;
          const float v1 = tp.v1 ? pl[tp.o1] : 0.f, v2 = tp.v2 ? pl[tp.o2] : 0.f;
          const float v3 = tp.v3 ? pl[tp.o3] : 0.f, v4 = tp.v4 ? pl[tp.o4] : 0.f;
          b = (tp.w1 * v1 + tp.w2 * v2 + tp.w3 * v3 + tp.w4 * v4) * q.m;
        }
        B[c] = b;
      }
    };
    // the eight blended values of the lane's pixel -> three exact bf16 pieces per channel pair
    unsigned Hq[4], Mq[4], Lq[4];
    auto split_pair = [&](int p, const float (&bf)[8]) {
      const float v0 = bf[2 * p], v1 = bf[2 * p + 1];
      const unsigned h = dcs_cvt_pk(v0, v1);
      const float r0 = v0 - __builtin_bit_cast(float, h << 16), r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
      const unsigned m = dcs_cvt_pk(r0, r1);
      const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
      Hq[p] = h; Mq[p] = m; Lq[p] = dcs_cvt_pk(q0, q1);
    };
    // ... and the fragments of both pixel rows: one half exchange per register (lower lanes hold row 0, upper lanes row 1).
    // Bq[..][row][0] = B1 = (hi | mid), [1] = B2 = (hi | lo)
    dsu4 Bq[2][2][2];   // [tap parity][pixel row][B1 | B2]
    auto frags = [&](int par) {
      unsigned hc[4], h1[4], m1[4], c1[4], l1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) hc[i] = Hq[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(Hq[i], Mq[i], false, false);
        h1[i] = s0[0]; m1[i] = s0[1];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const auto t0 = __builtin_amdgcn_permlane32_swap(hc[i], Lq[i], false, false);
        c1[i] = t0[0]; l1[i] = t0[1];
      }
      Bq[par][0][0] = dsu4{h1[0], h1[1], h1[2], h1[3]};
      Bq[par][1][0] = dsu4{m1[0], m1[1], m1[2], m1[3]};
      Bq[par][0][1] = dsu4{c1[0], c1[1], c1[2], c1[3]};
      Bq[par][1][1] = dsu4{l1[0], l1[1], l1[2], l1[3]};
    };

    float om[2][3];              // [tap parity][dh, dw, mask]
    float Bf[8];                 // the blended fp32 values of the tap being prepared
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    {  // tap 0 of the chunk: nothing to hide behind
      Geo gq;
      ds2f c[16];
      int fix = 0;
      om_issue(I0{}, om[0]);
      om_issue(I1{}, om[1]);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(om[0][0]), "+v"(om[0][1]), "+v"(om[0][2]));
      landed_om(om[1]);
      geom_a(I0{}, om[0], gq);
      geom_b1(gq); geom_b2(gq, fix);
      corners_issue(gq.addr, c, 0); corners_issue(gq.addr, c, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
      DCS_PIN8(c, 8);
      blend2(gq, c, 0, Bf); blend2(gq, c, 2, Bf); blend2(gq, c, 4, Bf); blend2(gq, c, 6, Bf);
      if (fix) fixup(gq, Bf);
      split_pair(0, Bf); split_pair(1, Bf); split_pair(2, Bf); split_pair(3, Bf);
      frags(0);
    }
    if (kc < 2) DCS_STAMP(4 + 30 * kc);
#define DCS_SB __builtin_amdgcn_sched_barrier(0)
#define DCS_MF(j, mt, nt)                                                                                             \
  do {                                                                                                                \
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dsbf8, Aop[cur][mt][j]),                 \
                                                          __builtin_bit_cast(dsbf8, Bq[cur][nt][(j) == 1 ? 1 : 0]), acc[mt][nt], 0, 0, 0); \
    DCS_SB;                                                                                                           \
  } while (0)
    auto tap_body = [&](auto TAP) {
      constexpr int tap = decltype(TAP)::value;
      constexpr int cur = tap & 1, nxt = cur ^ 1;
      constexpr bool nx = tap + 1 < KK;
      using TN = std::integral_constant<int, (tap + 1 < KK ? tap + 1 : tap)>;
      using TNN = std::integral_constant<int, (tap + 2 < KK ? tap + 2 : tap)>;
      Geo gq;
      ds2f c[16];
      int fix = 0;
      DCS_SB;
      if (nx) a_load(nxt, kc, tap + 1);
      if (nx) geom_a(TN{}, om[nxt], gq);
      DCS_MF(0, 0, 0);
      if (nx) { geom_b1(gq); geom_b2(gq, fix); }
      DCS_MF(0, 0, 1);
      if (nx) corners_issue(gq.addr, c, 0);
      DCS_MF(0, 1, 0);
      if (nx) corners_issue(gq.addr, c, 1);
      DCS_MF(0, 1, 1);
      if (tap + 2 < KK) om_issue(TNN{}, om[cur]);  // (om[cur] held this tap's values: consumed one tap ago)
      DCS_MF(2, 0, 0);
      if (nx) { landed_a(c); blend2(gq, c, 0, Bf); blend2(gq, c, 2, Bf); }
      DCS_MF(2, 0, 1);
      if (nx) {
        landed_b(c);
        if (tap + 2 < KK) landed_om(om[cur]);
        blend2(gq, c, 4, Bf); blend2(gq, c, 6, Bf);
      }
      DCS_MF(2, 1, 0);
      if (nx) {
        if (fix) fixup(gq, Bf);
      }
      DCS_MF(2, 1, 1);
      if (nx) split_pair(0, Bf);
      DCS_MF(1, 0, 0);
      if (nx) split_pair(1, Bf);
      DCS_MF(1, 0, 1);
      if (nx) split_pair(2, Bf);
      DCS_MF(1, 1, 0);
      if (nx) split_pair(3, Bf);
      DCS_MF(1, 1, 1);
      if (nx) frags(nxt);
      if (kc < 2) DCS_STAMP(5 + 30 * kc + tap);
    };
    tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{});
    tap_body(std::integral_constant<int, 2>{}); tap_body(std::integral_constant<int, 3>{});
    tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
    tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{});
    tap_body(std::integral_constant<int, 8>{});
#undef DCS_MF
#undef DCS_SB
#undef DCS_PIN8
  }

  DCS_STAMP(62);
  const TileOut t{a.out, wset_ptr(a.bias, a.b_gs, n, a.wdiv), nullptr, a.act, 0, 0, a.Cout, a.H, a.W};
  store_mfma_tile<2, 2>(acc, t, n, cb * 64, oy0, 8, ox0, oy0 + 2 * wave, lo, hi);
  DCS_STAMP(63);
}

int mdcn_fwd_split_launch(const DcnK2& k, int grid, int mask_logit, hipStream_t st) {
  static PerDeviceOnce attr_once_t, attr_once_f;
  set_dyn_lds_once(attr_once_t, (const void*)mdcn_fwd_split_kernel<true>, DcnSplitShape::LDS_BYTES);
  set_dyn_lds_once(attr_once_f, (const void*)mdcn_fwd_split_kernel<false>, DcnSplitShape::LDS_BYTES);
  if (mask_logit) hipLaunchKernelGGL(mdcn_fwd_split_kernel<true>, dim3(grid), dim3(256), DcnSplitShape::LDS_BYTES, st, k);
  else hipLaunchKernelGGL(mdcn_fwd_split_kernel<false>, dim3(grid), dim3(256), DcnSplitShape::LDS_BYTES, st, k);
  return check_launch("mdcn_fwd_split_kernel");
}

}  // namespace dvsr
