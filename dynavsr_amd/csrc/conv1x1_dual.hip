// Two 1x1 convolutions over the SAME input in one pass (TSA fusion: fea_fusion and sAtt_1 both read the gated 5 x 64-channel
// tensor, EDVR_arch.py:183-202): y0 = act(W0 x + b0), y1 = act(W1 x + b1), 64 output channels each.
//
// As two launches of the generic kernel the pair took 2 x 45 us at 1 x 320 x 180 x 320 -- 57600 pixels are 225 workgroups
// with one 32-channel chunk in flight each: a chain of ten dependent memory round trips -- for 2 x 2.4 GFLOP (15 us of fp32
// MFMA time each) and 2 x 74 MB of reads.  Here:
//   * a workgroup owns 128 CONSECUTIVE pixels of a sample (a 1x1 conv has no spatial structure: the plane is a flat array)
//     and all 128 output channels: wave w computes the 32 channels (w & 1) * 32 .. of convolution w >> 1 for the 128 pixels
//     (four 32x32 accumulator tiles); 450 workgroups at 1 x 180 x 320, two per CU;
//   * x is read ONCE, by LDS-DMA into a three-stage ring of 16-channel chunks ([channel][pixel], 8 KB a stage, + 8 KB of weights): two chunks
//     are in flight while one is consumed;
//   * the MFMA's pixel index is permuted so that pixel block j holds the pixels {4 n + j}: lane n's B operands of the four
//     blocks are ONE ds_read_b128 (x[c][4 n .. 4 n + 3]) and its sixteen results per output row are four consecutive pixels
//     -- 16-byte stores;
//   * the K index is permuted likewise: step i contracts channels (c0 + i, c0 + 8 + i), so lane half k needs the eight
//     consecutive weights W[o][c0 + 8 k ..]: the chunk's 128 x 16 weights ride in the same ring, DMAed straight from the
//     [Cout][Cin] parameters (no pack) in a channel-group-major order that makes the operand reads conflict-free.
// fp32 throughout (v_mfma_f32_32x32x2_f32: an fmaf chain per output, channel order 0, 8, 1, 9, ... within a chunk).
#include "common.h"
#include "kernels.h"

namespace dvsr {

struct Dual1x1K {
  const float* x; const float* w0; const float* b0; const float* w1; const float* b1; float* y0; float* y1;
  int N, Cin, HW, act, tiles_per_img;
  int wdiv; long long w_gs; int b_gs;   // per-sample weight sets (common.h: wset_ptr)
};

constexpr int D1_PX = 128, D1_CH = 16, D1_NST = 3;

__global__ __launch_bounds__(256, 2) void conv1x1_dual_kernel(Dual1x1K a) {
  // ring stage = [16 channels][128 pixels] of x, then the chunk's weights as [4 channel groups][128 couts][4 channels]
  __shared__ __attribute__((aligned(16))) float s_r[D1_NST][2 * D1_CH * D1_PX];
  const int tile = blockIdx.x;
  const int n = tile / a.tiles_per_img, p0 = (tile - n * a.tiles_per_img) * D1_PX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 31, k = lane >> 5;
  const int conv = wave >> 1, cw = (wave & 1) * 32;
  const float* bsel = wset_ptr(conv ? a.b1 : a.b0, a.b_gs, n, a.wdiv);
  const float* xn = a.x + (size_t)n * a.Cin * a.HW;
  const int nchunks = a.Cin / D1_CH;

  // DMA maps (one instruction = 64 lanes x 16 bytes, written to the LDS in lane order).
  // x: slot q = 64 (wave + 4 j) + lane = (channel q / 32 of the chunk, pixels 4 (q % 32) ..); pixels past the plane re-read its
  //    last group (they only feed outputs that are never stored).
  // w: slot q = (channel group q / 128, cout q % 128): four consecutive input channels of one output row, straight from the
  //    [64][Cin] parameters (couts 64 .. 127 = the second convolution) -- group-major, so that the 32 lanes of an operand read
  //    touch consecutive 16-byte slots.
  const char* xsrc[2];
  const char* wsrc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = 64 * (wave + 4 * j) + lane;
    int px = p0 + 4 * (q & 31);
    px = px < a.HW ? px : a.HW - 4;
    xsrc[j] = reinterpret_cast<const char*>(xn + (size_t)(q >> 5) * a.HW + px);
    const int co = q & 127, g = q >> 7;
    const float* wb = wset_ptr(co < 64 ? a.w0 : a.w1, a.w_gs, n, a.wdiv);
    wsrc[j] = reinterpret_cast<const char*>(wb + (size_t)(co & 63) * a.Cin + 4 * g);
  }
  auto dma = [&](int kc, int stage) __attribute__((always_inline)) {
    const size_t xo = (size_t)kc * D1_CH * a.HW * 4, wo = (size_t)kc * D1_CH * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[j] + xo),
                                       (__attribute__((address_space(3))) void*)(&s_r[stage][256 * (wave + 4 * j)]), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + wo),
                                       (__attribute__((address_space(3))) void*)(&s_r[stage][D1_CH * D1_PX + 256 * (wave + 4 * j)]), 16, 0, 0);
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // Every VMEM instruction of the loop is an LDS-DMA, four per chunk, and the body is branch-free (past the last chunk it
  // re-fetches the last one into the stage chunk kc - 1 just left): the queue position of each is static and the waits are
  // explicit.  (With compiler-tracked loads beside the DMAs the waitcnt pass put a vmcnt(0) behind the barrier, i.e. waited
  // for the chunk that had just been put in flight; __syncthreads does the same.)  The LDS reads are inline asm for the same
  // reason: the compiler cannot tell which stage a DMA in flight writes.
  const int last = nchunks - 1;
  dma(0, 0);
  dma(last < 1 ? last : 1, 1);
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) const float*)&s_r[0][0]);
  const unsigned xrd = (unsigned)(((8 * k) * D1_PX + 4 * nn) * 4);                                   // x[8 k + i][4 nn ..]
  const unsigned wrd = (unsigned)((D1_CH * D1_PX + (2 * k) * 512 + (conv * 64 + cw + nn) * 4) * 4);   // w[cout][8 k .. 8 k + 7]
  for (int kc = 0; kc < nchunks; ++kc) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // newest first: [chunk kc + 1: 4 DMAs] [chunk kc: 4 DMAs] -- chunk kc has landed
    __builtin_amdgcn_s_barrier();   // ... for every wave, and every wave is done with chunk kc - 1 (its stage is refilled below)
    __builtin_amdgcn_sched_barrier(0);
    dma(kc + 2 < last ? kc + 2 : last, (kc + 2) % D1_NST);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned sb = lds0 + (unsigned)((kc % D1_NST) * (2 * D1_CH * D1_PX * 4));
    f32x4 b[8], wv[2];
    asm volatile(
        "ds_read_b128 %8, %11\n\tds_read_b128 %9, %11 offset:2048\n\t"
        "ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:512\n\tds_read_b128 %2, %10 offset:1024\n\tds_read_b128 %3, %10 offset:1536\n\t"
        "ds_read_b128 %4, %10 offset:2048\n\tds_read_b128 %5, %10 offset:2560\n\tds_read_b128 %6, %10 offset:3072\n\tds_read_b128 %7, %10 offset:3584\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7]), "=&v"(wv[0]), "=&v"(wv[1])
        : "v"(sb + xrd), "v"(sb + wrd));
    static_assert(D1_PX * 4 == 512 && 128 * 16 == 2048, "the read offsets hard-code the pitches");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float av = wv[i >> 2][i & 3];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[i][j], acc[j], 0, 0, 0);
    }
    // (the LDS reads of this chunk have returned -- the asm waits for them -- before any wave passes the next barrier)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still be writing this workgroup's LDS when it ends

  const int px = p0 + 4 * nn;
  if (px >= a.HW) return;
  float* yb = (conv ? a.y1 : a.y0) + (size_t)n * 64 * a.HW + px;
  // (the sixteen bias values first: read inside the store loop each load waited behind the previous store, which the
  // compiler may not reorder it with -- sixteen dependent round trips per thread)
  float bvs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bvs[r] = bsel ? bsel[cw + (r & 3) + 8 * (r >> 2) + 4 * k] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = cw + (r & 3) + 8 * (r >> 2) + 4 * k;
    const float bv = bvs[r];
    f32x4 v = {apply_act(acc[0][r] + bv, a.act), apply_act(acc[1][r] + bv, a.act), apply_act(acc[2][r] + bv, a.act),
               apply_act(acc[3][r] + bv, a.act)};
    *reinterpret_cast<f32x4*>(yb + (size_t)co * a.HW) = v;
  }
}

// x [N][Cin][HW] dense, w0 / w1 [64][Cin], y0 / y1 [N][64][HW]; needs Cin % 16 == 0, HW % 4 == 0, HW >= 4, 16-byte aligned
// tensors (conv1x1_dual_ok).  wdiv / w_gs / b_gs: per-sample weight sets as in the conv kernels.
bool conv1x1_dual_ok(const float* x, const float* w0, const float* w1, const float* y0, const float* y1, int Cin, long long HW) {
  return Cin % 16 == 0 && HW % 4 == 0 && HW >= 4 && HW < (1ll << 28) &&
         ((((uintptr_t)x | (uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)y0 | (uintptr_t)y1) & 15) == 0);
}

int conv1x1_dual_run(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, float* y0, float* y1,
                     int N, int Cin, int HW, int act, hipStream_t st, int wdiv, long long w_gs, int b_gs) {
  DVSR_REQUIRE(x && w0 && w1 && y0 && y1, DVSR_ERR_INVALID, "conv1x1_dual: null pointer");
  DVSR_REQUIRE(conv1x1_dual_ok(x, w0, w1, y0, y1, Cin, HW), DVSR_ERR_UNSUPPORTED, "conv1x1_dual: Cin=%d HW=%d / alignment", Cin, HW);
  DVSR_REQUIRE(w_gs % 4 == 0, DVSR_ERR_UNSUPPORTED, "conv1x1_dual: weight-set stride %lld", w_gs);
  Dual1x1K k;
  k.x = x; k.w0 = w0; k.b0 = b0; k.w1 = w1; k.b1 = b1; k.y0 = y0; k.y1 = y1;
  k.N = N; k.Cin = Cin; k.HW = HW; k.act = act; k.tiles_per_img = ceil_div(HW, D1_PX);
  k.wdiv = wdiv > 0 ? wdiv : 1; k.w_gs = w_gs; k.b_gs = b_gs;
  hipLaunchKernelGGL(conv1x1_dual_kernel, dim3(N * k.tiles_per_img), dim3(256), 0, st, k);
  return check_launch("conv1x1_dual_kernel");
}

}  // namespace dvsr

extern "C" int dvsr_conv1x1_dual(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, float* y0,
                                 float* y1, int N, int Cin, int H, int W, int act, dvsr_stream_t stream) {
  DVSR_REQUIRE(N > 0 && H > 0 && W > 0, DVSR_ERR_INVALID, "conv1x1_dual: empty tensor");
  return dvsr::conv1x1_dual_run(x, w0, b0, w1, b1, y0, y1, N, Cin, H * W, act, (hipStream_t)stream);
}
