// Winograd F(2x2, 3x3) with the sixteen transformed-domain GEMMs on the bf16 matrix pipe at fp32 accuracy -- the PACKED WEIGHT
// IMAGE of that kernel family and its launcher.  Every fp32 operand (transformed weight U, transformed input V) is split EXACTLY
// into three bf16 pieces, x = hi + mid + lo (8 + 8 + 8 significand bits), and the six partial products above 2^-24,
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid,
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (EDVR_arch.py:254-313 is what is being computed).
//
// History.  Round 4 built four forms of the kernel in this file (DVSR_CONV_WINO3_BLK = 0..3: V through the LDS; DESIGN 3.1f keeps
// their description and measurements, profiles/r04_wino3_variants.txt the numbers); round 5's form 4 (conv2d_wino4.hip: the B
// operand built in registers) replaced them as the default and round 6 RETIRED them -- two rounds without a layer on which any
// of them won.  What remains here is what form 4 reads: the weight pack.
//   * U: pack_weights_wino3_kernel computes G g G^T in fp32 and stores the three pieces as the exact fragment image of a phase
//     (a chunk of 8 input channels = two phases p = the transformed-patch rows xi in {2p, 2p+1}),
//     [cout block][chunk][p][piece][xl][cout 64][8 channels] bf16 = 16-byte records, 24 KB per phase.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

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

// th = 4: 4 x 64-pixel workgroup tiles (TC = 32), th = 8: 8 x 32 (TC = 16), th = 16: 16 x 16 (TC = 8)
int conv2d_wino3_launch(const ConvK2& k, int th, hipStream_t st) { return conv2d_wino4_launch(k, th, st); }

}  // namespace dvsr
