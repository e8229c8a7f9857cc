// Weight / bias gradient of the dense convolutions (backward of conv2d.hip) on
// v_mfma_f32_32x32x2_f32:   dW[o][c][tap] = sum_{n,y,x} gy[n][o][y][x] * x[n][c][y*S+ty-pad][x*S+tx-pad]
//
// GEMM view per workgroup: D[o 64][c 64] (one D per tap) with K = pixels.  Each of the 4 waves owns
// one (o-half, c-half) 32x32 tile for ALL taps (9 accumulators = 144 VGPRs) and walks a strided
// list of 2x32-pixel tiles, so the pixel reduction stays in registers.  The per-workgroup sums are
// folded into NSLOT (<= 8) zero-initialised [tap][o][c] slots with coalesced hardware fp32 atomics
// (workgroup s -> slot s % NSLOT) and a second kernel sums the slots and transposes to OIHW; this
// keeps the flush traffic at NSLOT x 147 KB instead of nsplit x 147 KB (the r01 profile had the
// slot-less reduce at 51 us per call on the 44x80 inner-step shapes).  Summation order across
// workgroups is therefore not fixed (1e-7-level run-to-run differences, like cuDNN's wgrad).
// Operand reads are conflict-free: both LDS images use an odd plane stride.
//   A (lane l): gy[o = l&31][px = 2kk + (l>>5)]   <- s_g[o*65 + px]
//   B (lane l): x [c = l&31][px shifted by tap]   <- s_x[c*PLANEP + (py*S+ty)*IW + px*S+tx]
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "small_grid.h"

namespace dvsr {

#ifdef DVSR_CONV_TRACE
static long long* g_wgrad_trace = nullptr;
// every weight-gradient launch from now on stamps its timeline into buf (tools/wgrad_trace.py)
extern "C" int dvsr_debug_wgrad_trace(void* buf) { g_wgrad_trace = (long long*)buf; return 0; }
#endif

template <int KS, int S>
struct WgShape {
  static constexpr int TH = 2, TW = 32, NPX = TH * TW, KK = KS * KS;
  static constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
  static constexpr int PLANE = IH * IW, PLANEP = PLANE | 1, GROW = NPX + 1;
  static constexpr size_t LDS_BYTES = (size_t)(64 * GROW + 64 * PLANEP) * sizeof(float);
};

template <int KS, int S>
__global__ __launch_bounds__(256, 2) void conv2d_wgrad_kernel(WgradK a) {
  using Sh = WgShape<KS, S>;
  constexpr int KK = Sh::KK, IW = Sh::IW, PLANE = Sh::PLANE, PLANEP = Sh::PLANEP, GROW = Sh::GROW,
                NPX = Sh::NPX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_g = smem;
  float* s_x = smem + 64 * GROW;

  const int split = blockIdx.x, ob = blockIdx.y, cbk = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int ot = wave >> 1, ct = wave & 1;
  const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;

  f32x16 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float db = 0.f;

  const WgSpan sp = wg_span(a, split);
  for (int tile = sp.tile0; tile < sp.tile_end; tile += a.nsplit) {
    const int tx_ = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty_ = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int oy0 = ty_ * Sh::TH, ox0 = tx_ * Sh::TW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    __syncthreads();
    // ---- gy tile: s_g[o][p]
    for (int idx = tid; idx < 64 * NPX; idx += 256) {
      const int o = idx / NPX, p = idx - o * NPX;
      const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31), co = ob * 64 + o;
      float v = 0.f;
      if (co < a.Cout && oy < a.Ho && ox < a.Wo) {
        if (a.gy_ps)
          v = a.gy[(((size_t)n * (a.Cout >> 2) + (co >> 2)) * (2 * a.Ho) + 2 * oy + ((co >> 1) & 1)) *
                       (size_t)(2 * a.Wo) + 2 * ox + (co & 1)];
        else
          v = a.gy[((size_t)n * a.Cout + co) * HWo + (size_t)oy * a.Wo + ox];
      }
      s_g[o * GROW + p] = v;
    }
    // ---- x halo tile: s_x[c][iy][ix]
    const float* xn = a.x + (size_t)(n / a.x_bdiv) * a.x_bs;
    for (int idx = tid; idx < 64 * PLANE; idx += 256) {
      const int c = idx / PLANE, r = idx - c * PLANE;
      const int iy = r / IW, ix = r - iy * IW;
      const int gy_ = iy0 + iy, gx_ = ix0 + ix, ci = cbk * 64 + c;
      float v = 0.f;
      if (ci < a.Cin && (unsigned)gy_ < (unsigned)a.H && (unsigned)gx_ < (unsigned)a.W)
        v = xn[(size_t)ci * HW + (size_t)gy_ * a.W + gx_];
      s_x[c * PLANEP + r] = v;
    }
    __syncthreads();
    if (cbk == 0 && tid < 64) {
      float s = 0.f;
#pragma unroll 8
      for (int p = 0; p < NPX; ++p) s += s_g[tid * GROW + p];
      db += s;
    }
#pragma unroll 4
    for (int kk = 0; kk < NPX / 2; ++kk) {
      const int p = 2 * kk + hi;
      const int py = p >> 5, px = p & 31;
      const float av = s_g[(ot * 32 + lo) * GROW + p];
      const float* bx = s_x + (ct * 32 + lo) * PLANEP + (py * S) * IW + px * S;
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const int ty = t / KS, tx = t - ty * KS;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bx[ty * IW + tx], acc[t], 0, 0, 0);
      }
    }
  }

  // ---- partial[slot][tap][o][c]  (o, c padded to the 64-blocks of the grid), slot = split % nslot
  const int OP = a.nob * 64, CP = a.ncb * 64;
  const int slot = sp.slot;
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * 64 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int c = cbk * 64 + ct * 32 + lo;
      unsafeAtomicAdd(a.partial + (((size_t)slot * KK + t) * OP + o) * CP + c, acc[t][r]);
    }
  if (cbk == 0 && tid < 64) unsafeAtomicAdd(a.dbp + (size_t)slot * OP + ob * 64 + tid, db);
}

// -------------------------------------------------------------------------------------------------
// Pipelined variant for the stride-1 layers (3x3: all but two of EDVR's convolutions; 2x2 and 1x1: the
// estimator's re-expressed 4x4 convs, the TSA / fusion 1x1s and the DCN weight gradient).  Same
// decomposition and flush as above; what changes is how a tile gets into LDS:
//   * the tile of the NEXT iteration is fetched into registers before the MFMAs of the current one
//     (64 loads per lane, all in flight together) and written to the other LDS buffer after 3/4 of
//     them -- the simple kernel above pays a full memory latency per loop iteration of its staging
//     loops (measured: ~55 us per 64-pixel tile at 44x80, against 8.5 us of MFMA work);
//   * every wave stages whole channels (wave w: channels 16w..16w+15 of both operands), so a load's
//     address is a scalar channel-plane base plus a per-lane 32-bit offset that is fixed for the tile;
//   * the bias gradient falls out of the A operands the MFMA loop reads anyway (one v_add per k-step).
// 2 x 51.7 KB of LDS: one workgroup per CU, which is also what the pixel split produces.
// -------------------------------------------------------------------------------------------------
template <int KS, bool KYS = false>
__global__ __launch_bounds__(256, KYS ? 2 : 1) void conv2d_wgrad_pipe_kernel(WgradK a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // (7x7 / 9x9 kernel rows: 112 / 144 accumulators at two workgroups per CU leave no room for the fast path's operand ring)
  if constexpr (KS <= 3) {
    if (a.Cin - (int)blockIdx.z * 64 > 32 && a.Cout - (int)blockIdx.y * 64 > 32) {
      if (a.vx == 4) conv2d_wgrad_wide_item<KS, KYS, 4>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
      else if (a.vx == 2) conv2d_wgrad_wide_item<KS, KYS, 2>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
      else conv2d_wgrad_pipe_item<KS, KYS, true>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
      return;
    }
  }
  conv2d_wgrad_pipe_item<KS, KYS, false>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// dW[o][c_off + c][tap] = sum_s partial[s][tap][o][c];  db[o] = sum_s dbp[s][o]
// The slots are zeroed again while they are read, so the next wgrad launch on the same scratch can
// accumulate into them without a memset of its own (conv2d_wgrad_run's `scratch_is_zero` contract).
__global__ void wgrad_reduce_kernel(float* __restrict__ partial, float* __restrict__ dbp,
                                    float* __restrict__ dW, float* __restrict__ db, int nsplit, int KK,
                                    int OP, int CP, int Cout, int Cin, int Ctot, int c_off, long long dW_gs,
                                    long long db_gs) {
  // blockIdx.y = group (per-group gradients): its own nsplit slots, its own output
  partial += (size_t)blockIdx.y * nsplit * KK * OP * CP;
  dbp += (size_t)blockIdx.y * nsplit * OP;
  dW += (size_t)blockIdx.y * dW_gs;
  if (db) db += (size_t)blockIdx.y * db_gs;
  const int total = Cout * Cin * KK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % Cin;
    const int t2 = i / Cin;
    const int o = t2 % Cout;
    const int t = t2 / Cout;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
      float* q = partial + (((size_t)sp * KK + t) * OP + o) * CP + c;
      s += *q;
      *q = 0.f;
    }
    dW[((size_t)o * Ctot + c_off + c) * KK + t] = s;
  }
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < OP; o += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
      s += dbp[(size_t)sp * OP + o];
      dbp[(size_t)sp * OP + o] = 0.f;
    }
    if (db && o < Cout) db[o] = s;
  }
}

// Same reduction for a table of layers in ONE launch (blockIdx.y = layer): on the small inner-step clips a
// per-layer reduce is a 6 us kernel plus a launch gap behind every 44 us weight-gradient kernel.
__global__ void wgrad_reduce_batch_kernel(WgradReduceTable t) {
  const WgradReduceEntry& e = t.e[blockIdx.y];
  const int g = blockIdx.z;   // group of a per-group gradient (launch: z = the largest group count of the table)
  if (g >= e.ngroups) return;
  float* const partial = e.partial + (size_t)g * e.nslot * e.KK * e.OP * e.CP;
  float* const dbp = e.dbp + (size_t)g * e.nslot * e.OP;
  float* const dW = e.dW + (size_t)g * e.dW_gs;
  float* const db = e.db ? e.db + (size_t)g * e.db_gs : nullptr;
  const int total = e.Cout * e.Cin * e.KK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % e.Cin;
    const int t2 = i / e.Cin;
    const int o = t2 % e.Cout;
    const int tp = t2 / e.Cout;
    float s = 0.f;
    for (int sp = 0; sp < e.nslot; ++sp) {
      float* q = partial + (((size_t)sp * e.KK + tp) * e.OP + o) * e.CP + c;
      s += *q;
      *q = 0.f;
    }
    dW[((size_t)o * e.Ctot + e.c_off + c) * e.KK + tp] = s;
  }
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < e.OP; o += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int sp = 0; sp < e.nslot; ++sp) {
      s += dbp[(size_t)sp * e.OP + o];
      dbp[(size_t)sp * e.OP + o] = 0.f;
    }
    if (db && o < e.Cout) db[o] = s;
  }
}

int wgrad_reduce_batch(const WgradReduceEntry* entries, int n, hipStream_t st) {
  for (int base = 0; base < n; base += WGRAD_REDUCE_BATCH) {
    WgradReduceTable t;
    t.n = n - base < WGRAD_REDUCE_BATCH ? n - base : WGRAD_REDUCE_BATCH;
    int gmax = 1;
    for (int i = 0; i < t.n; ++i) { t.e[i] = entries[base + i]; gmax = t.e[i].ngroups > gmax ? t.e[i].ngroups : gmax; }
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(96, t.n, gmax), dim3(256), 0, st, t);
    int rc = check_launch("wgrad_reduce_batch_kernel");
    if (rc) return rc;
  }
  return DVSR_OK;
}

template <int KS, int S>
static void launch_wgrad(const WgradK& k, dim3 grid, hipStream_t st) {
  using Sh = WgShape<KS, S>;
  auto kern = conv2d_wgrad_kernel<KS, S>;
  static PerDeviceOnce attr_once;
  set_dyn_lds_once(attr_once, (const void*)kern, Sh::LDS_BYTES);
  const size_t lds = Sh::LDS_BYTES;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k);
}

static int wgrad_splits(int ntiles, int nob, int ncb, int KK, bool one_per_cu = false) {
  // one workgroup per CU when the pixel grid is small (latency-bound: every extra tile per workgroup
  // is serial time), two per CU for big grids
  (void)KK;
  static int force = -2;  // DVSR_WGRAD_SPLITS=<n> pins the number of pixel splits (A/B aid)
  if (force == -2) {
    const char* v = getenv("DVSR_WGRAD_SPLITS");
    force = v ? atoi(v) : -1;
  }
  int s = force > 0 ? force : ceil_div((ntiles >= 2048 && !one_per_cu) ? 512 : 256, nob * ncb);
  if (s > ntiles) s = ntiles;
  return s < 1 ? 1 : s;
}

// Pixel splits PER GROUP: the launch as a whole (groups x splits x cout blocks x cin blocks) aims at the same number of
// workgroups as an ungrouped one over the same tensor.
// (rounded DOWN: the un-split kernel runs one workgroup per CU, and 12 groups x ceil(256 / 12) = 264 workgroups are two
// rounds on 256 CUs -- the batched inner step measured 6.07 ms per frame at 12 frames against 5.38 at 8 and 5.09 at 16)
static int wgrad_group_splits(int gtiles, int nob, int ncb, int KK, int groups, bool one_per_cu = false) {
  if (groups <= 1) return wgrad_splits(gtiles, nob, ncb, KK, one_per_cu);
  int s = wgrad_splits(gtiles * groups, nob, ncb, KK, one_per_cu) / groups;
  if (s > gtiles) s = gtiles;
  return s < 1 ? 1 : s;
}

size_t conv2d_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int groups) {
  if (pad < 0) pad = ks / 2;
  if (groups < 1) groups = 1;
  const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
  const int gtiles = ceil_div(Wo, 32) * ceil_div(Ho, 2) * (N / groups);
  const int nob = ceil_div(Cout, 64), ncb = ceil_div(Cin, 64), KK = ks * ks;
  const int sp = wgrad_group_splits(gtiles, nob, ncb, KK, groups);
  const int ns = sp < 8 ? sp : 8;
  return (size_t)groups * ((size_t)ns * KK * nob * 64 * ncb * 64 + (size_t)ns * nob * 64) * sizeof(float);
}

int conv2d_wgrad_prepare(const float* x, long long x_bs, int x_bdiv, const float* gy, int gy_ps, float* dW, float* db,
                         int N, int Cin, int H, int W, int Cout, int Ctot, int c_off, int ks, int stride, void* ws,
                         size_t ws_bytes, hipStream_t st, int scratch_is_zero, int pad, WgradReduceEntry* defer,
                         WgradLaunch* out, int bf16, int groups, long long dW_gs, long long db_gs) {
  DVSR_REQUIRE(x && gy && dW && ws, DVSR_ERR_INVALID, "conv2d_wgrad: null pointer");
  DVSR_REQUIRE(((ks == 1 || ks == 2 || ks == 7 || ks == 9) && stride == 1) || (ks == 3 && (stride == 1 || stride == 2)),
               DVSR_ERR_UNSUPPORTED, "conv2d_wgrad: ks=%d stride=%d unsupported", ks, stride);
  if (groups < 1) groups = 1;
  DVSR_REQUIRE(N % groups == 0, DVSR_ERR_INVALID, "conv2d_wgrad: N=%d is not a multiple of groups=%d", N, groups);
  if (pad < 0) pad = ks / 2;
  const size_t need = conv2d_wgrad_workspace_bytes(N, Cin, H, W, Cout, ks, stride, pad, groups);
  DVSR_REQUIRE(ws_bytes >= need, DVSR_ERR_WORKSPACE, "conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
  WgradK& k = out->k;
  k.x = x; k.gy = gy; k.x_bs = x_bs > 0 ? x_bs : (long long)Cin * H * W; k.x_bdiv = x_bdiv > 0 ? x_bdiv : 1;
  k.N = N; k.Cin = Cin; k.H = H; k.W = W; k.Cout = Cout; k.pad = pad; k.gy_ps = gy_ps;
  k.Ho = (H + 2 * k.pad - ks) / stride + 1;
  k.Wo = (W + 2 * k.pad - ks) / stride + 1;
  k.tiles_x = ceil_div(k.Wo, 32); k.tiles_y = ceil_div(k.Ho, 2);
  k.ngroups = groups; k.gtiles = k.tiles_x * k.tiles_y * (N / groups); k.ntiles = k.gtiles * groups;
  k.nob = ceil_div(Cout, 64); k.ncb = ceil_div(Cin, 64);
  const int KK = ks * ks;
  // (all split counts below are PER GROUP; groups == 1 is the plain batch-summed gradient)
  k.nsplit = wgrad_group_splits(k.gtiles, k.nob, k.ncb, KK, groups, stride == 1);
  // (the estimator's 4x4 stride-2 convs, ks == 2 here: 128 / 256 / 512 / 1024 workgroups per launch measured 2.79 / 2.51 /
  // 2.54 / 2.56 ms per MFDN forward+backward -- the kernel is staging-bound, 16 accumulator tiles per staged tile against
  // 36 for 3x3, not parallelism-bound)
  k.nslot = k.nsplit < 8 ? k.nsplit : 8;
  out->ks = ks; out->stride = stride;
#ifdef DVSR_CONV_TRACE
  { static const int nf = getenv("DVSR_WGRAD_NOFLUSH") ? atoi(getenv("DVSR_WGRAD_NOFLUSH")) : 0; k.noflush = nf; }
  k.trace = g_wgrad_trace;
#endif
  // small pixel grids: one kernel row per workgroup (conv2d_wgrad_pipe_kernel<3, true>); the pixel split is then
  // sized for ~two workgroups per CU over the three rows.  DVSR_WGRAD_KYS_BELOW=<tiles x cout blocks x cin blocks>
  // moves the threshold (0 disables).  1024 (r03; 4096 before): the row split pays when a workgroup would otherwise
  // get less than ~4 tiles; above that its three-fold re-staging of x costs more than the parallelism gives -- the
  // batched inner step (8 frames, 40 x 44x80 = 2640 tiles per layer) measured 5.96 ms per frame with the row split on
  // those layers and 5.45 without (profiles/r03_inner_batch_sweeps.txt).
  static int kys_below = -1;
  if (kys_below < 0) {
    const char* v = getenv("DVSR_WGRAD_KYS_BELOW");
    kys_below = v ? atoi(v) : 1024;
  }
  auto per_group = [&](int s) {   // a launch-wide split count -> per group (rounded down), never more than a group has tiles
    s = groups > 1 ? s / groups : s;
    return s > k.gtiles ? k.gtiles : (s < 1 ? 1 : s);
  };
  out->bf = (bf16 && stride == 1 && (ks == 3 || (ks == 2 && bf16 == 2))) ? bf16 : 0;   // (the split kernel also has a 2x2 form)
  if (out->bf) {
    // the bf16 kernel is staging- and flush-bound (36 MFMAs per tile): fewer workgroups, each with more tiles, keep the
    // 147 KB-per-workgroup atomic flush small.  DVSR_WGRAD_BF_WGS=<workgroups per launch to aim for>.
    static int bf_wgs = -1;
    if (bf_wgs < 0) {
      const char* v = getenv("DVSR_WGRAD_BF_WGS");
      bf_wgs = v ? atoi(v) : 128;
    }
    // (128 at the 64x64 tiles of configs[4] -- sweep in profiles/r02_z_bf16_wgrad.txt -- growing with the pixel grid:
    // about eight tiles per workgroup, at most 512 workgroups)
    int target = k.ntiles / 8;
    target = target < bf_wgs ? bf_wgs : (target > 512 ? 512 : target);
    const int s2 = per_group(ceil_div(target, k.nob * k.ncb));
    if (s2 < k.nsplit) k.nsplit = s2;
  }
  out->kys = !out->bf && ((ks == 3 && stride == 1 && (long long)k.ntiles * k.nob * k.ncb < kys_below) || ks == 7 || ks == 9);
  if (out->kys) {
    static int kys_wgs = -1;   // DVSR_WGRAD_KYS_WGS=<workgroups per launch to aim for>
    if (kys_wgs < 0) {
      const char* v = getenv("DVSR_WGRAD_KYS_WGS");
      kys_wgs = v ? atoi(v) : 288;
    }
    k.nsplit = per_group(ceil_div(ks == 3 ? kys_wgs : 512, ks * k.nob * k.ncb));
  }
  // wide staging: gy rows as float4 (Wo % 4 == 0, not pixel-shuffled), x as float4 / float2 when the row pitch, the batch
  // stride and the pointers allow it (DVSR_WGRAD_WIDE=0 keeps the scalar loads: A/B aid)
  k.vx = 0;
  {
    static const bool wide = [] { const char* v = getenv("DVSR_WGRAD_WIDE"); return !(v && v[0] == '0'); }();
    const bool gy_ok = !gy_ps && k.Wo % 4 == 0 && ((uintptr_t)gy & 15) == 0;
    if (wide && gy_ok && stride == 1 && ks <= 3 && out->bf != 1) {   // (bf == 2: the split kernel has the same two vector forms)
      if (W % 4 == 0 && k.x_bs % 4 == 0 && ((uintptr_t)x & 15) == 0) k.vx = 4;
      else if (W % 2 == 0 && k.x_bs % 2 == 0 && ((uintptr_t)x & 7) == 0) k.vx = 2;
    }
  }
  if (out->bf == 2 && k.vx != 0) {
    // the split kernel's vector-staging form can run one kernel ROW per workgroup, two workgroups per CU (conv2d_wgrad_bf16.hip:
    // conv2d_wgrad_split3v_kernel<.., KYS>).  It stages gy three times and x one and a half times, so it pays where a
    // workgroup has few tiles and the fixed costs (first loads, flush, the tail of the last round) weigh most -- measured
    // (tools/wgrad_bench.py, us with / without): 40 x 44x80 93 / 102, 8 x 44x80 40 / 51, 40 x 22x40 47 / 61, 1 x 176x320 52 / 63,
    // but 40 x 176x320 1046 / 941, 64 -> 216 at 40 x 44x80 290 / 264.  DVSR_WGRAD_S3_KYS_BELOW=<tiles x cout blocks x cin
    // blocks> moves the threshold (0 = never); DVSR_WGRAD_S3V=0 keeps the round-4 kernel.  Switches are read once per process.
    static const int s3_below = [] {
      const char* w = getenv("DVSR_WGRAD_S3V");
      if (w && w[0] == '0') return 0;
      const char* v = getenv("DVSR_WGRAD_S3_KYS_BELOW");
      return v ? atoi(v) : 4000;
    }();
    const bool fits = (unsigned long long)Cin * H * W < (1ull << 30) && (unsigned long long)Cout * k.Ho * k.Wo < (1ull << 30);
    const long long work = (long long)k.ntiles * k.nob * k.ncb;
    // (only where conv2d_wgrad_split3_launch has a vector-staging form: 3x3 with pad 0 / 1, the 2x2 form with pad 0 -- any
    // other pad runs the scalar-staging kernel, which has no row split)
    const bool vec_form = (ks == 3 && (k.pad == 0 || k.pad == 1)) || (ks == 2 && k.pad == 0);
    if (work < s3_below && fits && vec_form) {
      // two workgroups per CU from ~1000 tiles, one below; rounded DOWN (one workgroup more than the CUs hold at once is a
      // second round with one workgroup in it)
      static const int s3_wgs = [] { const char* v = getenv("DVSR_WGRAD_S3_WGS"); return v ? atoi(v) : 0; }();
      const int wgs = s3_wgs > 0 ? s3_wgs : (work >= 1000 ? 512 : 256);
      out->kys = 1;
      k.nsplit = per_group(wgs / (ks * k.nob * k.ncb));
    }
  }
  if (k.nslot > k.nsplit) k.nslot = k.nsplit;   // (the slot region was sized for the un-split launch: never larger)
  out->grid = dim3((out->kys ? ks : 1) * groups * k.nsplit, k.nob, k.ncb);
  // slot regions: [group][slot][tap][o][c] partial sums, then [group][slot][o] bias sums
  const size_t pfloats = (size_t)groups * k.nslot * KK * k.nob * 64 * k.ncb * 64;
  k.partial = (float*)ws;
  k.dbp = k.partial + pfloats;
  if (!scratch_is_zero) {
    const size_t zbytes = (pfloats + (size_t)groups * k.nslot * k.nob * 64) * sizeof(float);
    DVSR_REQUIRE(hipMemsetAsync(ws, 0, zbytes, st) == hipSuccess, DVSR_ERR_HIP, "conv2d_wgrad: memset failed");
  }
  if (defer) {  // the caller reduces a batch of layers later (wgrad_reduce_batch); `ws` must stay untouched until then
    *defer = WgradReduceEntry{k.partial, k.dbp, dW, db, k.nslot, KK, k.nob * 64, k.ncb * 64, Cout, Cin, Ctot, c_off};
    defer->ngroups = groups; defer->dW_gs = dW_gs; defer->db_gs = db_gs;
  }
  return DVSR_OK;
}

int conv2d_wgrad_launch(const WgradLaunch& l, hipStream_t st) {
  if (l.bf == 2) return conv2d_wgrad_split3_launch(l, st);
  if (l.bf) return conv2d_wgrad_bf16_launch(l, st);
  const WgradK& k = l.k;
  const dim3 grid = l.grid;
  const int ks = l.ks, stride = l.stride;
  static int use_simple = -1;  // DVSR_WGRAD_SIMPLE=1: the non-pipelined kernel for every shape (A/B aid)
  if (use_simple < 0) {
    const char* v = getenv("DVSR_WGRAD_SIMPLE");
    use_simple = (v && v[0] == '1') ? 1 : 0;
  }
  if (stride == 1 && !use_simple) {
    auto launch_pipe = [&](auto ks_tag, auto kys_tag) {
      constexpr int KS_ = decltype(ks_tag)::value;
      constexpr bool KYS_ = decltype(kys_tag)::value;
      constexpr size_t lds_a = WgPipeShape<KS_, KYS_>::LDS_BYTES;
      constexpr size_t lds_b = KS_ <= 3 ? (WgWideShape<KS_, KYS_, 4>::LDS_BYTES > WgWideShape<KS_, KYS_, 2>::LDS_BYTES
                                               ? WgWideShape<KS_, KYS_, 4>::LDS_BYTES : WgWideShape<KS_, KYS_, 2>::LDS_BYTES) : 0;
      constexpr size_t lds = lds_a > lds_b ? lds_a : lds_b;
      static PerDeviceOnce attr_once;
      set_dyn_lds_once(attr_once, (const void*)conv2d_wgrad_pipe_kernel<KS_, KYS_>, lds);
      hipLaunchKernelGGL((conv2d_wgrad_pipe_kernel<KS_, KYS_>), grid, dim3(256), lds, st, k);
    };
    if (ks == 7) launch_pipe(std::integral_constant<int, 7>{}, std::true_type{});
    else if (ks == 9) launch_pipe(std::integral_constant<int, 9>{}, std::true_type{});
    else if (ks == 3 && l.kys) launch_pipe(std::integral_constant<int, 3>{}, std::true_type{});
    else if (ks == 3) launch_pipe(std::integral_constant<int, 3>{}, std::false_type{});
    else if (ks == 2) launch_pipe(std::integral_constant<int, 2>{}, std::false_type{});
    else launch_pipe(std::integral_constant<int, 1>{}, std::false_type{});
  } else if (ks == 3 && stride == 1) launch_wgrad<3, 1>(k, grid, st);
  else if (ks == 3) launch_wgrad<3, 2>(k, grid, st);
  else if (ks == 2) launch_wgrad<2, 1>(k, grid, st);
  else launch_wgrad<1, 1>(k, grid, st);
  return check_launch("conv2d_wgrad_kernel");
}

// x: one input of the conv ([N/x_bdiv][Cin][H][W], batch stride x_bs or dense), gy: gradient of the
// conv's pre-activation output.  Writes dW[:, c_off:c_off+Cin, :, :] of a [Cout][Ctot][ks][ks]
// gradient (and db when non-null).
int conv2d_wgrad_run(const float* x, long long x_bs, int x_bdiv, const float* gy, int gy_ps, float* dW,
                     float* db, int N, int Cin, int H, int W, int Cout, int Ctot, int c_off, int ks,
                     int stride, void* ws, size_t ws_bytes, hipStream_t st, int scratch_is_zero, int pad,
                     WgradReduceEntry* defer, int groups, long long dW_gs, long long db_gs) {
  WgradLaunch l;
  int rc = conv2d_wgrad_prepare(x, x_bs, x_bdiv, gy, gy_ps, dW, db, N, Cin, H, W, Cout, Ctot, c_off, ks, stride, ws,
                                ws_bytes, st, scratch_is_zero, pad, defer, &l, 0, groups, dW_gs, db_gs);
  if (rc) return rc;
  rc = conv2d_wgrad_launch(l, st);
  if (rc || defer) return rc;
  const WgradK& k = l.k;
  const int KK = ks * ks, total = Cout * Cin * KK;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 256), k.ngroups), dim3(256), 0, st, k.partial, k.dbp,
                     dW, db, k.nslot, KK, k.nob * 64, k.ncb * 64, Cout, Cin, Ctot, c_off, dW_gs, db_gs);
  return check_launch("wgrad_reduce_kernel");
}

}  // namespace dvsr

extern "C" size_t dvsr_conv2d_backward_workspace_bytes(const dvsr_conv2d_desc* d) {
  if (!d) return 0;
  size_t a = dvsr::conv2d_wgrad_workspace_bytes(d->N, d->c0 > d->c1 ? d->c0 : d->c1, d->H, d->W, d->Cout,
                                                d->ks, d->stride);
  return a;
}

// Backward of dvsr_conv2d_forward for the plain layout (no pixel shuffle): `gy` is the gradient
// w.r.t. the PRE-activation output (the caller multiplies by act' first).  Any of gx0/gx1/gw/gb
// may be NULL to skip it.  gx1 has the shape of x1 only when x1_bdiv == 1.
extern "C" int dvsr_conv2d_backward(const dvsr_conv2d_desc* d, const float* gy, float* gx0, float* gx1,
                                    float* gw, float* gb, void* workspace, size_t workspace_bytes,
                                    dvsr_stream_t stream) {
  using namespace dvsr;
  DVSR_REQUIRE(d && gy, DVSR_ERR_INVALID, "conv2d_backward: null argument");
  DVSR_REQUIRE(d->pixel_shuffle == 0 && d->x1_bdiv <= 1, DVSR_ERR_UNSUPPORTED,
               "conv2d_backward: pixel_shuffle / broadcast x1 are handled by the EDVR engine only");
  hipStream_t st = (hipStream_t)stream;
  const int pad = d->ks / 2, ctot = d->c0 + d->c1;
  const int Ho = (d->H + 2 * pad - d->ks) / d->stride + 1, Wo = (d->W + 2 * pad - d->ks) / d->stride + 1;
  int rc;
  if (gw) {
    rc = conv2d_wgrad_run(d->x0, d->x0_bstride, 1, gy, 0, gw, gb, d->N, d->c0, d->H, d->W, d->Cout, ctot, 0,
                          d->ks, d->stride, workspace, workspace_bytes, st, 0);
    if (rc) return rc;
    if (d->c1) {
      rc = conv2d_wgrad_run(d->x1, d->x1_bstride, 1, gy, 0, gw, nullptr, d->N, d->c1, d->H, d->W, d->Cout,
                            ctot, d->c0, d->ks, d->stride, workspace, workspace_bytes, st, 0);
      if (rc) return rc;
    }
  }
  for (int which = 0; which < 2; ++which) {
    float* gx = which ? gx1 : gx0;
    const int ci = which ? d->c1 : d->c0;
    if (!gx || !ci) continue;
    dvsr_conv2d_desc g = {};
    g.x0 = gy; g.w = d->w; g.y = gx; g.N = d->N; g.c0 = d->Cout; g.Cout = ci; g.ks = d->ks; g.stride = 1;
    g.pad = pad; g.act = ACT_NONE; g.x1_bdiv = 1;
    ConvExtra ex;
    ex.wt = 1; ex.w_ctot = ctot; ex.w_coff = which ? d->c0 : 0;
    if (d->stride == 2) { ex.in_dil = 2; ex.Hs = Ho; ex.Ws = Wo; g.H = d->H; g.W = d->W; }
    else { g.H = Ho; g.W = Wo; }
    rc = conv2d_run(g, ex, st);
    if (rc) return rc;
  }
  return DVSR_OK;
}

// Weight / bias gradient of a single-input 3x3 stride-1 conv with bf16 operands on the bf16 MFMA (fp32 accumulate):
// the op-level face of conv2d_wgrad_bf16.hip (the EDVR plan uses it when network_G.bf16_mfma = 1).  Workspace:
// dvsr_conv2d_backward_workspace_bytes.
static int wgrad_bf_mode(const dvsr_conv2d_desc* d, const float* gy, float* gw, float* gb, void* workspace,
                         size_t workspace_bytes, dvsr_stream_t stream, int mode) {
  using namespace dvsr;
  DVSR_REQUIRE(d && gy && gw && d->x0, DVSR_ERR_INVALID, "conv2d_wgrad_bf16: null argument");
  DVSR_REQUIRE(d->ks == 3 && d->stride == 1 && d->c1 == 0 && d->pixel_shuffle == 0, DVSR_ERR_UNSUPPORTED,
               "conv2d_wgrad_bf16: 3x3 stride-1 single-input convolutions only");
  hipStream_t st = (hipStream_t)stream;
  WgradLaunch l;
  int rc = conv2d_wgrad_prepare(d->x0, d->x0_bstride, 1, gy, 0, gw, gb, d->N, d->c0, d->H, d->W, d->Cout, d->c0, 0, 3, 1,
                                workspace, workspace_bytes, st, 0, d->pad, nullptr, &l, mode);
  if (rc) return rc;
  rc = conv2d_wgrad_launch(l, st);
  if (rc) return rc;
  const WgradK& k = l.k;
  const int total = d->Cout * d->c0 * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, k.partial, k.dbp, gw, gb, k.nslot, 9,
                     k.nob * 64, k.ncb * 64, d->Cout, d->c0, d->c0, 0, 0LL, 0LL);
  return check_launch("wgrad_reduce_kernel");
}

extern "C" int dvsr_conv2d_wgrad_bf16(const dvsr_conv2d_desc* d, const float* gy, float* gw, float* gb, void* workspace,
                                      size_t workspace_bytes, dvsr_stream_t stream) {
  return wgrad_bf_mode(d, gy, gw, gb, workspace, workspace_bytes, stream, 1);
}

// The same on the exact 3-way bf16 split of both operands (fp32 accuracy; what the plans run for their 3x3 stride-1 weight
// gradients unless DVSR_WGRAD_SPLIT3=0).
extern "C" int dvsr_conv2d_wgrad_split3(const dvsr_conv2d_desc* d, const float* gy, float* gw, float* gb, void* workspace,
                                        size_t workspace_bytes, dvsr_stream_t stream) {
  return wgrad_bf_mode(d, gy, gw, gb, workspace, workspace_bytes, stream, 2);
}

