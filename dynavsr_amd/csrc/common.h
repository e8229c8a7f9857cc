// Shared host/device helpers for libdynavsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <type_traits>
#include <utility>

namespace dvsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Error codes returned by every C-ABI entry point (0 = success).
enum : int {
  DVSR_OK = 0,
  DVSR_ERR_INVALID = -1,      // bad shapes / null pointers / contract violation
  DVSR_ERR_UNSUPPORTED = -2,  // valid for the reference op, not implemented by these kernels
  DVSR_ERR_WORKSPACE = -3,    // caller-provided workspace too small
  DVSR_ERR_HIP = -4,          // a HIP runtime call or kernel launch failed
};

void set_error(const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> DVSR_ERR_HIP + message

#define DVSR_REQUIRE(cond, code, ...)   \
  do {                                  \
    if (!(cond)) {                      \
      ::dvsr::set_error(__VA_ARGS__);   \
      return (code);                    \
    }                                   \
  } while (0)

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// f(integral_constant<int, I>) for I in [B, E): a loop whose index is a compile-time constant in the body (operand
// addresses become immediates, register arrays stay in registers).
template <int B, int... I, class F>
__host__ __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (E > B) static_for_impl<B>(std::make_integer_sequence<int, E - B>{}, static_cast<F&&>(f));
}

// Per-sample weight sets: batch item n takes the weights / bias at p + (n / wdiv) * gs (gs == 0: one set for the batch).
__device__ __forceinline__ const float* wset_ptr(const float* p, long long gs, int n, int wdiv) {
  return (gs && p) ? p + (size_t)(n / wdiv) * gs : p;
}

// Activation codes shared by conv / dcn epilogues.
enum : int { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2 };

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_LRELU) return v > 0.f ? v : 0.1f * v;  // LeakyReLU(0.1), EDVR_arch.py:93
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// d(act)/d(pre-activation) expressed through the POST-activation value y (sign is preserved by
// both activations, so the saved output is enough).
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  if (act == ACT_LRELU) return y > 0.f ? 1.f : 0.1f;
  if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}

// ---- output-tile writer shared by the MFMA kernels (conv2d*, mdcn forward) -----------------------
// D layout of v_mfma_f32_32x32x2_f32: register r of lane (lo, hi) holds row (r&3) + 8*(r>>2) + 4*hi,
// column lo.  Rows are output channels, columns are pixels of one image row.
// gfx9 counts stores in vmcnt too: a predicated store per basic block makes the compiler put
// s_waitcnt vmcnt(0) in front of every one of them, i.e. each store waits for the acknowledge of the
// previous one.  Full tiles therefore take a straight-line path (all loads first, one wait, 16
// back-to-back stores per accumulator); only edge tiles use the predicated path.
struct TileOut {
  float* y;            // NCHW output (or its 2x pixel-shuffled form when ps)
  const float* bias;   // [Cout] or null
  const float* res;    // residual added after the activation, same layout as y, or null
  int act, ps, accum;  // accum: y += result
  int Cout, Ho, Wo;
  // data-gradient launches: multiply the result by act'(gmask) -- the saved OUTPUT of the layer whose
  // gradient this is (same layout as y) -- i.e. the activation backward of the producer, fused
  const float* gmask = nullptr;
  int gmask_act = 0;
};

// Raw buffer view of one image of an NCHW tensor: loads / stores whose byte offset is >= num_records are
// dropped (loads return 0) by the hardware, so edge tiles need no predication at all.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const float* base, size_t floats) {
  const size_t bytes = floats * 4;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes < 0xFFFFFFFFull ? (int)bytes : -1, 0x00020000);
}

template <bool FULL, int MT, int NT>
__device__ __forceinline__ void store_mfma_tile_impl(const f32x16 (&acc)[MT][NT], const TileOut& t, int n,
                                                     int co_block, int hi, int oy_first, int ox) {
  const size_t HWo = (size_t)t.Ho * t.Wo;
  const float slope = t.act == ACT_LRELU ? 0.1f : (t.act == ACT_RELU ? 0.f : 1.f);  // branch-free act
  const int co_base = co_block + 4 * hi;
  float bv[MT][16];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[mt][r] = 0.f;
  if (t.bias) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2);
        bv[mt][r] = t.bias[(FULL || co < t.Cout) ? co : t.Cout - 1];
      }
  }
  const float neg = t.gmask_act == ACT_LRELU ? 0.1f : (t.gmask_act == ACT_RELU ? 0.f : 1.f);
  if (FULL) {
    // Full tiles: address = wave-uniform channel plane (scalar registers) + one per-lane 32-bit byte offset
    // shared by all 16 registers, i.e. no vector address arithmetic per store.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy_first + nt;
        if (t.ps == 0) {
          const unsigned voff = (unsigned)(((size_t)(4 * hi) * HWo + (size_t)oy * t.Wo + ox) * 4);
          auto addr = [&](const float* base, int r) -> const float* {
            const int cu = co_block + mt * 32 + (r & 3) + 8 * (r >> 2);  // uniform part of the channel
            return reinterpret_cast<const float*>(
                reinterpret_cast<const char*>(base + ((size_t)n * t.Cout + cu) * HWo) + voff);
          };
          if (!t.res && !t.accum && !t.gmask) {
            // the common case: three VALU instructions per value (nothing hides behind the fp32 MFMAs of the
            // co-resident workgroups, profiles/r02_z_mfma_shadow.txt).  act(v) = max(v, slope v) for slope in [0, 1]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = acc[mt][nt][r] + bv[mt][r];
              *const_cast<float*>(addr(t.y, r)) = fmaxf(v, slope * v);
            }
            continue;
          }
          float extra[16], gm[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) { extra[r] = 0.f; gm[r] = 1.f; }
          if (t.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) extra[r] = *addr(t.res, r);
          }
          if (t.accum) {
#pragma unroll
            for (int r = 0; r < 16; ++r) extra[r] += *addr(t.y, r);
          }
          if (t.gmask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gm[r] = *addr(t.gmask, r) > 0.f ? 1.f : neg;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[mt][nt][r] + bv[mt][r];
            v = (fmaxf(v, slope * v) + extra[r]) * gm[r];  // act: v > 0 ? v : slope * v
            *const_cast<float*>(addr(t.y, r)) = v;
          }
        } else {
          // pixel shuffle: channels co and co + 1 (registers r, r + 1) are the horizontal neighbours 2 ox, 2 ox + 1
          // of one output row -> one 8-byte store per pair, 256 contiguous bytes per 32 lanes
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2);
            float v0 = acc[mt][nt][r] + bv[mt][r], v1 = acc[mt][nt][r + 1] + bv[mt][r + 1];
            v0 = fmaxf(v0, slope * v0);
            v1 = fmaxf(v1, slope * v1);
            const int cq = co >> 2, dy = (co >> 1) & 1;
            *reinterpret_cast<f32x2*>(&t.y[(((size_t)n * (t.Cout >> 2) + cq) * (2 * t.Ho) + (2 * oy + dy)) * (size_t)(2 * t.Wo) +
                                         2 * ox]) = f32x2{v0, v1};
          }
        }
      }
    }
    return;
  }
  // Edge tiles (common on the small inner-step clips: 44x80 is 2.5 tiles wide): every access goes through
  // a bounds-checked raw buffer of this image with the byte offset forced out of range for invalid
  // (row, column, channel) combinations -- still straight-line code, no store waits for another store.
  const size_t img = (size_t)t.Cout * HWo;
  const __amdgpu_buffer_rsrc_t ry = image_rsrc(t.y + (size_t)n * img, img);
  const __amdgpu_buffer_rsrc_t rr = image_rsrc(t.res ? t.res + (size_t)n * img : t.y, t.res ? img : 0);
  const __amdgpu_buffer_rsrc_t rg = image_rsrc(t.gmask ? t.gmask + (size_t)n * img : t.y, t.gmask ? img : 0);
  const __amdgpu_buffer_rsrc_t ra = image_rsrc(t.y + (size_t)n * img, t.accum ? img : 0);
  const bool col_ok = ox < t.Wo;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = oy_first + nt;
      const bool px_ok = col_ok && oy < t.Ho;
      unsigned off[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2);
        size_t o;
        if (t.ps == 0) o = (size_t)co * HWo + (size_t)oy * t.Wo + ox;
        else o = ((size_t)(co >> 2) * (2 * t.Ho) + (2 * oy + ((co >> 1) & 1))) * (size_t)(2 * t.Wo) + (2 * ox + (co & 1));
        off[r] = (px_ok && co < t.Cout) ? (unsigned)(o * 4) : 0xFFFFFFFFu;
      }
      float extra[16], gm[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        extra[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, off[r], 0, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r)
        extra[r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, off[r], 0, 0));
      // (all sixteen mask loads first, unconditionally -- without a mask the resource is empty and they return 0 --: written as
      // `t.gmask && load(...) <= 0` every load sat in its own branch with a full wait: sixteen dependent round trips per tile row)
#pragma unroll
      for (int r = 0; r < 16; ++r) gm[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, off[r], 0, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r) gm[r] = (t.gmask && gm[r] <= 0.f) ? neg : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[mt][nt][r] + bv[mt][r];
        v = (fmaxf(v, slope * v) + extra[r]) * gm[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, off[r], 0, 0);
      }
    }
  }
}

// Compute units of the current device (cached per ordinal; 256 on MI355X).
inline int device_cus() {
  static int cus[64] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  d &= 63;
  if (!cus[d]) {
    hipDeviceProp_t prop;
    cus[d] = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cus[d];
}

// Largest dynamic LDS a workgroup of the current device may ask for (cached per ordinal; 160 KB on MI355X): the Winograd
// kernels need 147 - 155 KB and are only chosen where that exists.
inline size_t device_lds_optin() {
  static size_t lds[64] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  d &= 63;
  if (!lds[d]) {
    hipDeviceProp_t prop;
    lds[d] = hipGetDeviceProperties(&prop, d) == hipSuccess ? (size_t)prop.sharedMemPerBlockOptin : 0;
    if (!lds[d]) lds[d] = 65536;
  }
  return lds[d];
}

// hipFuncSetAttribute (the dynamic-LDS limit of a kernel) applies to the CURRENT device only: a launcher keeps one flag per
// device ordinal, so a process that drives several GPUs (or rebuilds plans after hipSetDevice) sets it on each of them.
struct PerDeviceOnce {
  bool done[64] = {};
  static int slot() {
    int d = 0;
    (void)hipGetDevice(&d);
    return (d >= 0 && d < 64) ? d : 63;   // (devices past the table share its last slot: the call is merely repeated)
  }
  // needs(): the per-device action has not been COMPLETED on this device yet; mark(): it has (call it after the action
  // succeeded).  Two host threads may both run the action -- it is idempotent (hipFuncSetAttribute) -- but none can launch
  // before one of them has finished it, which the earlier set-then-act order allowed.
  bool needs() const { return !__atomic_load_n(&done[slot()], __ATOMIC_ACQUIRE); }
  void mark() { __atomic_store_n(&done[slot()], true, __ATOMIC_RELEASE); }
};
// Raises the dynamic-LDS limit of a kernel once per device.  The flag is set only after the call SUCCEEDED: a failure is
// retried by the next launch, and that launch fails loudly (check_launch) instead of running with a stale limit.
inline void set_dyn_lds_once(PerDeviceOnce& once, const void* kern, size_t bytes) {
  if (!once.needs()) return;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) once.mark();
  else (void)hipGetLastError();
}

// Store modes carried in TileOut::ps beyond 0 (plain) and 2 (PixelShuffle(2)): the tile goes STRAIGHT into the
// explicitly padded tensor the next convolution of the estimator reads (LRimg_estimator.py:82-86: ReflectionPad2d(1) in
// front of every 2-D conv), so that no pad kernel re-reads and re-writes the whole activation (r03):
//   PS_PAD_REFLECT      P[n][c][oy + 1][ox + 1] of a [N][C][Ho + 2][Wo + 2] tensor, plus the mirrored copies the
//                       reflection ring holds of rows 1, Ho - 2 and columns 1, Wo - 2 (up to four stores per value,
//                       in the tiles that touch them);
//   PS_PAD_REFLECT_S2D  the same padded image in the space-to-depth layout [N][4C][(Ho + 2) / 2][(Wo + 2) / 2] of the
//                       re-expressed 4x4 stride-2 convs (channel 4c + 2 (py & 1) + (px & 1), position (py >> 1, px >> 1)).
enum : int { PS_PAD_REFLECT = 16, PS_PAD_REFLECT_S2D = 17 };

template <int MT, int NT>
__device__ __forceinline__ void store_mfma_tile_padded(const f32x16 (&acc)[MT][NT], const TileOut& t, int n, int co_block,
                                                       int hi, int oy_first, int ox) {
  const bool s2d = t.ps == PS_PAD_REFLECT_S2D;
  const int Hp = t.Ho + 2, Wp = t.Wo + 2;
  const int Hs = s2d ? Hp >> 1 : Hp, Ws = s2d ? Wp >> 1 : Wp, cmul = s2d ? 4 : 1;
  const size_t plane = (size_t)Hs * Ws, img = (size_t)t.Cout * cmul * plane;
  const __amdgpu_buffer_rsrc_t ry = image_rsrc(t.y + (size_t)n * img, img);
  const float slope = t.act == ACT_LRELU ? 0.1f : (t.act == ACT_RELU ? 0.f : 1.f);
  const int co_base = co_block + 4 * hi;
  // element offset of padded position (py, px) inside channel group 0 (the channel term is added per register)
  auto loff = [&](int py, int px) -> size_t {
    return s2d ? (size_t)(2 * (py & 1) + (px & 1)) * plane + (size_t)(py >> 1) * Ws + (px >> 1) : (size_t)py * Ws + px;
  };
  const bool col_ok = ox < t.Wo;
  // this lane's column also feeds the ring: column 1 -> padded column 0, column Wo - 2 -> padded column Wp - 1 (two
  // independent mirrors, like the rows: with Wo == 3 column 1 feeds BOTH)
  const int pxm[2] = {ox == 1 ? 0 : -1, ox == t.Wo - 2 ? Wp - 1 : -1};
  const bool col_ring[2] = {__builtin_amdgcn_ballot_w64(pxm[0] >= 0 && col_ok) != 0,
                            __builtin_amdgcn_ballot_w64(pxm[1] >= 0 && col_ok) != 0};
  // (the bias values of all registers first, with a clamped index: fetched one per value inside the loop below each load sat in
  // its own branch with a full wait -- sixteen dependent round trips per M-tile and thread)
  float bvp[MT][16];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2);
      bvp[mt][r] = t.bias ? t.bias[co < t.Cout ? co : t.Cout - 1] : 0.f;
    }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float v[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = acc[mt][nt][r] + bvp[mt][r];
        v[nt][r] = fmaxf(x, slope * x);
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = oy_first + nt;                                    // wave-uniform
      if (oy >= t.Ho) continue;
      const int pys[3] = {oy + 1, oy == 1 ? 0 : -1, oy == t.Ho - 2 ? Hp - 1 : -1};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (pys[a] < 0) continue;                                      // wave-uniform
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          if (b >= 1 && !col_ring[b - 1]) continue;                    // wave-uniform
          const int px = b == 0 ? ox + 1 : pxm[b - 1];
          const bool ok = col_ok && px >= 0;
          const size_t lo_ = ok ? loff(pys[a], px) : 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co_base + mt * 32 + (r & 3) + 8 * (r >> 2);
            const size_t o = (size_t)co * cmul * plane + lo_;
            const unsigned off = (ok && co < t.Cout) ? (unsigned)(o * 4) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[nt][r]), ry, off, 0, 0);
          }
        }
      }
    }
  }
}

// co_block: first output channel of the workgroup's 32*MT block; (oy0, ox0): tile origin, TH rows;
// oy_first: first of the NT rows this wave owns; lane = (lo, hi).
template <int MT, int NT>
__device__ __forceinline__ void store_mfma_tile(const f32x16 (&acc)[MT][NT], const TileOut& t, int n,
                                                int co_block, int oy0, int th, int ox0, int oy_first, int lo,
                                                int hi) {
  if (t.ps >= PS_PAD_REFLECT) {   // (no residual / accumulate / gradient mask in these modes: forward of the estimator)
    store_mfma_tile_padded<MT, NT>(acc, t, n, co_block, hi, oy_first, ox0 + lo);
    return;
  }
  const bool full = oy0 + th <= t.Ho && ox0 + 32 <= t.Wo && co_block + MT * 32 <= t.Cout;
  const int ox = ox0 + lo;
  if (full) {
    store_mfma_tile_impl<true, MT, NT>(acc, t, n, co_block, hi, oy_first, ox);
  } else {
    store_mfma_tile_impl<false, MT, NT>(acc, t, n, co_block, hi, oy_first, ox);
  }
}

}  // namespace dvsr
