// Shared host/device helpers for libdynavsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

namespace dvsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

}  // namespace dvsr
