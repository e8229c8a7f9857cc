// Random patch crops of the inner MAML step (train.maml.use_patch): test_dynavsr.py:118-145 / train_dynavsr.py:208-243
// draw `num_patch` positions with preprocessing.common_crop (:57-85) and stack the crops of the SLR clip
// [T][C][h][w] and of its target [C][s*h][s*w] into batches.  The positions are drawn on the host (python `random`,
// same order as the reference); the gather of all patches is ONE launch, its adjoint (patches overlap) one launch
// with atomics.  dst[p][n][y][x] = src[n][s*py_p + y][s*px_p + x], n = plane (T*C), patch edge = s*ps.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "kernels.h"

namespace dvsr {

constexpr int PATCH_MAX = 64;
struct PatchPos {
  int n;
  int py[PATCH_MAX], px[PATCH_MAX];
};

template <bool BWD>
__global__ void patch_gather_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ gsrc,
                                    const float* __restrict__ gdst, PatchPos pos, int planes, int H, int W, int E) {
  const size_t per = (size_t)planes * E * E, total = per * pos.n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i / per);
    size_t r = i - (size_t)p * per;
    const int x = (int)(r % E); r /= E;
    const int y = (int)(r % E);
    const int n = (int)(r / E);
    const size_t s = ((size_t)n * H + pos.py[p] + y) * W + pos.px[p] + x;
    if (BWD) unsafeAtomicAdd(gsrc + s, gdst[i]);
    else dst[i] = src[s];
  }
}

}  // namespace dvsr

using namespace dvsr;

static int patch_args(const int* py, const int* px, int P, int planes, int H, int W, int edge, int scale, PatchPos* pos) {
  DVSR_REQUIRE(py && px && P > 0 && P <= PATCH_MAX && planes > 0 && H > 0 && W > 0 && edge > 0 && scale > 0, DVSR_ERR_INVALID,
               "patch_gather: bad argument (patches=%d, at most %d)", P, PATCH_MAX);
  pos->n = P;
  for (int i = 0; i < P; ++i) {
    DVSR_REQUIRE(py[i] >= 0 && px[i] >= 0 && scale * (py[i] + edge) <= H && scale * (px[i] + edge) <= W, DVSR_ERR_INVALID,
                 "patch_gather: patch %d at (%d, %d) x %d (scale %d) leaves the %dx%d source", i, py[i], px[i], edge, scale, H, W);
    pos->py[i] = scale * py[i];
    pos->px[i] = scale * px[i];
  }
  return DVSR_OK;
}

extern "C" int dvsr_patch_gather_forward(const float* src, float* dst, const int* py, const int* px, int P, int planes,
                                         int H, int W, int edge, int scale, dvsr_stream_t stream) {
  DVSR_REQUIRE(src && dst, DVSR_ERR_INVALID, "patch_gather_forward: null pointer");
  PatchPos pos;
  int rc = patch_args(py, px, P, planes, H, W, edge, scale, &pos);
  if (rc) return rc;
  const int E = edge * scale;
  const size_t n = (size_t)P * planes * E * E;
  size_t g = (n + 255) / 256;
  hipLaunchKernelGGL(patch_gather_kernel<false>, dim3((unsigned)(g < 4096 ? g : 4096)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, nullptr, nullptr, pos, planes, H, W, E);
  return check_launch("patch_gather_kernel");
}

extern "C" int dvsr_patch_gather_backward(const float* grad_dst, float* grad_src, const int* py, const int* px, int P,
                                          int planes, int H, int W, int edge, int scale, dvsr_stream_t stream) {
  DVSR_REQUIRE(grad_dst && grad_src, DVSR_ERR_INVALID, "patch_gather_backward: null pointer");
  PatchPos pos;
  int rc = patch_args(py, px, P, planes, H, W, edge, scale, &pos);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  DVSR_REQUIRE(hipMemsetAsync(grad_src, 0, (size_t)planes * H * W * sizeof(float), st) == hipSuccess, DVSR_ERR_HIP,
               "patch_gather_backward: memset failed");
  const int E = edge * scale;
  const size_t n = (size_t)P * planes * E * E;
  size_t g = (n + 255) / 256;
  hipLaunchKernelGGL(patch_gather_kernel<true>, dim3((unsigned)(g < 4096 ? g : 4096)), dim3(256), 0, st, nullptr, nullptr,
                     grad_src, grad_dst, pos, planes, H, W, E);
  return check_launch("patch_gather_kernel(bwd)");
}
