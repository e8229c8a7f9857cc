// Thread-local error string behind dvsr_last_error().
#include "common.h"
#include "../../include/dynavsr_hip.h"

namespace dvsr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DVSR_ERR_HIP;
  }
  return DVSR_OK;
}
}  // namespace dvsr

extern "C" const char* dvsr_last_error(void) { return dvsr::g_err; }
extern "C" int dvsr_version(void) { return 100; }
