// Host-side plumbing shared by every entry point: thread-local error text, launch check, version.
#include <stdarg.h>

#include "pd_common.h"

namespace pd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return PD_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return PD_ERR_LAUNCH;
}

}  // namespace pd

extern "C" int pd_version(void) { return 200; /* 0.2.0: PD_HOMO_UNIFORM, PD_BWD_ACCUMULATE, pd_homography_matrices_*, pd_masked_photometric_*, pd_crop_grid; 0.1.1: fused mean of ph_map */ }
extern "C" const char* pd_last_error(void) { return pd::g_err; }
