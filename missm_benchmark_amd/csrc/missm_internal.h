// Internal host-side helpers: error reporting for the C ABI (int status + thread-local message).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/missm_hip.h"

void missm_set_error(const char* fmt, ...);
// per-stream scratch of the library (gemm.hip: the split-K workspace, >= 32 MB, grown on demand; users on one stream are stream-ordered)
int missm_stream_workspace(void* stream, size_t bytes, float** ws);

#define MISSM_CHECK_ARG(cond, msg)                         \
  do {                                                     \
    if (!(cond)) {                                         \
      missm_set_error("%s (%s:%d)", msg, __FILE__, __LINE__); \
      return MISSM_ERR_INVALID;                            \
    }                                                      \
  } while (0)

static inline int missm_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    missm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MISSM_ERR_LAUNCH;
  }
  return MISSM_OK;
}
