// shems_internal.h -- shared by the translation units of libshems_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/shems_hip.h"

namespace shems {
int set_error(int code, const char *fmt, ...);          // records the thread-local message, returns code
int hip_ok(hipError_t e, const char *what);             // SHEMS_OK or SHEMS_ERR_HIP (+ message)
int check_view(const shems_view *v, const char *fn);    // every entry point that dereferences a caller-built view (shems_env.hip)
}
