// shems_internal.h -- shared by the translation units of libshems_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "../../include/shems_hip.h"

namespace shems {
int set_error(int code, const char *fmt, ...);          // records the thread-local message, returns code
int hip_ok(hipError_t e, const char *what);             // SHEMS_OK or SHEMS_ERR_HIP (+ message)
// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, and the attribute is PER DEVICE: a
// C-ABI caller may drive several GPUs from one process (shems_create(..., device, ...)), so the opt-in is remembered per
// (kernel, current device) -- `mask` is the kernel's own static bit set, bit = device ordinal.
inline int lds_optin(std::atomic<uint64_t> &mask, const void *fn, int bytes, const char *what)
{
    int dev = 0;
    if (int rc = hip_ok(hipGetDevice(&dev), "hipGetDevice")) return rc;
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return SHEMS_OK;
    if (int rc = hip_ok(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what)) return rc;
    mask.fetch_or(bit, std::memory_order_release);
    return SHEMS_OK;
}
int check_view(const shems_view *v, const char *fn);    // every entry point that dereferences a caller-built view (shems_env.hip)
}
