// Shared helpers for libmerfish_hip.so (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "merfish_hip.h"

namespace mf {

// thread-local error text behind mf_last_error_string()
char *error_buffer();
int fail(int code, const char *fmt, ...);

#define MF_HIP_CHECK(expr)                                                    \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess)                                                 \
            return mf::fail(MF_EHIP, "%s failed: %s (%s:%d)", #expr,          \
                            hipGetErrorString(e_), __FILE__, __LINE__);       \
    } while (0)

int device_cu_count();

}  // namespace mf
