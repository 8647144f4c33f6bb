// common.hpp — error plumbing shared by the host-side translation units of libdawn_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/dawn_hip.h"

namespace dawn {

std::string& last_error();  // thread-local

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

#define DAWN_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ::dawn::fail(_e == hipErrorOutOfMemory ? DAWN_ERR_OOM : DAWN_ERR_HIP, "%s: %s", \
                                #expr, hipGetErrorString(_e));                                      \
    } while (0)

#define DAWN_TRY(expr)          \
    do {                        \
        int _rc = (expr);       \
        if (_rc != DAWN_OK) return _rc; \
    } while (0)

// Is there a usable HIP device?  (no CPU fallback anywhere in this library)
int require_device(int device);

// vector.rs host restatements used by the ABI-side validation
bool host_is_normalized(const float* v);

}  // namespace dawn
