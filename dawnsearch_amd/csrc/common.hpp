// common.hpp — error plumbing shared by the host-side translation units of libdawn_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "host_common.hpp"

#define DAWN_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ::dawn::fail(_e == hipErrorOutOfMemory ? DAWN_ERR_OOM : DAWN_ERR_HIP, "%s: %s", \
                                #expr, hipGetErrorString(_e));                                      \
    } while (0)

namespace dawn {

// Is there a usable HIP device?  (no CPU fallback anywhere in this library)
int require_device(int device);

}  // namespace dawn
