// device_probe.cpp — the two ABI pieces that only ask the HIP runtime which devices exist.  (Kept apart from
// host_helpers.cpp so that file stays free of HIP headers.)
#include "common.hpp"

namespace dawn {

int require_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(DAWN_ERR_NO_DEVICE, "no usable HIP device (%s); libdawn_hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(DAWN_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    return DAWN_OK;
}

}  // namespace dawn

extern "C" int dawn_device_count(int* count) {
    if (!count) return dawn::fail(DAWN_ERR_INVALID_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return dawn::fail(DAWN_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return DAWN_OK;
}
