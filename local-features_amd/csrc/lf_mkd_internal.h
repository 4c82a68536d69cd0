// What the other translation units of the C ABI may ask of a handle (its definition stays in lf_mkd.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/lf_mkd.h"

int lf_mkd_internal_device(const lf_mkd *h);
hipStream_t lf_mkd_internal_stream(lf_mkd *h);
int lf_mkd_internal_fail(lf_mkd *h, int code, const std::string &msg);

namespace lfmkd {
// The caller's current device, saved on entry and restored when the scope ends (any return path).  One hipGetDevice per call
// (~0.1 us: the runtime keeps the ordinal in thread-local storage); hipSetDevice only when the handle lives on another device.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int device) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev == device) return hipSuccess;
        e = hipSetDevice(device);
        switched = e == hipSuccess;
        return e;
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceScope() = default;
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
}  // namespace lfmkd
