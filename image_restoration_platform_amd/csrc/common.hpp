// common.hpp -- shared host-side helpers for libire.so (error plumbing, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "errors.hpp"

namespace ire {

void set_last_error(int code, const std::string& msg);

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        char buf[512];
        std::snprintf(buf, sizeof(buf), "internal: %s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
        // a lost / absent device is reported as 503 so the worker's retry policy applies
        int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? IRE_ERR_UNAVAILABLE : IRE_ERR_INTERNAL;
        if (e == hipErrorOutOfMemory) {     // a retry (smaller batch / later) can succeed: 503, not an internal error
            code = IRE_ERR_UNAVAILABLE;
            (void)hipGetLastError();
            std::snprintf(buf, sizeof(buf), "service unavailable: out of device memory (%s, %s:%d)", what, file, line);
        } else if (code == IRE_ERR_UNAVAILABLE)
            std::snprintf(buf, sizeof(buf), "service unavailable: %s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
        throw Error{code, buf};
    }
}
#define IRE_HIP(expr) ::ire::hip_check((expr), #expr, __FILE__, __LINE__)

template <typename T>
inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

}  // namespace ire
