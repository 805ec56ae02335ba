// Shared host-side helpers for libspk_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/spk.h"

namespace spk {

inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SPK_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SPK_OK;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace spk

#define SPK_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return spk::fail(SPK_EINVAL, __VA_ARGS__); \
    } while (0)
