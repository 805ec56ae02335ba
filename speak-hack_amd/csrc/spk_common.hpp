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

#ifdef __HIPCC__
// Sum over each group of N consecutive lanes (N = 8, 16, 32, 64), valid in the LAST lane of every group only (the other lanes
// hold partial sums).  row_shr steps inside the 16-lane DPP rows, then row_bcast:15 / row_bcast:31 across rows: VALU adds
// with DPP operands, no LDS crossbar (`__shfl_xor` compiles to ds_bpermute_b32 here: 10 LDS round trips per pair of sums,
// which made the BatchNorm statistics a quarter of the 64 -> 256 @64^2 trunk conv: tools/lab_gemm1x1.py).  Every lane of the
// wave must execute it.
#define SPK_DPP_ADD(ctrl_, rows_) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, rows_, 0xf, false))
template <int N>
__device__ inline float lane_group_sum_hi(float v) {
    static_assert(N == 8 || N == 16 || N == 32 || N == 64, "group of 8, 16, 32 or 64 lanes");
    SPK_DPP_ADD(0x111, 0xf);                        // row_shr:1
    SPK_DPP_ADD(0x112, 0xf);                        // row_shr:2
    SPK_DPP_ADD(0x114, 0xf);                        // row_shr:4
    if constexpr (N >= 16) SPK_DPP_ADD(0x118, 0xf); // row_shr:8
    if constexpr (N >= 32) SPK_DPP_ADD(0x142, 0xa); // row_bcast:15 into rows 1 and 3
    if constexpr (N == 64) SPK_DPP_ADD(0x143, 0xc); // row_bcast:31 into rows 2 and 3
    return v;
}
#undef SPK_DPP_ADD
__device__ inline float half_wave_sum_hi(float v) { return lane_group_sum_hi<32>(v); }
#endif

}  // namespace spk

#define SPK_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return spk::fail(SPK_EINVAL, __VA_ARGS__); \
    } while (0)
