// Does the f32 MFMA rate depend on WHERE the accumulators live (AGPRs vs VGPRs) and on waves per SIMD?
// Register-only chains of v_mfma_f32_32x32x2_f32 on real (non-constant) operands.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_acc_lab.hip -o tools/_bin/mfma_acc_lab && tools/_bin/mfma_acc_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BODY                                                                                              \
    f32x16 acc[4];                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 16; ++j) acc[i][j] = 0.f; \
    float a0 = 1.f + threadIdx.x * 1e-3f, a1 = 0.5f - threadIdx.x * 2e-3f, b0 = 1.f + (threadIdx.x & 31) * 3e-3f, b1 = -0.7f; \
    for (int it = 0; it < iters; ++it) {                                                                  \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);                           \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);                           \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);                           \
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);                           \
    }                                                                                                     \
    float s = 0.f;                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 16; ++j) s += acc[i][j]; \
    if (s == 12345.678f) out[0] = s;

__global__ __launch_bounds__(256) void k_default(float* out, int iters) { BODY }
__global__ __launch_bounds__(256, 2) void k_lb2(float* out, int iters) { BODY }
__global__ __launch_bounds__(256, 3) void k_lb3(float* out, int iters) { BODY }
__global__ __launch_bounds__(256, 4) void k_lb4(float* out, int iters) { BODY }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class K>
static int run(const char* name, K kern, int wgs_per_cu, int iters, float* d, int lds = 0) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int grid = p.multiProcessorCount * wgs_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<grid, 256, lds>>>(d, iters / 8);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        kern<<<grid, 256, lds>>>(d, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double flop = double(grid) * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
    printf("%-34s WG/CU=%d iters=%6d  best %.3f ms = %7.2f TFLOP/s\n", name, wgs_per_cu, iters, best, flop / best / 1e9);
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 64));
    // workgroup turnover: the same total work as 3 resident workgroups per CU of `iters`, cut into more, shorter workgroups
    // (3 per CU resident by the LDS allocation): what does refilling a slot cost?
    printf("-- turnover: 50 KB of LDS per workgroup (3 resident per CU), total MFMAs fixed at 3 x 2048 x 4 per SIMD --\n");
    for (int parts : {1, 2, 4, 8, 16}) {
        char nm[64]; snprintf(nm, sizeof nm, "lb3, %d round(s), lds 50 KB", parts);
        if (run(nm, k_lb3, 3 * parts, 2048 / parts, d, 49920)) return 1;
    }
    for (int parts : {1, 2, 4, 8, 16}) {
        char nm[64]; snprintf(nm, sizeof nm, "lb3, %d round(s), no lds", parts);
        if (run(nm, k_lb3, 3 * parts, 2048 / parts, d, 0)) return 1;
    }
    for (int iters : {64}) {
        for (int w = 1; w <= 4; ++w) {
            if (run("launch_bounds(256)    ", k_default, w, iters, d)) return 1;
            if (w <= 2 && run("launch_bounds(256,2)", k_lb2, w, iters, d)) return 1;
            if (w <= 3 && run("launch_bounds(256,3)", k_lb3, w, iters, d)) return 1;
            if (run("launch_bounds(256,4)", k_lb4, w, iters, d)) return 1;
        }
    }
    return 0;
}
