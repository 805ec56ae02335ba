// Sustained f32 MFMA ceiling on this box: register-only chains of v_mfma_f32_32x32x2_f32, no memory traffic.
// Tells how much of the 157.3 TFLOP/s paper peak (256 CU x 256 FLOP/clk x 2.4 GHz) the clocks actually sustain,
// i.e. the real ceiling the conv kernels are measured against.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/_bin/mfma_peak && tools/_bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(out)[1] = t1 - t0;   // s_memtime ticks of the loop
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) out[0] = s;   // keep the chain alive
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC>
static int run(const char* name, int wgs_per_cu, int iters, float* d) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int grid = p.multiProcessorCount * wgs_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mfma_loop<NACC><<<grid, 256>>>(d, iters / 8, 1.f, 1.f);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0.f; const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        mfma_loop<NACC><<<grid, 256>>>(d, iters, 1.f, 1.f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        sum += ms;
    }
    unsigned long long ticks = 0; CK(hipMemcpy(&ticks, d + 2, 8, hipMemcpyDeviceToHost));
    printf("   s_memtime: %llu ticks for %d x %d MFMAs of one wave (x%d waves/SIMD) = %.2f ticks/MFMA; tick rate %.0f MHz\n", ticks, iters, NACC,
           wgs_per_cu, double(ticks) / (double(iters) * NACC), ticks / (sum / reps) / 1e3);
    double flop = double(grid) * 4 /*waves*/ * iters * NACC * (2.0 * 32 * 32 * 2);
    printf("%-28s grid=%5d iters=%7d  best %.3f ms = %7.2f TFLOP/s   mean %.3f ms = %7.2f TFLOP/s  (implied clock %.0f MHz)\n", name, grid,
           iters, best, flop / best / 1e9, sum / reps, flop / (sum / reps) / 1e9,
           flop / (sum / reps) / 1e9 * 1e6 / (p.multiProcessorCount * 256.0));
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 64));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s  CUs=%d  clockRate=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    // short (≈ a conv launch) and long (thermal steady state) runs, 1 and 2 workgroups per CU
    if (run<4>("4 acc, 1 WG/CU, ~0.4 ms", 1, 20000, d)) return 1;
    if (run<4>("4 acc, 2 WG/CU, ~0.8 ms", 2, 20000, d)) return 1;
    if (run<4>("4 acc, 2 WG/CU, ~40 ms", 2, 1000000, d)) return 1;
    if (run<8>("8 acc, 1 WG/CU, ~40 ms", 1, 1000000, d)) return 1;
    CK(hipFree(d));
    return 0;
}
