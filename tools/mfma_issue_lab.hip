// What slows a wave's MFMA stream down?  Three workgroups per CU (launch_bounds(256, 3), 50 KB of LDS each), every wave runs
// `tiles` iterations of 32 x v_mfma_f32_32x32x2_f32 (4 accumulators) with, per iteration and by FLAGS:
//   1  a few VALU compares + an exec-masked (never taken) ds_write     2  four 4-byte global loads, waited for one iteration later
//   4  one s_barrier per iteration                                    8  16 ds_read_b32 feeding the MFMA operands
//  16  two ds_write_b128 per iteration
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_issue_lab.hip -o tools/_bin/mfma_issue_lab && tools/_bin/mfma_issue_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FLAGS>
__global__ __launch_bounds__(256, 3) void k(float* out, const float* tab, int tiles, float magic) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    float a0 = 1.f + lane * 1e-3f, a1 = 0.5f - lane * 2e-3f, b0 = 1.f + (lane & 31) * 3e-3f, b1 = -0.7f;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    lds[tid] = a0; lds[tid + 256] = a1; lds[tid + 512] = b0; lds[tid + 768] = b1;
    __syncthreads();
    for (int it = 0; it < tiles; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (FLAGS & 8) { a0 = lds[lane + 64 * s]; a1 = lds[lane + 256 + 64 * s]; b0 = lds[lane + 512 + 64 * s]; b1 = lds[lane + 768 + 64 * s]; }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (FLAGS & 1) {
            float v = fmaxf(g[0] * g[1] + g[2], 0.f);
            if (v == magic && tid + it < tiles) lds[1024 + tid] = v + g[3];
        }
        if (FLAGS & 16) {
            *reinterpret_cast<float4*>(lds + 2048 + tid * 4) = make_float4(g[0], g[1], g[2], g[3]);
            *reinterpret_cast<float4*>(lds + 4096 + tid * 4) = make_float4(g[1], g[2], g[3], g[0]);
        }
        if (FLAGS & 4) __syncthreads();
        if (FLAGS & 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = tab[(it * 4 + q) & 63];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 2; s < 8; ++s) {
            if (FLAGS & 8) { a0 = lds[lane + 64 * (s & 3)]; a1 = lds[lane + 256 + 64 * (s & 3)]; b0 = lds[lane + 512 + 64 * (s & 3)]; b1 = lds[lane + 768 + 64 * (s & 3)]; }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    float s = g[0] + g[1] + g[2] + g[3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FLAGS>
static int run(int wgs_per_cu, int tiles, float* d, const float* tab) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int grid = p.multiProcessorCount * wgs_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<FLAGS><<<grid, 256, 49920>>>(d, tab, tiles, 1e30f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        k<FLAGS><<<grid, 256, 49920>>>(d, tab, tiles, 1e30f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double flop = double(grid) * 4 * tiles * 32 * (2.0 * 32 * 32 * 2);
    printf("flags %2d  WG/CU=%d tiles=%4d  best %8.1f us = %7.2f TFLOP/s\n", FLAGS, wgs_per_cu, tiles, best * 1e3, flop / best / 1e9);
    return 0;
}

int main() {
    float *d, *tab; CK(hipMalloc(&d, 64)); CK(hipMalloc(&tab, 256)); CK(hipMemset(tab, 0, 256));
    for (int tiles : {8, 64}) {
        for (int w : {3, 6}) {
            if (run<0>(w, tiles, d, tab)) return 1;
            if (run<1>(w, tiles, d, tab)) return 1;
            if (run<2>(w, tiles, d, tab)) return 1;
            if (run<3>(w, tiles, d, tab)) return 1;
            if (run<4>(w, tiles, d, tab)) return 1;
            if (run<8>(w, tiles, d, tab)) return 1;
            if (run<12>(w, tiles, d, tab)) return 1;
            if (run<16>(w, tiles, d, tab)) return 1;
            if (run<31>(w, tiles, d, tab)) return 1;
        }
    }
    return 0;
}
