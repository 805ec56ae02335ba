// Where do workgroups land?  Records HW_ID / XCC_ID of wave 0 of every workgroup of a (gx, gy) grid with the conv
// kernel's footprint (256 threads, ~43 KB LDS) and prints, per CU, the linear workgroup ids it received and their
// wave slots -- the facts an XCD-aware tile order and a co-residency stagger have to be built on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
        const unsigned id = blockIdx.x + gridDim.x * blockIdx.y;
        out[2 * id] = hw; out[2 * id + 1] = xcc;
    }
    lds[threadIdx.x] = 1.f;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);   // keep the workgroup resident while the grid fills
}

int main() {
    unsigned* d; CK(hipMalloc(&d, 1 << 20));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grids[3][2] = {{256, 2}, {1024, 1}, {2048, 1}};
    for (auto& g : grids) {
        const int n = g[0] * g[1];
        CK(hipMemset(d, 0, 1 << 20));
        probe<<<dim3(g[0], g[1]), 256, 43424>>>(d, 2000);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(2 * n); CK(hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost));
        std::map<unsigned, std::vector<int>> cu;   // key: xcc | se | sh | cu
        for (int i = 0; i < n; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
            cu[key].push_back(i); cu[key].push_back(hw & 0xf);
        }
        printf("grid %dx%d: %zu distinct CUs\n", g[0], g[1], cu.size());
        int shown = 0;
        for (auto& kv : cu) {
            if (shown++ % 37 && shown > 6) continue;
            printf("  xcc %u se %u sh %u cu %2u :", kv.first >> 16, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 0xf);
            for (size_t j = 0; j < kv.second.size(); j += 2) printf(" wg%d(slot%d)", kv.second[j], kv.second[j + 1]);
            printf("\n");
        }
    }
    return 0;
}
