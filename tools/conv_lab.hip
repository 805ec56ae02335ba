// Stand-alone timing of the conv kernel on the decoder's heavy layers (B = 8, production tile configs, zero data),
// without PyTorch in the process:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_lab.hip -o tools/_bin/conv_lab && tools/_bin/conv_lab
// With -DSPK_LAB_STEPTIME the kernel stamps s_memtime after every k-step of chunk 10 of one workgroup and the tool
// prints the per-step cycle counts of its four waves (how profiles/r01_e_* located the staging costs).
#include "../speak-hack_amd/csrc/conv_mfma_f32.hpp"
#ifndef SPK_LAB
#define SPK_LAB 0
#endif

#include <cstdlib>
#include <vector>

namespace spkconv {
int launch_splitk_epilogue(const ConvArgs&, const float*, int, hipStream_t) { return 0; }
}  // namespace spkconv

using namespace spkconv;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class C, int MODE>
static int bench(const char* name, int B, int Cin, int Cout, int H, float* x, float* w, float* y, float* aux) {
    spk_conv2d_desc d = {};
    const bool ups = MODE == MODE_UPSAMPLE;
    d.x = x; d.w_packed = w; d.y = y; d.bias = aux; d.noise_w = aux; d.noise = aux; d.style = aux;
    d.B = B; d.Cin = Cin; d.Cout = Cout; d.H = H; d.W = H; d.Hin = ups ? H / 2 : H; d.Win = d.Hin; d.kh = d.kw = 3; d.stride = 1;
    d.style_stride = 0; d.flags = SPK_EPI_BIAS | SPK_EPI_NOISE | SPK_EPI_LRELU | SPK_EPI_STYLE | (ups ? SPK_CONV_UPSAMPLE2X : 0);
    d.lrelu_slope = 0.2f; d.out_scale = 1.f; d.ksplit = 1; d.act_gain = 1.f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if (run<C, 3, 3, 1, MODE>(&d, 0) != 0) { printf("%s: %s\n", name, spk::err_buf()); return 1; }
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) run<C, 3, 3, 1, MODE>(&d, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
#ifdef SPK_LAB_STEPTIME
    {   // time stamps after every k-step of chunk 10 (workgroup 8, its four waves)
        unsigned long long* dbg; CK(hipMalloc(&dbg, 4096)); CK(hipMemset(dbg, 0, 4096));
        d.stats = reinterpret_cast<double*>(dbg);
        run<C, 3, 3, 1, MODE>(&d, 0);
        CK(hipDeviceSynchronize());
        unsigned long long h[256]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w) {
            printf("   wave%d step durations:", w);
            for (int k = 1; k < 64 && h[w * 64 + k]; ++k) printf(" %lld", (long long)(h[w * 64 + k] - h[w * 64 + k - 1]));
            printf("\n");
        }
        d.stats = nullptr;
        CK(hipFree(dbg));
    }
#endif
    const double fl = 2.0 * 9 * Cin * Cout * (double)H * H * B;
    const Geometry g = geometry<C, 3, 3, 1>(B, Cin, Cout, H, H);
    printf("LAB=%d %-34s %8.1f us  %7.2f TFLOP/s   (WGs %d, LDS %zu B, tile %dx%dx%d)\n", SPK_LAB, name, ms * 1e3, fl / ms / 1e9,
           g.tiles_x * g.tiles_y * g.tiles_b * g.co_tiles, g.lds_bytes, g.TW, g.TH, g.TB);
    return 0;
}

int main() {
    const size_t n = (size_t)8 * 128 * 256 * 256;   // largest activation in the decoder
    float *x, *w, *y, *aux;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&w, (size_t)64 << 20)); CK(hipMalloc(&aux, (size_t)16 << 20));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(w, 0, (size_t)64 << 20)); CK(hipMemset(aux, 0, (size_t)16 << 20));
    int rc = 0;
    rc |= bench<Cfg4, MODE_PLAIN>("64^2 256->256 plain cfg4", 8, 256, 256, 64, x, w, y, aux);
    rc |= bench<Cfg4, MODE_UPSAMPLE>("64^2 512->256 ups cfg4", 8, 512, 256, 64, x, w, y, aux);
    rc |= bench<Cfg4, MODE_PLAIN>("128^2 128->128 plain cfg4", 8, 128, 128, 128, x, w, y, aux);
    rc |= bench<Cfg5, MODE_PLAIN>("256^2 64->64 plain cfg5", 8, 64, 64, 256, x, w, y, aux);
    rc |= bench<Cfg5, MODE_UPSAMPLE>("256^2 128->64 ups cfg5", 8, 128, 64, 256, x, w, y, aux);
    rc |= bench<Cfg4, MODE_PLAIN>("32^2 512->512 plain cfg4 ks1", 8, 512, 512, 32, x, w, y, aux);
    return rc;
}
