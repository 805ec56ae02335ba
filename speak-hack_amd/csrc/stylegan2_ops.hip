// StyleGAN2-specific pieces of the build-defined decoder variant: the demodulation
// coefficients of a modulated convolution, upfirdn2d, and the modulated 1x1 toRGB.  The modulated 3x3 conv
// itself is the MFMA conv kernel with the modulation applied to the *input* in staging and the demodulation
// to the *output* in the epilogue, so the weights stay shared across the batch (no per-sample weight tensor).
#include "spk_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

// d[b,co] = rsqrt(scale^2 * sum_ci s[b,ci]^2 * (sum_k w[co,ci,k]^2) + eps): one wave per output channel,
// lanes over ci (weight row read once), batch looped in registers.
__global__ __launch_bounds__(256) void demod_kernel(const float* __restrict__ w, const float* __restrict__ s, float* __restrict__ d,
                                                   int B, int Cin, int Cout, int taps, float scale2, float eps) {
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (co >= Cout) return;
    const float* wr = w + (size_t)co * Cin * taps;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ci = lane; ci < Cin; ci += 64) {
            float wsq = 0.f;
            for (int k = 0; k < taps; ++k) { const float v = wr[(size_t)ci * taps + k]; wsq += v * v; }
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (b0 + b < B) { const float sv = s[(size_t)(b0 + b) * Cin + ci]; acc[b] += sv * sv * wsq; }
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[b] += __shfl_xor(acc[b], off);
            if (lane == 0 && b0 + b < B) d[(size_t)(b0 + b) * Cout + co] = rsqrtf(acc[b] * scale2 + eps);
        }
    }
}

// the demodulation vectors of several layers in one launch (they all depend only on the step's modulations)
struct DemodGroups {
    spk_demod_group g[SPK_DEMOD_MAX_GROUPS];
    int row_start[SPK_DEMOD_MAX_GROUPS + 1];   // in units of 4-output-channel blocks
    int n;
};

__global__ __launch_bounds__(256) void demod_grouped_kernel(const DemodGroups a, int B, float eps) {
    int gi = 0;
    while (gi + 1 < a.n && (int)blockIdx.x >= a.row_start[gi + 1]) ++gi;
    const spk_demod_group& q = a.g[gi];
    const int co = ((int)blockIdx.x - a.row_start[gi]) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (co >= q.Cout) return;
    const float* wr = q.w + (size_t)co * q.Cin * q.taps;
    const float scale2 = q.scale * q.scale;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ci = lane; ci < q.Cin; ci += 64) {
            float wsq = 0.f;
            for (int k = 0; k < q.taps; ++k) { const float v = wr[(size_t)ci * q.taps + k]; wsq += v * v; }
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (b0 + b < B) { const float sv = q.s[(size_t)(b0 + b) * q.Cin + ci]; acc[b] += sv * sv * wsq; }
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[b] += __shfl_xor(acc[b], off);
            if (lane == 0 && b0 + b < B) q.d[(size_t)(b0 + b) * q.Cout + co] = rsqrtf(acc[b] * scale2 + eps);
        }
    }
}

struct Fir2 { float f[49]; int k; };

// out[oy,ox] = gain * sum_{ky,kx} f[k-1-ky][k-1-kx] * xu[oy*down + ky - pad0][ox*down + kx - pad0],
// xu = x with (up-1) zeros inserted after every sample (size H*up), zero outside.
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const float* __restrict__ x, float* __restrict__ y, Fir2 fir, long long planes,
                                                       int H, int W, int Ho, int Wo, int up, int down, int pad0, float gain) {
    const int k = fir.k;
    const long long total = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho);
        const long long pl = idx / ((long long)Wo * Ho);
        const float* xp = x + pl * H * W;
        float acc = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int uy = oy * down + ky - pad0;
            if (uy < 0 || uy >= H * up || uy % up) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ux = ox * down + kx - pad0;
                if (ux < 0 || ux >= W * up || ux % up) continue;
                acc += fir.f[(k - 1 - ky) * k + (k - 1 - kx)] * xp[(size_t)(uy / up) * W + ux / up];
            }
        }
        y[idx] = acc * gain;
    }
}

// Separable form (every FIR StyleGAN2 uses is an outer product, e.g. [1,3,3,1] x [1,3,3,1]): a wave owns 64 consecutive output
// columns of one output row.  For each tap ROW that lands on a real (non-inserted) input row the wave loads the row segment it
// needs once -- one or two coalesced dwords per lane -- and every lane picks its taps' operands out of its neighbours'
// registers with wave shuffles (ds_bpermute: the LDS crossbar, no memory traffic); up / down / k are compile-time, so the
// zero-insertion is a parity select and the floor divisions are shifts -- no integer division or modulo in any loop.
struct Fir1 { float fx[8], fy[8]; };      // FLIPPED 1-D factors (true convolution), fx * fy^T = the 2-D filter

template <int UP, int DOWN, int K>
__global__ __launch_bounds__(256) void upfirdn2d_sep_kernel(const float* __restrict__ x, float* __restrict__ y, Fir1 fir, int H, int W, int Ho,
                                                           int Wo, int pad0, float gain) {
    constexpr int LG = UP == 2 ? 1 : 0;                    // UP in {1, 2}
    constexpr int NLD = (64 * DOWN + K + UP - 1) / UP / 64 + 1;          // 64-column register segments a wave needs per input row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ox0 = blockIdx.x * 64, ox = ox0 + lane;
    const int oy = blockIdx.y * 4 + wave;
    if (oy >= Ho) return;                                  // whole wave
    const float* xp = x + (size_t)blockIdx.z * H * W;
    const int c0 = (ox0 * DOWN - pad0) >> LG;              // first input column any lane of this wave can touch (floor division)
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int uy = oy * DOWN + ky - pad0;
        if (uy < 0 || (uy & (UP - 1)) || (uy >> LG) >= H) continue;          // wave-uniform
        const float* row = xp + (size_t)(uy >> LG) * W;
        float r[NLD];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int c = c0 + lane + 64 * q;
            r[q] = (c >= 0 && c < W) ? row[c] : 0.f;
        }
        float hs = 0.f;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int u = ox * DOWN + kx - pad0;
            const int idx = (u >> LG) - c0;                // 0 <= idx < 64 NLD
            float v = __shfl(r[0], idx & 63);
#pragma unroll
            for (int q = 1; q < NLD; ++q) {
                const float w = __shfl(r[q], idx & 63);
                v = (idx >> 6) == q ? w : v;
            }
            hs += (u & (UP - 1)) ? 0.f : fir.fx[kx] * v;   // an inserted zero column
        }
        acc += fir.fy[ky] * hs;
    }
    if (ox < Wo) y[((size_t)blockIdx.z * Ho + oy) * Wo + ox] = acc * gain;
}

template <int UP, int DOWN>
static bool launch_sep(int k, const float* x, float* y, const Fir1& fir, long long planes, int H, int W, int Ho, int Wo, int pad0,
                       float gain, hipStream_t s) {
    dim3 grid((unsigned)((Wo + 63) / 64), (unsigned)((Ho + 3) / 4), (unsigned)planes);
    if (k == 4) hipLaunchKernelGGL((upfirdn2d_sep_kernel<UP, DOWN, 4>), grid, dim3(256), 0, s, x, y, fir, H, W, Ho, Wo, pad0, gain);
    else if (k == 3) hipLaunchKernelGGL((upfirdn2d_sep_kernel<UP, DOWN, 3>), grid, dim3(256), 0, s, x, y, fir, H, W, Ho, Wo, pad0, gain);
    else if (k == 2) hipLaunchKernelGGL((upfirdn2d_sep_kernel<UP, DOWN, 2>), grid, dim3(256), 0, s, x, y, fir, H, W, Ho, Wo, pad0, gain);
    else return false;
    return true;
}

// f2d[i][j] == fy[i] * fx[j]?  (rank one; the factors of the FLIPPED filter are the flipped factors)
static bool separate(const float* f, int k, float* fx, float* fy) {
    int pi = -1, pj = -1;
    for (int i = 0; i < k && pi < 0; ++i)
        for (int j = 0; j < k; ++j)
            if (f[i * k + j] != 0.f) { pi = i; pj = j; break; }
    if (pi < 0) return false;
    const float piv = f[pi * k + pj];
    float scale = 0.f;
    for (int j = 0; j < k; ++j) { fx[j] = f[pi * k + j]; scale = std::max(scale, std::fabs(fx[j])); }
    for (int i = 0; i < k; ++i) fy[i] = f[i * k + pj] / piv;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
            if (std::fabs(f[i * k + j] - fy[i] * fx[j]) > 1e-6f * scale) return false;
    return true;
}

inline unsigned sgrid2(long long n) { return (unsigned)std::max(1ll, std::min((n + 255) / 256, 256ll * 16)); }

}  // namespace

extern "C" {

int spk_modconv_demod(const float* w, const float* s, float* d, int B, int Cin, int Cout, int taps, float scale, float eps,
                      void* stream) {
    SPK_REQUIRE(w && s && d && B > 0 && Cin > 0 && Cout > 0 && taps > 0, "modconv_demod: bad arguments");
    hipLaunchKernelGGL(demod_kernel, dim3((unsigned)spk::ceil_div(Cout, 4)), dim3(256), 0, (hipStream_t)stream, w, s, d, B, Cin, Cout,
                       taps, scale * scale, eps);
    return spk::check_launch("demod_kernel");
}

int spk_modconv_demod_grouped(const spk_demod_group* groups, int n_groups, int B, float eps, void* stream) {
    SPK_REQUIRE(groups && n_groups > 0 && n_groups <= SPK_DEMOD_MAX_GROUPS && B > 0, "modconv_demod_grouped: bad arguments (1..%d groups)",
                SPK_DEMOD_MAX_GROUPS);
    DemodGroups a;
    a.n = n_groups;
    a.row_start[0] = 0;
    for (int i = 0; i < n_groups; ++i) {
        const spk_demod_group& q = groups[i];
        SPK_REQUIRE(q.w && q.s && q.d && q.Cin > 0 && q.Cout > 0 && q.taps > 0, "modconv_demod_grouped: group %d: bad arguments", i);
        a.g[i] = q;
        a.row_start[i + 1] = a.row_start[i] + spk::ceil_div(q.Cout, 4);
    }
    hipLaunchKernelGGL(demod_grouped_kernel, dim3((unsigned)a.row_start[n_groups]), dim3(256), 0, (hipStream_t)stream, a, B, eps);
    return spk::check_launch("demod_grouped_kernel");
}

int spk_upfirdn2d_fwd(const float* x, float* y, const float* filter_host, int k, int64_t planes, int H, int W, int up, int down,
                      int pad0, int pad1, float gain, void* stream) {
    SPK_REQUIRE(x && y && filter_host && k >= 1 && k <= 7 && planes > 0 && H > 0 && W > 0 && up >= 1 && down >= 1,
                "upfirdn2d: bad arguments (k <= 7)");
    const int Ho = (H * up + pad0 + pad1 - k) / down + 1, Wo = (W * up + pad0 + pad1 - k) / down + 1;
    SPK_REQUIRE(Ho > 0 && Wo > 0, "upfirdn2d: empty output");
    static const bool allow_sep = [] { const char* e = getenv("SPK_UPFIRDN_SEP"); return !e || atoi(e) != 0; }();
    float fx[8], fy[8];
    if (allow_sep && k <= 4 && up <= 2 && down <= 2 && planes < 65536 && (Ho + 3) / 4 < 65536 && separate(filter_host, k, fx, fy)) {
        Fir1 f1;
        for (int i = 0; i < k; ++i) { f1.fx[i] = fx[k - 1 - i]; f1.fy[i] = fy[k - 1 - i]; }     // flipped: upfirdn2d is a true convolution
        bool ok;
        if (up == 1) ok = down == 1 ? launch_sep<1, 1>(k, x, y, f1, planes, H, W, Ho, Wo, pad0, gain, (hipStream_t)stream)
                                    : launch_sep<1, 2>(k, x, y, f1, planes, H, W, Ho, Wo, pad0, gain, (hipStream_t)stream);
        else ok = down == 1 ? launch_sep<2, 1>(k, x, y, f1, planes, H, W, Ho, Wo, pad0, gain, (hipStream_t)stream)
                            : launch_sep<2, 2>(k, x, y, f1, planes, H, W, Ho, Wo, pad0, gain, (hipStream_t)stream);
        if (ok) return spk::check_launch("upfirdn2d_sep_kernel");
    }
    Fir2 fir;
    fir.k = k;
    for (int i = 0; i < k * k; ++i) fir.f[i] = filter_host[i];
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3(sgrid2((long long)planes * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, y, fir,
                       (long long)planes, H, W, Ho, Wo, up, down, pad0, gain);
    return spk::check_launch("upfirdn2d_kernel");
}

}  // extern "C"
