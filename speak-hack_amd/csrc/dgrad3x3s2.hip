// Data gradient of a 3x3 stride-2 pad-1 convolution with EXACTLY its taps (tile config 13 of SPK_CONV_DGRAD_S2):
//     dx[b, c, Y, X] = sum_k sum_{ky,kx : 2m+ky-1 = Y, 2n+kx-1 = X} W[k][c][ky][kx] * g[b, k, m, n]
// An input pixel (2m+py, 2n+px) is reached through 1 (even, even), 2, 2 or 4 (odd, odd) of the 9 taps.  The one-launch form of
// conv_inst_2x2.hip runs the four parity classes as four zero-padded 2x2 kernels -- 16 taps executed where 9 are needed -- and a
// launch per class with its own tap count (tried: 1x1 / 1x2 / 2x1 / 2x2 kernels) loses what it saves to stride-2 scattered
// stores: 32-byte sectors written half at a time by two launches (31 TFLOP/s for the 1-tap class).  Here ONE wave owns all four
// classes of its 32 channels x 64 gradient pixels: per pair of contraction channels it reads the 9 weight fragments (the 3x3
// kernel itself) and the 4 window positions g[m+a, n+b] and issues the 9 MFMAs, tap (ky, kx) feeding the accumulator of class
// (ky != 1, kx != 1) from window position (ky == 0, kx == 0).  The epilogue pairs the px = 0 / 1 classes: a lane stores two
// adjacent pixels, 16 lanes a full 128-byte line.
//   block  : 64 channels (the conv's INPUT channels) x 128 gradient pixels (16 x 8), 4 waves = 2 (channel halves) x 2 (pixel halves)
//   chunk  : 8 contraction channels (the conv's OUTPUT channels) = 4 k-steps of v_mfma_f32_32x32x2_f32, 72 MFMAs per wave
//   weights: packed [co tile][chunk][tap][8][64] (the layout of tile config 2 with transpose_flip = 1), a chunk = 18 KB
//            contiguous, copied by LDS-DMA one chunk ahead; the gradient tile (8 x 9 x 17 with the right / bottom halo) through
//            registers one chunk ahead; two stages of 23.3 KB, two workgroups per CU
// replaces: the input-gradient half of F.conv2d's backward for the stride-2 3x3 convs on the path -- conv2 of the first
// bottleneck of layer2-4 of the torchvision trunk (model.py:60-62), conv2 of every StyleDiscriminator block
// (styleganv1.py:644-657).
#include "conv_mfma_f32.hpp"      // (ConvArgs + launch_splitk_epilogue for the sliced form)

#include <algorithm>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace spkdg {

constexpr int CO_T = 64, CI_T = 8, TW = 16, TH = 8, PIX_T = TW * TH;
constexpr int PW = TW + 1, PH = TH + 1, PLANE = PH * PW;            // 153 (odd: the row pitch of a channel's plane)
constexpr int W_FLOATS = 9 * CI_T * CO_T;                           // 4608 = 18 KB
constexpr int G_FLOATS = CI_T * PLANE;                              // 1224
constexpr int STAGE = (W_FLOATS + G_FLOATS + 1 + 3) & ~3;           // floats (+ a dump slot for the idle lanes of the last round)
constexpr int GROUNDS = (G_FLOATS + 255) / 256;                     // gradient elements a thread stages per chunk (5)
constexpr int W_BLOCKS = W_FLOATS * 4 / 1024;                       // 1 KB LDS-DMA blocks per chunk (18)

struct Args {
    const float* g;          // [B, G*K, Hg, Wg]   output-side gradient
    const float* wp;         // packed weights, per group [co tile][chunk][tap][8][64]
    float* y;                // [B, G*Cc, Hd, Wd]  input-side gradient
    int B, K, Cc, Hg, Wg, Hd, Wd;      // K / Cc per group
    int G, gin, Cg, Cy, co_tiles_g, n_chunks;
    int cps;                 // chunks per contraction slice (blockIdx.z): slice z writes its partial sums at y + z * slice_floats
    size_t slice_floats;
    int tiles_x, tiles_y;
    int accumulate;
    float out_scale;
    const float* out_scale_dev;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__global__ __launch_bounds__(256, 2) void dgrad3x3s2_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wc = wave & 1, wpix = wave >> 1;

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x; bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int grp = (int)blockIdx.y / p.co_tiles_g;
    const int co0 = ((int)blockIdx.y - grp * p.co_tiles_g) * CO_T;          // within the group
    const size_t HWg = (size_t)p.Hg * p.Wg;
    const int c_begin = (int)blockIdx.z * p.cps, c_end = min(c_begin + p.cps, p.n_chunks);      // (host: c_begin < n_chunks)

    // ---- gradient-tile staging roles: element e = tid + 256 i of the [8][9][17] tile (LDS offset = e) ----
    int g_off[GROUNDS], g_k[GROUNDS];
    bool g_ok[GROUNDS];
#pragma unroll
    for (int i = 0; i < GROUNDS; ++i) {
        const int e = tid + 256 * i;
        const int k = e / PLANE, pos = e - k * PLANE, r = pos / PW, c = pos - r * PW;
        g_ok[i] = e < G_FLOATS && y0 + r < p.Hg && x0 + c < p.Wg;
        g_k[i] = k;
        g_off[i] = g_ok[i] ? (k * p.Hg + r) * p.Wg + c : 0;
    }
    const float* gbase = p.g + ((size_t)b * p.Cg + (size_t)grp * p.gin) * HWg + (size_t)y0 * p.Wg + x0;
    const char* wsrc = reinterpret_cast<const char*>(p.wp) + (size_t)blockIdx.y * p.n_chunks * (W_FLOATS * 4);

    float gq[GROUNDS];
    auto g_issue = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < GROUNDS; ++i) {
            const bool ok = g_ok[i] && chunk * CI_T + g_k[i] < p.K;
            gq[i] = gbase[ok ? (size_t)chunk * CI_T * HWg + g_off[i] : 0];      // masked at store time
        }
    };
    auto g_store = [&](float* stage, int chunk) {
#pragma unroll
        for (int i = 0; i < GROUNDS; ++i) {
            const int e = tid + 256 * i;
            const bool ok = g_ok[i] && chunk * CI_T + g_k[i] < p.K;
            stage[W_FLOATS + (e < G_FLOATS ? e : G_FLOATS)] = ok ? gq[i] : 0.f;
        }
    };
    auto w_dma = [&](float* stage, int chunk) {
        const char* src = wsrc + (size_t)chunk * (W_FLOATS * 4);
#pragma unroll
        for (int i = 0; i < (W_BLOCKS + 3) / 4; ++i) {
            const int blk = i * 4 + wave;                                       // uniform
            if (blk < W_BLOCKS)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(src + blk * 1024) + lane,
                                                 reinterpret_cast<char*>(stage) + blk * 1024, 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][n][r] = 0.f;

    // ---- prologue: the slice's first chunk -> stage 0 ----
    w_dma(smem, c_begin);
    g_issue(c_begin);
    g_store(smem, c_begin);
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): this wave's LDS-DMA blocks have landed
    __syncthreads();

    // fragment addressing (floats, relative to a stage)
    const int a_base = half * CO_T + wc * 32 + l32;                                                      // + tap' * 512 + 2 s * 64
    int b_base[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) b_base[n] = W_FLOATS + half * PLANE + (2 * (2 * wpix + n) + (l32 >> 4)) * PW + (l32 & 15);   // + 2 s PLANE + a PW + b

    for (int i = c_begin; i < c_end; ++i) {
        const float* cur = smem + ((i - c_begin) & 1) * STAGE;
        float* nxt = smem + ((i - c_begin + 1) & 1) * STAGE;
        const bool more = i + 1 < c_end;                                        // uniform
        if (more) {
            w_dma(nxt, i + 1);
            g_issue(i + 1);
        }
        float wa[2][9], gb[2][2][2][2];                                          // [slot][tap] / [slot][a][b][n]
#define SPK_DG_FRAG(s_, slot_)                                                                               \
    {                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) wa[slot_][t] = cur[a_base + (8 - t) * (CI_T * CO_T) + 2 * (s_) * CO_T]; \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                        \
            _Pragma("unroll") for (int bb = 0; bb < 2; ++bb)                                                 \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
                    gb[slot_][a][bb][n] = cur[b_base[n] + 2 * (s_) * PLANE + a * PW + bb];                   \
    }
        SPK_DG_FRAG(0, 0);
        static_for<0, CI_T / 2>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            if constexpr (s + 1 < CI_T / 2) SPK_DG_FRAG(s + 1, (s + 1) & 1);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int q = 2 * (ky != 1) + (kx != 1), a = ky == 0, bb = kx == 0;
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[q][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s & 1][ky * 3 + kx], gb[s & 1][a][bb][n], acc[q][n], 0, 0, 0);
                }
        });
#undef SPK_DG_FRAG
        if (more) {
            __builtin_amdgcn_sched_barrier(0);             // the zero-select of the stores must not be hoisted up to the loads
            g_store(nxt, i + 1);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0): the next stage's weight image has landed
        __syncthreads();
    }

    // ---- epilogue: the px = 0 / 1 classes of a pixel are adjacent in memory: one 8-byte store per (channel, row parity) ----
    const float osc = p.out_scale_dev ? p.out_scale * *p.out_scale_dev : p.out_scale;   // (uniform)
    const bool pair_ok = (p.Wd & 1) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 7) == 0;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int py = 2 * (2 * wpix + n) + (l32 >> 4), px = l32 & 15;
        const int m = y0 + py, nn = x0 + px;
        if (m >= p.Hg || nn >= p.Wg) continue;
#pragma unroll
        for (int qy = 0; qy < 2; ++qy) {
            const int Y = 2 * m + qy, X = 2 * nn;
            if (Y >= p.Hd || X >= p.Wd) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = co0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c >= p.Cc) continue;
                float* dst = p.y + (size_t)blockIdx.z * p.slice_floats + (((size_t)b * p.Cy + (size_t)grp * p.Cc + c) * p.Hd + Y) * p.Wd + X;
                float v0 = acc[2 * qy][n][r] * osc, v1 = acc[2 * qy + 1][n][r] * osc;
                if (pair_ok) {                             // Wd even: X + 1 exists and the pair is 8-byte aligned
                    if (p.accumulate) {
                        const f32x2 old = *reinterpret_cast<const f32x2*>(dst);
                        v0 += old[0]; v1 += old[1];
                    }
                    *reinterpret_cast<f32x2*>(dst) = f32x2{v0, v1};
                } else {
                    dst[0] = p.accumulate ? dst[0] + v0 : v0;
                    if (X + 1 < p.Wd) dst[1] = p.accumulate ? dst[1] + v1 : v1;
                }
            }
        }
    }
}

}  // namespace spkdg

namespace spkconv {

// floats of the packed image (per group): [co tile 64][chunk 8][tap 9][8][64], i.e. tile config 2's layout of the transposed operator
long long dgrad_s2_fused_packed_floats(int K, int Cc) {
    return (long long)spk::ceil_div(Cc, spkdg::CO_T) * spk::ceil_div(K, spkdg::CI_T) * spkdg::W_FLOATS;
}

bool dgrad_s2_fused_takes(int B, int K, int Cc, int Hg, int Wg) {
    // 32-bit element offsets inside an image's channel group; everything else is masked
    return B > 0 && K > 0 && Cc > 0 && Hg > 0 && Wg > 0 && (long long)(spkdg::CI_T + 1) * Hg * Wg < (1ll << 31);
}

// d = the SPK_CONV_DGRAD_S2 descriptor: x = g [B, G*Cin, Hin, Win], y = dx [B, G*Cout, H, W]
int run_dgrad_s2_fused(const spk_conv2d_desc* d, hipStream_t stream) {
    using namespace spkdg;
    Args a;
    a.g = static_cast<const float*>(d->x); a.wp = static_cast<const float*>(d->w_packed); a.y = static_cast<float*>(d->y);
    a.B = d->B; a.K = d->Cin; a.Cc = d->Cout; a.Hg = d->Hin; a.Wg = d->Win; a.Hd = d->H; a.Wd = d->W;
    a.G = d->groups > 1 ? d->groups : 1;
    a.gin = a.G > 1 ? d->group_in_stride : d->Cin;
    a.Cg = a.gin * (a.G - 1) + d->Cin;
    a.Cy = a.G * d->Cout;
    a.co_tiles_g = spk::ceil_div(d->Cout, CO_T);
    a.n_chunks = spk::ceil_div(d->Cin, CI_T);
    a.tiles_x = spk::ceil_div(d->Win, TW); a.tiles_y = spk::ceil_div(d->Hin, TH);
    a.accumulate = (d->flags & SPK_EPI_ACCUM) ? 1 : 0;
    a.out_scale = d->out_scale; a.out_scale_dev = d->out_scale_dev;
    SPK_REQUIRE(dgrad_s2_fused_takes(d->B, d->Cin, d->Cout, d->Hin, d->Win), "conv2d: DGRAD_S2 (config 13): gradient plane too large");
    SPK_REQUIRE((reinterpret_cast<uintptr_t>(d->w_packed) & 15) == 0, "conv2d: DGRAD_S2 (config 13): w_packed must be 16-byte aligned");
    const long long gx = (long long)a.tiles_x * a.tiles_y * d->B;
    SPK_REQUIRE(gx < (1ll << 31) && a.G * a.co_tiles_g < 65536, "conv2d: grid too large");
    // few workgroups (a 16^2 gradient at B = 8: 16 pixel tiles x 8 channel tiles for 512 slots of two per CU): the contraction runs in
    // slices, partial sums through the caller's split-K workspace, the direct kernels' finisher adds them up (scale, accumulate)
    const int ks = dgrad_s2_ksplit(d->B, d->Cin, d->Cout, d->Hin, d->Win, a.G);
    const size_t out_floats = (size_t)d->B * a.Cy * d->H * d->W;
    const bool split = ks > 1 && d->ksplit != 1 && d->workspace && (reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0 &&
                       d->workspace_bytes >= (int64_t)ks * (int64_t)out_floats * 4;
    a.cps = split ? spk::ceil_div(a.n_chunks, ks) : a.n_chunks;
    a.slice_floats = split ? out_floats : 0;
    const int nz = split ? spk::ceil_div(a.n_chunks, a.cps) : 1;
    if (split) { a.y = static_cast<float*>(d->workspace); a.accumulate = 0; a.out_scale = 1.f; a.out_scale_dev = nullptr; }
    dim3 grid((unsigned)gx, (unsigned)(a.G * a.co_tiles_g), (unsigned)nz);
    hipLaunchKernelGGL(dgrad3x3s2_kernel, grid, dim3(256), 2 * STAGE * sizeof(float), stream, a);
    int rc = spk::check_launch("dgrad3x3s2_kernel");
    if (rc != SPK_OK || !split) return rc;
    ConvArgs f = {};
    f.y = static_cast<float*>(d->y); f.B = d->B; f.Cin = d->Cin; f.Cout = d->Cout; f.Cy = a.Cy; f.Cx = a.Cg; f.G = a.G; f.H = d->H; f.W = d->W;
    f.flags = d->flags & SPK_EPI_ACCUM; f.slope = 1.f; f.out_scale = d->out_scale; f.act_gain = 1.f; f.out_scale_dev = d->out_scale_dev;
    return launch_splitk_epilogue(f, static_cast<const float*>(d->workspace), nz, stream);
}

// slices of the contraction for the exact-tap kernel: 1 unless the grid fills less than half of the 512 workgroup slots
int dgrad_s2_ksplit(int B, int K, int Cc, int Hg, int Wg, int G) {
    const long long wgs = (long long)spk::ceil_div(Wg, spkdg::TW) * spk::ceil_div(Hg, spkdg::TH) * B * G * spk::ceil_div(Cc, spkdg::CO_T);
    const int n_chunks = spk::ceil_div(K, spkdg::CI_T);
    int ks = 1;
    while (wgs * ks < 256 && n_chunks / (2 * ks) >= 8) ks *= 2;
    return ks;
}

}  // namespace spkconv
