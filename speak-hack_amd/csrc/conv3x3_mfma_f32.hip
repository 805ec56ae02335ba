// 3x3 convolution (stride 1, pad 1) as an LDS-staged implicit GEMM on the gfx950 f32 MFMA pipe,
// with the StyleGAN layer epilogue (bias -> noise -> LeakyReLU -> style scale/shift) fused in and,
// optionally, the bilinear x2 upsampling of the input folded into the LDS staging.
//
// GEMM view (per image group):  D[co][pix] = sum_k A[co][k] * Bm[k][pix],  k = (tap, ci)
//   A  = weights, pre-packed [co_tile][ci_chunk][tap][ci][co]  (co contiguous -> conflict-free
//        ds_read_b32 of the A fragment: lane l reads A[i = l&31][k = l>>5])
//   Bm = input window; the tile's input planes (with 1-pixel halo) sit in LDS as
//        [ci][tb][TH+2][TW+2]; lane l reads Bm[k = l>>5][j = l&31] = 32 consecutive pixels
//   D  : v_mfma_f32_32x32x2_f32, col = lane&31 = pixel (x-contiguous -> 128-B coalesced NCHW
//        stores), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = output channel.
// One workgroup = WM x WN waves; each wave owns MT x NT accumulator tiles of 32co x 32pix.
// The f32 MFMA is bit-for-bit an fmaf chain (exact fp32), 64 cycles/SIMD per instruction, so the
// kernel is matrix-pipe bound as long as staging overlaps (>=2 workgroups per CU).
//
// Replaces the ATen call sites listed at spk_conv3x3_fwd in include/spk.h.
#include "spk_common.hpp"

#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvArgs {
    const float* x;
    const float* wp;
    const float* bias;
    const float* noise_w;
    const float* noise;
    const float* style;
    float* y;
    int B, Cin, Cout, H, W;  // output spatial size
    int Hs, Ws;              // source spatial size (H/2, W/2 when upsampling)
    int lgTW, lgTH, lgTB;    // log2 of the pixel-tile geometry
    int tiles_x, tiles_y;
    int n_chunks;
    int style_stride;
    unsigned flags;
    float slope, in_scale;
};

template <int WM_, int WN_, int MT_, int NT_, int CIT_>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_, CI_T = CIT_;
    static constexpr int NW = WM * WN, NTHREADS = NW * 64;
    static constexpr int CO_T = WM * MT * 32, PIX_T = WN * NT * 32;
    static constexpr int W_FLOATS = 9 * CI_T * CO_T;
    static constexpr int JMAX = PIX_T >= 256 ? 6 : (PIX_T >= 128 ? 4 : 3);
};

template <class C, bool UPS>
__global__ __launch_bounds__(C::NTHREADS) void conv3x3_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const w_s = smem;
    float* const in_s = smem + C::W_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave / C::WN, wn = wave % C::WN;

    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH, TB = 1 << p.lgTB;
    const int PW = TW + 2, PLANE = (TH + 2) * PW;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int tbi = bx / p.tiles_y;
    const int b0 = tbi << p.lgTB, y0 = ty << p.lgTH, x0 = tx << p.lgTW;
    const int co_tile0 = blockIdx.y * C::CO_T;

    // ---- per-lane gather descriptors of one (ci, tb) input plane: fixed for the whole kernel ----
    int g_off[C::JMAX];
    int g_dy[UPS ? C::JMAX : 1], g_dx[UPS ? C::JMAX : 1];
    float g_ly[UPS ? C::JMAX : 1], g_lx[UPS ? C::JMAX : 1];
    unsigned g_valid = 0;
#pragma unroll
    for (int j = 0; j < C::JMAX; ++j) {
        const int pidx = lane + 64 * j;
        g_off[j] = 0;
        if (UPS) { g_dy[j] = 0; g_dx[j] = 0; g_ly[j] = 0.f; g_lx[j] = 0.f; }
        if (pidx < PLANE) {
            const int r = pidx / PW, c = pidx - r * PW;
            const int uy = y0 + r - 1, ux = x0 + c - 1;
            const bool v = uy >= 0 && uy < p.H && ux >= 0 && ux < p.W;
            if (v) {
                g_valid |= 1u << j;
                if (!UPS) {
                    g_off[j] = uy * p.W + ux;
                } else {
                    // torch area_pixel_compute_source_index(scale=0.5, align_corners=False)
                    const float sy = fmaxf(0.5f * (uy + 0.5f) - 0.5f, 0.f);
                    const float sx = fmaxf(0.5f * (ux + 0.5f) - 0.5f, 0.f);
                    const int iy0 = (int)sy, ix0 = (int)sx;
                    const int iy1 = min(iy0 + 1, p.Hs - 1), ix1 = min(ix0 + 1, p.Ws - 1);
                    g_off[j] = iy0 * p.Ws + ix0;
                    g_dy[j] = (iy1 - iy0) * p.Ws;
                    g_dx[j] = ix1 - ix0;
                    g_ly[j] = sy - (float)iy0;
                    g_lx[j] = sx - (float)ix0;
                }
            }
        }
    }

    // ---- per-lane fragment addresses ----
    int b_off[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int pt = (wn * C::NT + n) * 32 + l32;
        const int px = pt & (TW - 1), py = (pt >> p.lgTW) & (TH - 1), tb = pt >> (p.lgTW + p.lgTH);
        b_off[n] = half * TB * PLANE + tb * PLANE + py * PW + px;
    }
    const int a_off = half * C::CO_T + wm * C::MT * 32 + l32;
    const int ci_stride2 = 2 * TB * PLANE;

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < C::NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const size_t src_plane = (size_t)p.Hs * p.Ws;
    const float4* wsrc = reinterpret_cast<const float4*>(p.wp) + (size_t)blockIdx.y * p.n_chunks * (C::W_FLOATS / 4);

    for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
        __syncthreads();  // everyone is done reading the previous chunk
        // weights: one contiguous 9*CI_T*CO_T block per (co_tile, chunk)
        for (int i = tid; i < C::W_FLOATS / 4; i += C::NTHREADS) reinterpret_cast<float4*>(w_s)[i] = wsrc[i];
        wsrc += C::W_FLOATS / 4;
        // input planes (zero-filled halo / out-of-range channels and images)
        for (int q = wave; q < C::CI_T * TB; q += C::NW) {
            const int ci = q >> p.lgTB, tb = q & (TB - 1);
            const int cig = chunk * C::CI_T + ci, b = b0 + tb;
            const bool pv = (cig < p.Cin) && (b < p.B);
            const float* src = p.x + ((size_t)(pv ? b : 0) * p.Cin + (pv ? cig : 0)) * src_plane;
            float* dst = in_s + q * PLANE;
#pragma unroll
            for (int j = 0; j < C::JMAX; ++j) {
                const int pidx = lane + 64 * j;
                if (pidx < PLANE) {
                    float v = 0.f;
                    if (pv && ((g_valid >> j) & 1u)) {
                        if (!UPS) {
                            v = src[g_off[j]];
                        } else {
                            const float* s0 = src + g_off[j];
                            const float v00 = s0[0], v01 = s0[g_dx[j]];
                            const float v10 = s0[g_dy[j]], v11 = s0[g_dy[j] + g_dx[j]];
                            const float lx1 = g_lx[j], lx0 = 1.f - lx1, ly1 = g_ly[j], ly0 = 1.f - ly1;
                            v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
                        }
                    }
                    dst[pidx] = v;
                }
            }
        }
        __syncthreads();

#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int tapoff = (tap / 3) * PW + (tap % 3);
#pragma unroll
            for (int kk = 0; kk < C::CI_T / 2; ++kk) {
                float a[C::MT], bv[C::NT];
#pragma unroll
                for (int m = 0; m < C::MT; ++m) a[m] = w_s[a_off + (tap * C::CI_T + 2 * kk) * C::CO_T + m * 32];
#pragma unroll
                for (int n = 0; n < C::NT; ++n) bv[n] = in_s[b_off[n] + kk * ci_stride2 + tapoff];
#pragma unroll
                for (int m = 0; m < C::MT; ++m)
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
    }

    // ---- fused epilogue: *in_scale -> +bias -> +noise_w*noise -> lrelu -> *(s0+1)+s1 -> store ----
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_noise = p.flags & SPK_EPI_NOISE;
    const bool f_lrelu = p.flags & SPK_EPI_LRELU, f_style = p.flags & SPK_EPI_STYLE;
    const bool f_accum = p.flags & SPK_EPI_ACCUM;
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int pt = (wn * C::NT + n) * 32 + l32;
        const int px = pt & (TW - 1), py = (pt >> p.lgTW) & (TH - 1), tb = pt >> (p.lgTW + p.lgTH);
        const int b = b0 + tb, yy = y0 + py, xx = x0 + px;
        const bool pvalid = b < p.B && yy < p.H && xx < p.W;
        if (!pvalid) continue;
        const size_t pix = (size_t)yy * p.W + xx;
        const float nz = f_noise ? p.noise[(size_t)b * p.H * p.W + pix] : 0.f;
        const float* st = f_style ? p.style + (size_t)b * p.style_stride : nullptr;
        float* yb = p.y + (size_t)b * p.Cout * p.H * p.W + pix;
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_tile0 + (wm * C::MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < p.Cout) {
                    float v = acc[m][n][r] * p.in_scale;
                    if (f_bias) v += p.bias[co];
                    if (f_noise) v += p.noise_w[co] * nz;
                    if (f_lrelu) v = v > 0.f ? v : v * p.slope;
                    if (f_style) v = v * (st[co] + 1.f) + st[p.Cout + co];
                    float* dst = yb + (size_t)co * p.H * p.W;
                    if (f_accum) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin_orig, int Cout_orig,
                                    int opCin, int opCout, int CO_T, int CI_T, int n_chunks, int tf, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    long long t = idx;
    const int co_in = (int)(t % CO_T); t /= CO_T;
    const int ci = (int)(t % CI_T); t /= CI_T;
    const int tap = (int)(t % 9); t /= 9;
    const int chunk = (int)(t % n_chunks);
    const int cot = (int)(t / n_chunks);
    const int co = cot * CO_T + co_in, cig = chunk * CI_T + ci;
    float v = 0.f;
    if (co < opCout && cig < opCin) {
        if (!tf) v = w[((size_t)co * Cin_orig + cig) * 9 + tap];
        else     v = w[((size_t)cig * Cin_orig + co) * 9 + (8 - tap)];
    }
    wp[idx] = v;
}

// ---- tile configs -------------------------------------------------------------------------------
//                WM WN MT NT CI_T       CO_T  PIX_T
typedef Cfg<2, 2, 2, 2, 8> Cfg0;  //  128   128   large layers
typedef Cfg<1, 4, 2, 2, 8> Cfg1;  //   64   256   Cout <= 64, many pixels
typedef Cfg<2, 2, 1, 1, 8> Cfg2;  //   64    64   small layers (more workgroups)
typedef Cfg<1, 4, 1, 1, 8> Cfg3;  //   32   128   Cout <= 32
constexpr int kNumConfigs = 4;

template <class C>
int run(const spk_conv3x3_desc* d, hipStream_t stream) {
    const bool ups = d->flags & SPK_CONV_UPSAMPLE2X;
    ConvArgs a;
    a.x = d->x; a.wp = d->w_packed; a.bias = d->bias; a.noise_w = d->noise_w; a.noise = d->noise;
    a.style = d->style; a.y = d->y;
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W;
    if (ups) {
        SPK_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0, "conv3x3: upsampled output size must be even (%dx%d)", d->H, d->W);
        a.Hs = d->H / 2; a.Ws = d->W / 2;
    } else {
        a.Hs = d->H; a.Ws = d->W;
    }
    int TW = std::min(32, spk::pow2_ceil(d->W));
    int TH = std::min(C::PIX_T / TW, spk::pow2_ceil(d->H));
    while ((TH + 2) * (TW + 2) > 64 * C::JMAX && TH > 1) TH >>= 1;
    const int TB = C::PIX_T / (TW * TH);
    const int PLANE = (TH + 2) * (TW + 2);
    SPK_REQUIRE(PLANE <= 64 * C::JMAX, "conv3x3: tile plane %d too large", PLANE);
    a.lgTW = spk::ilog2(TW); a.lgTH = spk::ilog2(TH); a.lgTB = spk::ilog2(TB);
    a.tiles_x = spk::ceil_div(d->W, TW); a.tiles_y = spk::ceil_div(d->H, TH);
    const int tiles_b = spk::ceil_div(d->B, TB);
    a.n_chunks = spk::ceil_div(d->Cin, C::CI_T);
    a.style_stride = d->style_stride; a.flags = d->flags; a.slope = d->lrelu_slope; a.in_scale = d->in_scale;
    const size_t lds = (size_t)(C::W_FLOATS + C::CI_T * TB * PLANE) * sizeof(float);
    SPK_REQUIRE(lds <= 64 * 1024, "conv3x3: LDS request %zu > 64 KiB", lds);
    const long long gx = (long long)a.tiles_x * a.tiles_y * tiles_b;
    SPK_REQUIRE(gx < (1ll << 31), "conv3x3: grid too large");
    dim3 grid((unsigned)gx, (unsigned)spk::ceil_div(d->Cout, C::CO_T));
    if (ups) hipLaunchKernelGGL((conv3x3_kernel<C, true>), grid, dim3(C::NTHREADS), lds, stream, a);
    else     hipLaunchKernelGGL((conv3x3_kernel<C, false>), grid, dim3(C::NTHREADS), lds, stream, a);
    return spk::check_launch("conv3x3_kernel");
}

void config_dims(int cfg, int* co_t, int* ci_t, int* pix_t) {
    int co = 0, ci = 0, px = 0;
    switch (cfg) {
        case 0: co = Cfg0::CO_T; ci = Cfg0::CI_T; px = Cfg0::PIX_T; break;
        case 1: co = Cfg1::CO_T; ci = Cfg1::CI_T; px = Cfg1::PIX_T; break;
        case 2: co = Cfg2::CO_T; ci = Cfg2::CI_T; px = Cfg2::PIX_T; break;
        case 3: co = Cfg3::CO_T; ci = Cfg3::CI_T; px = Cfg3::PIX_T; break;
    }
    if (co_t) *co_t = co;
    if (ci_t) *ci_t = ci;
    if (pix_t) *pix_t = px;
}

}  // namespace

extern "C" {

int spk_conv3x3_num_configs(void) { return kNumConfigs; }

int spk_conv3x3_pick_config(int B, int Cin, int Cout, int H, int W) {
    (void)Cin;
    const long long pixels = (long long)B * H * W;
    if (Cout <= 32) return 3;
    if (Cout <= 64) return pixels >= 256ll * 1024 ? 1 : 2;
    const long long wgs0 = (long long)spk::ceil_div(Cout, 128) * ((pixels + 127) / 128);
    return wgs0 >= 512 ? 0 : 2;
}

int spk_conv3x3_config_info(int config, int* co_tile, int* ci_tile, int* pix_tile) {
    SPK_REQUIRE(config >= 0 && config < kNumConfigs, "conv3x3: bad config %d", config);
    config_dims(config, co_tile, ci_tile, pix_tile);
    return SPK_OK;
}

int64_t spk_conv3x3_packed_floats(int config, int Cin, int Cout) {
    if (config < 0 || config >= kNumConfigs || Cin <= 0 || Cout <= 0) return -1;
    int co_t, ci_t;
    config_dims(config, &co_t, &ci_t, nullptr);
    return (int64_t)spk::ceil_div(Cout, co_t) * spk::ceil_div(Cin, ci_t) * 9 * ci_t * co_t;
}

int spk_conv3x3_pack_weights(const float* w, float* w_packed, int Cin, int Cout, int config, int transpose_flip,
                             void* stream) {
    SPK_REQUIRE(w && w_packed, "pack_weights: null pointer");
    SPK_REQUIRE(config >= 0 && config < kNumConfigs, "pack_weights: bad config %d", config);
    SPK_REQUIRE(Cin > 0 && Cout > 0, "pack_weights: bad channels");
    int co_t, ci_t;
    config_dims(config, &co_t, &ci_t, nullptr);
    const int opCin = transpose_flip ? Cout : Cin, opCout = transpose_flip ? Cin : Cout;
    const int n_chunks = spk::ceil_div(opCin, ci_t);
    const long long total = (long long)spk::ceil_div(opCout, co_t) * n_chunks * 9 * ci_t * co_t;
    const int threads = 256;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0,
                       (hipStream_t)stream, w, w_packed, Cin, Cout, opCin, opCout, co_t, ci_t, n_chunks,
                       transpose_flip ? 1 : 0, total);
    return spk::check_launch("pack_weights_kernel");
}

int spk_conv3x3_fwd(const spk_conv3x3_desc* d, void* stream) {
    SPK_REQUIRE(d, "conv3x3: null descriptor");
    SPK_REQUIRE(d->x && d->w_packed && d->y, "conv3x3: null tensor pointer");
    SPK_REQUIRE(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->H > 0 && d->W > 0, "conv3x3: bad shape");
    SPK_REQUIRE(!(d->flags & SPK_EPI_BIAS) || d->bias, "conv3x3: SPK_EPI_BIAS without bias");
    SPK_REQUIRE(!(d->flags & SPK_EPI_NOISE) || (d->noise && d->noise_w), "conv3x3: SPK_EPI_NOISE without noise");
    SPK_REQUIRE(!(d->flags & SPK_EPI_STYLE) || d->style, "conv3x3: SPK_EPI_STYLE without style");
    SPK_REQUIRE((long long)d->B * d->Cout * d->H * d->W < (1ll << 40), "conv3x3: tensor too large");
    int cfg = d->config;
    if (cfg < 0) cfg = spk_conv3x3_pick_config(d->B, d->Cin, d->Cout, d->H, d->W);
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
        case 0: return run<Cfg0>(d, s);
        case 1: return run<Cfg1>(d, s);
        case 2: return run<Cfg2>(d, s);
        case 3: return run<Cfg3>(d, s);
        default: return spk::fail(SPK_EINVAL, "conv3x3: bad config %d", cfg);
    }
}

}  // extern "C"
