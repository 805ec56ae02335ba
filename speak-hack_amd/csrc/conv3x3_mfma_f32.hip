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
//
// Pipeline: LDS is double buffered.  While the waves run the 9*CI_T/2 MFMA k-steps of chunk c out
// of buffer c&1, the global loads of chunk c+1 (weights: 16-B loads of one contiguous packed
// block; input: per-lane gathers whose addresses, validity and bilinear codes were computed once
// at kernel start) are already in flight into registers; they are written to buffer (c+1)&1 after
// the k-steps, followed by the only barrier of the chunk.  The f32 MFMA is 64 cycles/SIMD per
// instruction, so all of the staging issue fits in its shadow.
//
// Split-K: gridDim.z slices the ci-chunk range; slices write raw partial sums to a workspace and
// splitk_epilogue_kernel reduces them in a fixed order (bitwise reproducible) and applies the
// epilogue.  Used for the low-resolution layers, whose output tiles alone cannot fill 256 CUs.
//
// Replaces the ATen call sites listed at spk_conv3x3_fwd in include/spk.h.
#include "spk_common.hpp"

#include <algorithm>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void spk_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        spk_static_for<I + 1, N>(f);
    }
}

struct ConvArgs {
    const float* x;
    const float* wp;
    const float* bias;
    const float* noise_w;
    const float* noise;
    const float* style;
    float* y;   // output, or the split-K workspace [ksplit][B][Cout][H][W]
    int B, Cin, Cout, H, W;  // output spatial size
    int Hs, Ws;              // source spatial size (H/2, W/2 when upsampling)
    int lgTW, lgTH, lgTB;    // log2 of the pixel-tile geometry
    int tiles_x, tiles_y;
    int n_chunks;            // ceil(Cin / CI_T)
    int chunks_per_split;
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
    static constexpr int WV = (W_FLOATS / 4 + NTHREADS - 1) / NTHREADS;  // float4 per thread per chunk
    // register slots (64 input elements each) a wave prefetches per chunk
    static constexpr int NSLOT_FULL = PIX_T >= 256 ? 11 : (PIX_T >= 128 ? 7 : 5);   // for CI_T = 8
    static constexpr int NSLOT = (NSLOT_FULL * CI_T + 7) / 8;
    // fragment prefetch distance in k-steps: about 256 MFMA cycles of cover for the ds_read latency
    static constexpr int PD = MT * NT >= 4 ? 1 : (MT * NT >= 2 ? 2 : 4);
};

// slot descriptor bits
constexpr unsigned D_VALID = 1u;        // element is inside the image (else: zero padding)
constexpr unsigned D_CI_SHIFT = 1;      // 4 bits: ci within the chunk
constexpr unsigned D_DX = 1u << 5;      // bilinear: second tap is one column to the right
constexpr unsigned D_DY = 1u << 6;      // bilinear: second row is one source row below
constexpr unsigned D_LX_SHIFT = 7;      // 2 bits: lambda code (0: 0, 1: 0.25, 2: 0.75)
constexpr unsigned D_LY_SHIFT = 9;

__device__ __forceinline__ float lambda_of(unsigned code) { return code == 0 ? 0.f : (code == 1 ? 0.25f : 0.75f); }

template <class C, bool UPS>
__global__ __launch_bounds__(C::NTHREADS) void conv3x3_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave / C::WN, wn = wave % C::WN;

    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH, TB = 1 << p.lgTB;
    const int PW = TW + 2, PLANE = (TH + 2) * PW;
    const int IN_FLOATS = C::CI_T * TB * PLANE;
    const int BUF_FLOATS = C::W_FLOATS + ((IN_FLOATS + 3) & ~3);
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int tbi = bx / p.tiles_y;
    const int b0 = tbi << p.lgTB, y0 = ty << p.lgTH, x0 = tx << p.lgTW;
    const int co_tile0 = blockIdx.y * C::CO_T;
    const int c_begin = blockIdx.z * p.chunks_per_split;
    const int c_end = min(p.n_chunks, c_begin + p.chunks_per_split);

    // ---- per-lane gather slots: wave w owns planes [w*ppw, (w+1)*ppw), contiguous in LDS ----
    const int ppw = (C::CI_T * TB) / C::NW;
    const int wave_elems = ppw * PLANE;
    const size_t src_plane = (size_t)p.Hs * p.Ws;
    int s_off[C::NSLOT];
    unsigned s_desc[C::NSLOT];
#pragma unroll
    for (int s = 0; s < C::NSLOT; ++s) {
        s_off[s] = 0;
        s_desc[s] = 0;
        const int e = s * 64 + lane;
        if (e < wave_elems) {
            const int pl = e / PLANE, pidx = e - pl * PLANE;
            const int q = wave * ppw + pl;
            const int ci = q >> p.lgTB, tb = q & (TB - 1);
            const int r = pidx / PW, c = pidx - r * PW;
            const int uy = y0 + r - 1, ux = x0 + c - 1;
            unsigned d = (unsigned)ci << D_CI_SHIFT;
            if (uy >= 0 && uy < p.H && ux >= 0 && ux < p.W && b0 + tb < p.B) {
                d |= D_VALID;
                int goff;
                if (!UPS) {
                    goff = uy * p.W + ux;
                } else {
                    // torch area_pixel_compute_source_index(scale=0.5, align_corners=False):
                    // src = max(0.5*(dst+0.5)-0.5, 0); lambdas are exactly 0, 0.25 or 0.75
                    const int iy0 = uy == 0 ? 0 : (uy - 1) >> 1, ix0 = ux == 0 ? 0 : (ux - 1) >> 1;
                    const unsigned ly = uy == 0 ? 0u : ((uy & 1) ? 1u : 2u);
                    const unsigned lx = ux == 0 ? 0u : ((ux & 1) ? 1u : 2u);
                    if (iy0 + 1 < p.Hs) d |= D_DY;
                    if (ix0 + 1 < p.Ws) d |= D_DX;
                    d |= (lx << D_LX_SHIFT) | (ly << D_LY_SHIFT);
                    goff = iy0 * p.Ws + ix0;
                }
                s_off[s] = (int)((size_t)(tb * p.Cin + ci) * src_plane) + goff;
            } else {
                s_off[s] = ci * (int)src_plane;  // masked at store time; any address inside the tensor will do
            }
            s_desc[s] = d;
        }
    }
    const float* xblk = p.x + (size_t)b0 * p.Cin * src_plane;

    // ---- per-lane fragment addresses ----
    int b_off[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int pt = (wn * C::NT + n) * 32 + l32;
        const int px = pt & (TW - 1), py = (pt >> p.lgTW) & (TH - 1);
        const int tb = min(pt >> (p.lgTW + p.lgTH), TB - 1);  // pixel groups beyond the tile idle (results dropped)
        b_off[n] = C::W_FLOATS + half * TB * PLANE + tb * PLANE + py * PW + px;
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

    const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wp) +
                         ((size_t)blockIdx.y * p.n_chunks + c_begin) * (C::W_FLOATS / 4);

    // prefetch registers
    f32x4 wreg[C::WV];
    float xin[UPS ? 4 * C::NSLOT : C::NSLOT];

    // staging is written as macros (not lambdas / functions) so that the prefetch arrays stay in
    // registers: every index is a compile-time constant after unrolling.
#define SPK_ISSUE_LOADS(chunk_)                                                                               \
    {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < C::WV; ++i) {                                                   \
            const int idx = tid + i * C::NTHREADS;                                                            \
            if (C::W_FLOATS / 4 % C::NTHREADS == 0 || idx < C::W_FLOATS / 4) wreg[i] = wsrc[idx];             \
        }                                                                                                     \
        wsrc += C::W_FLOATS / 4;                                                                              \
        const float* xc = xblk + (size_t)(chunk_) * C::CI_T * src_plane;                                      \
        const int ci_left_ = p.Cin - (chunk_) * C::CI_T;                                                      \
        _Pragma("unroll") for (int s = 0; s < C::NSLOT; ++s) {                                                \
            /* unconditional loads from always-valid addresses; masking happens at LDS-store time.            \
               Channels past Cin (zero-padded last chunk) are folded onto channel 0 of the chunk. */          \
            const unsigned d = s_desc[s];                                                                     \
            const int ci_ = (int)((d >> D_CI_SHIFT) & 15u);                                                   \
            const int off_ = s_off[s] - (ci_ >= ci_left_ ? ci_ * (int)src_plane : 0);                         \
            if (!UPS) {                                                                                       \
                xin[s] = xc[off_];                                                                            \
            } else {                                                                                          \
                const float* s0 = xc + off_;                                                                  \
                const int dx = (d & D_DX) ? 1 : 0, dy = (d & D_DY) ? p.Ws : 0;                                \
                xin[4 * s + 0] = s0[0];                                                                       \
                xin[4 * s + 1] = s0[dx];                                                                      \
                xin[4 * s + 2] = s0[dy];                                                                      \
                xin[4 * s + 3] = s0[dy + dx];                                                                 \
            }                                                                                                 \
        }                                                                                                     \
    }

#define SPK_STORE_LDS(buf_, chunk_)                                                                           \
    {                                                                                                         \
        float* const sbuf = (buf_);                                                                           \
        const int ci_left = p.Cin - (chunk_) * C::CI_T; /* channels of this chunk that exist */               \
        _Pragma("unroll") for (int i = 0; i < C::WV; ++i) {                                                   \
            const int idx = tid + i * C::NTHREADS;                                                            \
            if (C::W_FLOATS / 4 % C::NTHREADS == 0 || idx < C::W_FLOATS / 4)                                  \
                reinterpret_cast<f32x4*>(sbuf)[idx] = wreg[i];                                                \
        }                                                                                                     \
        float* dst = sbuf + C::W_FLOATS + wave * wave_elems;                                                  \
        _Pragma("unroll") for (int s = 0; s < C::NSLOT; ++s) {                                                \
            const int e = s * 64 + lane;                                                                      \
            if (e < wave_elems) {                                                                             \
                float v;                                                                                      \
                const unsigned d = s_desc[s];                                                                 \
                const bool ok = (d & D_VALID) && (int)((d >> D_CI_SHIFT) & 15u) < ci_left;                    \
                if (!UPS) {                                                                                   \
                    v = xin[s];                                                                               \
                } else {                                                                                      \
                    const float lx1 = lambda_of((d >> D_LX_SHIFT) & 3u), ly1 = lambda_of((d >> D_LY_SHIFT) & 3u); \
                    const float lx0 = 1.f - lx1, ly0 = 1.f - ly1;                                             \
                    v = ly0 * (lx0 * xin[4 * s] + lx1 * xin[4 * s + 1]) +                                     \
                        ly1 * (lx0 * xin[4 * s + 2] + lx1 * xin[4 * s + 3]);                                  \
                }                                                                                             \
                dst[e] = ok ? v : 0.f;                                                                        \
            }                                                                                                 \
        }                                                                                                     \
    }

    if (c_begin < c_end) {
        SPK_ISSUE_LOADS(c_begin);
        SPK_STORE_LDS(smem, c_begin);
    }
    __syncthreads();

    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const float* buf = smem + ((chunk - c_begin) & 1) * BUF_FLOATS;
        const bool more = chunk + 1 < c_end;
        if (more) SPK_ISSUE_LOADS(chunk + 1);

        // k-steps: step = (tap, kk).  Fragments are read PD steps ahead of the MFMAs that use them
        // (register ring, all indices static after unrolling) so LDS latency hides under the MFMAs;
        // sched_group_barrier pins the ds_read / MFMA interleave the source states.
        {
            constexpr int STEPS = 9 * (C::CI_T / 2);
            constexpr int PD = C::PD;
            float fa[PD + 1][C::MT], fb[PD + 1][C::NT];
#define SPK_LOAD_FRAG(step_)                                                                                  \
    {                                                                                                         \
        constexpr int tap_ = (step_) / (C::CI_T / 2), kk_ = (step_) % (C::CI_T / 2);                          \
        const int tapoff_ = (tap_ / 3) * PW + (tap_ % 3);                                                     \
        _Pragma("unroll") for (int m = 0; m < C::MT; ++m)                                                     \
            fa[(step_) % (PD + 1)][m] = buf[a_off + (tap_ * C::CI_T + 2 * kk_) * C::CO_T + m * 32];           \
        _Pragma("unroll") for (int n = 0; n < C::NT; ++n)                                                     \
            fb[(step_) % (PD + 1)][n] = buf[b_off[n] + kk_ * ci_stride2 + tapoff_];                           \
    }
            spk_static_for<0, PD>([&](auto i) { SPK_LOAD_FRAG(decltype(i)::value); });
            spk_static_for<0, STEPS>([&](auto i) {
                constexpr int st = decltype(i)::value;
                if constexpr (st + PD < STEPS) {
                    SPK_LOAD_FRAG(st + PD);
                    __builtin_amdgcn_sched_group_barrier(0x100, C::MT + C::NT, 0);
                }
#pragma unroll
                for (int m = 0; m < C::MT; ++m)
#pragma unroll
                    for (int n = 0; n < C::NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st % (PD + 1)][m], fb[st % (PD + 1)][n],
                                                                        acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, C::MT * C::NT, 0);
            });
#undef SPK_LOAD_FRAG
        }

        if (more) SPK_STORE_LDS(smem + (((chunk - c_begin) & 1) ^ 1) * BUF_FLOATS, chunk + 1);
        __syncthreads();
    }

#undef SPK_ISSUE_LOADS
#undef SPK_STORE_LDS

    // ---- epilogue ----
    const bool split = gridDim.z > 1;
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_noise = p.flags & SPK_EPI_NOISE;
    const bool f_lrelu = p.flags & SPK_EPI_LRELU, f_style = p.flags & SPK_EPI_STYLE;
    const bool f_accum = p.flags & SPK_EPI_ACCUM;
    const size_t HW = (size_t)p.H * p.W;
    float* ybase = p.y + (split ? (size_t)blockIdx.z * p.B * p.Cout * HW : 0);
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int pt = (wn * C::NT + n) * 32 + l32;
        const int px = pt & (TW - 1), py = (pt >> p.lgTW) & (TH - 1), tb = pt >> (p.lgTW + p.lgTH);
        const int b = b0 + tb, yy = y0 + py, xx = x0 + px;
        const bool pvalid = tb < TB && b < p.B && yy < p.H && xx < p.W;
        if (!pvalid) continue;
        const size_t pix = (size_t)yy * p.W + xx;
        float* yb = ybase + (size_t)b * p.Cout * HW + pix;
        if (split) {  // raw partial sums; splitk_epilogue_kernel finishes
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_tile0 + (wm * C::MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < p.Cout) yb[(size_t)co * HW] = acc[m][n][r];
                }
            continue;
        }
        const float nz = f_noise ? p.noise[(size_t)b * HW + pix] : 0.f;
        const float* st = f_style ? p.style + (size_t)b * p.style_stride : nullptr;
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
                    float* dst = yb + (size_t)co * HW;
                    if (f_accum) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// y = epi( sum_z ws[z] ) -- fixed summation order, one element per thread
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvArgs p, const float* __restrict__ ws, int ksplit) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * p.Cout * HW;
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_noise = p.flags & SPK_EPI_NOISE;
    const bool f_lrelu = p.flags & SPK_EPI_LRELU, f_style = p.flags & SPK_EPI_STYLE;
    const bool f_accum = p.flags & SPK_EPI_ACCUM;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < ksplit; ++z) v += ws[(size_t)z * total + idx];
        const size_t pix = idx % HW;
        const int co = (int)((idx / HW) % p.Cout);
        const int b = (int)(idx / (HW * p.Cout));
        v *= p.in_scale;
        if (f_bias) v += p.bias[co];
        if (f_noise) v += p.noise_w[co] * p.noise[(size_t)b * HW + pix];
        if (f_lrelu) v = v > 0.f ? v : v * p.slope;
        if (f_style) {
            const float* st = p.style + (size_t)b * p.style_stride;
            v = v * (st[co] + 1.f) + st[p.Cout + co];
        }
        if (f_accum) v += p.y[idx];
        p.y[idx] = v;
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
typedef Cfg<2, 2, 2, 2, 8> Cfg0;  //  128   128   general
typedef Cfg<1, 4, 2, 2, 8> Cfg1;  //   64   256   Cout <= 64, many pixels
typedef Cfg<2, 2, 1, 1, 8> Cfg2;  //   64    64   small problems
typedef Cfg<1, 4, 1, 1, 8> Cfg3;  //   32   128   Cout <= 32
typedef Cfg<2, 2, 2, 2, 4> Cfg4;  //  128   128   half-depth chunks: 43 KiB LDS -> 2-3 workgroups per CU
typedef Cfg<1, 4, 2, 2, 4> Cfg5;  //   64   256   half-depth chunks
constexpr int kNumConfigs = 6;

struct Geometry {
    int TW, TH, TB, PLANE, tiles_x, tiles_y, tiles_b, n_chunks, co_tiles;
    size_t lds_bytes;
    bool ok;
};

template <class C>
Geometry geometry(int B, int Cin, int Cout, int H, int W) {
    Geometry g;
    g.TW = std::min(32, spk::pow2_ceil(W));
    g.TH = std::min(C::PIX_T / g.TW, spk::pow2_ceil(H));
    g.TB = C::PIX_T / (g.TW * g.TH);
    // the wave's share of the input tile must fit its prefetch slots
    auto slots = [&]() { return spk::ceil_div(C::CI_T * g.TB / C::NW * (g.TH + 2) * (g.TW + 2), 64); };
    while (slots() > C::NSLOT && g.TB > 1 && (C::CI_T * (g.TB / 2)) % C::NW == 0) g.TB >>= 1;  // idle pixel groups
    while (slots() > C::NSLOT && g.TH > 1) g.TH >>= 1;
    g.PLANE = (g.TH + 2) * (g.TW + 2);
    g.ok = slots() <= C::NSLOT && (C::CI_T * g.TB) % C::NW == 0;
    g.tiles_x = spk::ceil_div(W, g.TW);
    g.tiles_y = spk::ceil_div(H, g.TH);
    g.tiles_b = spk::ceil_div(B, g.TB);
    g.n_chunks = spk::ceil_div(Cin, C::CI_T);
    g.co_tiles = spk::ceil_div(Cout, C::CO_T);
    const size_t in_floats = ((size_t)C::CI_T * g.TB * g.PLANE + 3) & ~(size_t)3;
    g.lds_bytes = 2 * (C::W_FLOATS + in_floats) * sizeof(float);
    if (g.lds_bytes > 160 * 1024) g.ok = false;
    return g;
}

Geometry geometry_cfg(int cfg, int B, int Cin, int Cout, int H, int W) {
    switch (cfg) {
        case 0: return geometry<Cfg0>(B, Cin, Cout, H, W);
        case 1: return geometry<Cfg1>(B, Cin, Cout, H, W);
        case 2: return geometry<Cfg2>(B, Cin, Cout, H, W);
        case 3: return geometry<Cfg3>(B, Cin, Cout, H, W);
        case 4: return geometry<Cfg4>(B, Cin, Cout, H, W);
        default: return geometry<Cfg5>(B, Cin, Cout, H, W);
    }
}

// number of ci-chunk slices so that the grid fills the chip
int pick_ksplit(const Geometry& g) {
    const long long tiles = (long long)g.tiles_x * g.tiles_y * g.tiles_b * g.co_tiles;
    int ks = 1;
    while (tiles * ks < 512 && g.n_chunks / (ks * 2) >= 4 && ks < 64) ks *= 2;
    return ks;
}

int resolve_ksplit(const Geometry& g, int requested, int* chunks_per_split) {
    int ks = requested > 0 ? requested : pick_ksplit(g);
    ks = std::max(1, std::min(ks, g.n_chunks));
    const int cps = spk::ceil_div(g.n_chunks, ks);
    if (chunks_per_split) *chunks_per_split = cps;
    return spk::ceil_div(g.n_chunks, cps);
}

template <class C>
int set_lds_attr(bool ups, size_t lds) {
    // dynamic LDS above 64 KiB needs the attribute raised; done once per instantiation
    static bool raised[2] = {false, false};
    if (lds <= 64 * 1024 || raised[ups]) return SPK_OK;
    hipError_t e = ups ? hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<C, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                       : hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<C, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
    raised[ups] = true;
    return SPK_OK;
}

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
    const Geometry g = geometry<C>(d->B, d->Cin, d->Cout, d->H, d->W);
    SPK_REQUIRE(g.ok, "conv3x3: config does not fit this shape (%dx%d, B=%d)", d->H, d->W, d->B);
    SPK_REQUIRE((size_t)g.TB * d->Cin * a.Hs * a.Ws < (1ull << 31), "conv3x3: image group too large for 32-bit offsets");
    a.lgTW = spk::ilog2(g.TW); a.lgTH = spk::ilog2(g.TH); a.lgTB = spk::ilog2(g.TB);
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
    a.n_chunks = g.n_chunks;
    a.style_stride = d->style_stride; a.flags = d->flags; a.slope = d->lrelu_slope; a.in_scale = d->in_scale;
    const int ksplit = resolve_ksplit(g, d->ksplit, &a.chunks_per_split);
    const size_t out_floats = (size_t)d->B * d->Cout * d->H * d->W;
    if (ksplit > 1) {
        SPK_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= ksplit * out_floats * sizeof(float),
                    "conv3x3: split-K x%d needs a %zu-byte workspace (see spk_conv3x3_workspace_bytes)", ksplit,
                    ksplit * out_floats * sizeof(float));
        a.y = static_cast<float*>(d->workspace);
    }
    int rc = set_lds_attr<C>(ups, g.lds_bytes);
    if (rc != SPK_OK) return rc;
    const long long gx = (long long)g.tiles_x * g.tiles_y * g.tiles_b;
    SPK_REQUIRE(gx < (1ll << 31), "conv3x3: grid too large");
    dim3 grid((unsigned)gx, (unsigned)g.co_tiles, (unsigned)ksplit);
    if (ups) hipLaunchKernelGGL((conv3x3_kernel<C, true>), grid, dim3(C::NTHREADS), g.lds_bytes, stream, a);
    else     hipLaunchKernelGGL((conv3x3_kernel<C, false>), grid, dim3(C::NTHREADS), g.lds_bytes, stream, a);
    rc = spk::check_launch("conv3x3_kernel");
    if (rc != SPK_OK || ksplit == 1) return rc;
    a.y = d->y;
    const unsigned blocks = (unsigned)std::min<size_t>((out_floats + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, stream, a,
                       static_cast<const float*>(d->workspace), ksplit);
    return spk::check_launch("splitk_epilogue_kernel");
}

void config_dims(int cfg, int* co_t, int* ci_t, int* pix_t) {
    int co = 0, ci = 0, px = 0;
    switch (cfg) {
        case 0: co = Cfg0::CO_T; ci = Cfg0::CI_T; px = Cfg0::PIX_T; break;
        case 1: co = Cfg1::CO_T; ci = Cfg1::CI_T; px = Cfg1::PIX_T; break;
        case 2: co = Cfg2::CO_T; ci = Cfg2::CI_T; px = Cfg2::PIX_T; break;
        case 3: co = Cfg3::CO_T; ci = Cfg3::CI_T; px = Cfg3::PIX_T; break;
        case 4: co = Cfg4::CO_T; ci = Cfg4::CI_T; px = Cfg4::PIX_T; break;
        case 5: co = Cfg5::CO_T; ci = Cfg5::CI_T; px = Cfg5::PIX_T; break;
    }
    if (co_t) *co_t = co;
    if (ci_t) *ci_t = ci;
    if (pix_t) *pix_t = px;
}

}  // namespace

extern "C" {

int spk_conv3x3_num_configs(void) { return kNumConfigs; }

int spk_conv3x3_pick_config(int B, int Cin, int Cout, int H, int W) {
    // Measured on MI355X (tools/bench_conv.py, profiles/): the half-depth-chunk configs (4, 5) win on
    // every decoder layer because 2-3 workgroups fit a CU and cover each other's barriers/epilogues.
    const long long pixels = (long long)B * H * W;
    int want;
    if (Cout <= 32) want = 3;
    else if (Cout <= 64) want = pixels >= 64 * 1024 ? 5 : 2;
    else want = pixels >= 2048 ? 4 : 2;
    // fall back along a fixed order if the preferred config cannot host the shape
    static const int order[6][6] = {{0, 4, 2, 3, 1, 5}, {1, 5, 2, 3, 0, 4}, {2, 3, 4, 0, 5, 1},
                                    {3, 2, 4, 0, 5, 1}, {4, 0, 2, 3, 5, 1}, {5, 1, 2, 3, 4, 0}};
    for (int i = 0; i < 6; ++i)
        if (geometry_cfg(order[want][i], B, Cin, Cout, H, W).ok) return order[want][i];
    return want;
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

int64_t spk_conv3x3_workspace_bytes(int config, int ksplit, int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return -1;
    if (config < 0) config = spk_conv3x3_pick_config(B, Cin, Cout, H, W);
    if (config >= kNumConfigs) return -1;
    const Geometry g = geometry_cfg(config, B, Cin, Cout, H, W);
    if (!g.ok) return -1;
    const int ks = resolve_ksplit(g, ksplit, nullptr);
    return ks > 1 ? (int64_t)ks * B * Cout * H * W * (int64_t)sizeof(float) : 0;
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
        case 4: return run<Cfg4>(d, s);
        case 5: return run<Cfg5>(d, s);
        default: return spk::fail(SPK_EINVAL, "conv3x3: bad config %d", cfg);
    }
}

}  // extern "C"
