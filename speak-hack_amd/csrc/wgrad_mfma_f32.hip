// Weight gradient of the 2-D convolution on the gfx950 f32 MFMA pipe:
//     dW[co][ci][ky][kx] = sum_{b,y,x} g[b,co,y,x] * xin[b,ci,y*S+ky-P,x*S+kx-P]
// as the GEMM  D[co][(tap,ci)] = sum_pix G[co][pix] * X[ci][pix (+) tap]  with the pixel axis as K.
//   A fragment (v_mfma_f32_32x32x2_f32): lane l holds G[co = l&31][pix = l>>5]   (LDS row pitch odd)
//   B fragment, one per tap:             lane l holds X[ci = l&31][pix (+) tap]   (LDS row pitch odd)
//   D: col = lane&31 = ci, row = output channel; one 32x32 accumulator tile PER TAP per wave, so
//   the A fragment is read once per k-step and reused by all taps.
// A workgroup owns a 64co x CI_T block of dW and walks a strided share of the pixel tiles
// (64 pixels each, with halo, xin formed exactly as the forward pass forms it: plain, bilinear x2
// upsampled, or BatchNorm-affine+ReLU).  Partial blocks go to a slab per (pixel split, wave pixel
// half) in a ci-contiguous layout (coalesced stores); wgrad_reduce_kernel sums the slabs in a fixed
// order (bitwise reproducible, no float atomics) and transposes to [Cout][Cin][kh][kw].
//
// replaces: the weight-gradient half of F.conv2d's backward for every conv on the path
// (styleganv1.py:625,630 conv1/conv2; the torchvision trunk convs of model.py:60-62).
#include "spk_common.hpp"
#ifndef WGRAD_CI32
#define WGRAD_CI32 0
#endif

#include <algorithm>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
// LDS-typed fragment pointers: a volatile read stays ONE ds_read_b32 with a 16-bit immediate offset.  Left to itself the compiler
// merges neighbouring reads into ds_read2_b32 (1 KB reach) and pays a v_add_u32 per pair -- and on gfx950 every vector-ALU
// instruction of an f32-MFMA kernel is paid for in matrix time: 67 of the 183 per 288 MFMAs of the wide kernel.
typedef __attribute__((address_space(3))) float wg_lds_f32;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wg_static_for<I + 1, N>(f);
    }
}

namespace {

// WG_BATCH_SCALE: the modulated convolution (StyleGAN2 variant): the forward input was x * s[b,ci] and the gradient reaching
// the conv output is g * d'[b,co] (demodulation x activation gain) -- both factors are applied while the tiles are staged, so
// neither rescaled tensor exists in HBM (in_scale = s [B][Cx], g_scale = d' [B][Cy])
enum { WG_PLAIN = 0, WG_UPSAMPLE = 1, WG_AFFINE_RELU = 2, WG_BATCH_SCALE = 3 };

struct WgradArgs {
    const float* g;        // [B,Cout,H,W]   output-side gradient
    const float* x;        // [B,Cin,Hs,Ws]  forward input
    const float* in_scale; // WG_AFFINE_RELU: [Cx]; WG_BATCH_SCALE: s [B][Cx]
    const float* in_shift;
    const float* g_scale;  // WG_BATCH_SCALE: d' [B][Cy]
    float* slabs;          // [n_slabs][Cout][TAPS][Cin]
    int B, Cin, Cout, H, W, Hs, Ws;   // Cin / Cout PER GROUP
    int lgTW, lgTH, lgTB;
    int tiles_x, tiles_y, n_tiles;
    // grouped form (G independent convs per launch): g has Cy = G*Cout channels, x has Cx, group q reads x channels
    // [q*gin, q*gin + Cin); Cout is a multiple of CO_T so that a workgroup's co block lies in one group
    int Cx, Cy, gin;
};

template <int KH, int KW, int S>
struct WShape {
    static constexpr int TAPS = KH * KW;
    // 7x7 (the ResNet stem, Cin = 3): the 32 lanes of the B fragment are (kx, ci) pairs instead of 32 input channels --
    // 7*3 = 21 of 32 lanes do work, where one-channel-per-lane would use 3 -- and there is one accumulator tile per tap
    // ROW (ky); needs KW * Cin <= 32.
    static constexpr bool PACK = KH == 7;
    // 4x4 (the weight gradient of ConvTranspose2d(4, s2, p1), styleganv1.py:231): 16 accumulator tiles would be 256 registers
    // on top of the staging state, so a workgroup owns ONE tap row (blockIdx.y carries the row) and holds KW tiles
    static constexpr bool ROWPASS = KH == 4;
    static constexpr int TP = PACK ? KH : (ROWPASS ? KW : TAPS);     // accumulator tiles per wave
    static constexpr int CI_T = PACK ? 4 : ((S == 2 || WGRAD_CI32) ? 32 : 64);
    static constexpr int WCI = PACK ? 1 : CI_T / 32, WPX = 2 / WCI;    // 4 waves = 2 (co) x WCI x WPX
    static constexpr int CO_T = 64, PIX_T = 64, PAD = (KH - 1) / 2;
    // Staging of the input tile: a thread owns KX plane positions and, of the CI_T channels, one of CS slices -- with
    // CS = 2 (the 64-channel 3x3 stride-1 form, whose tile has <= 128 positions) all 256 threads work on 32 channels
    // each instead of half of them on 64.
    static constexpr int CS = (!PACK && S == 1 && CI_T == 64) ? 2 : 1;
    static constexpr int NPT = 256 / CS;                        // threads per channel slice = positions per round
    static constexpr int CPT = CI_T / CS;                       // channels a thread stages
    static constexpr int KX = KH == 7 ? 3 : (S == 2 ? 2 : 1);   // plane positions (x NPT) a thread stages
};

// GEO = 0: the pixel-tile shape (TW x TH x TB) comes from the arguments.  GEO = 1: fixed 16 x 4 x 1 -- the shape of every
// layer whose plane is at least 16 x 4.  With the shape known at compile time the LDS offsets of the 32 k-steps are
// immediates instead of ~100 hoisted address registers, the kernel fits 256 registers and TWO workgroups share a CU
// (one covers the other's staging phase; with runtime geometry the compiler needs 402 registers: one wave per SIMD,
// and the MFMA pipe idles through every store / barrier phase).
template <int KH, int KW, int S, int MODE, int GEO>
__global__ __launch_bounds__(256, GEO ? 2 : 1) void wgrad_kernel(const WgradArgs p) {
    using SH = WShape<KH, KW, S>;
    constexpr int TAPS = SH::TAPS, TP = SH::TP, CI_T = SH::CI_T, CO_T = SH::CO_T, PIX_T = SH::PIX_T, PAD = SH::PAD;
    constexpr bool AFF = MODE == WG_AFFINE_RELU;
    constexpr int KX = SH::KX, NG = CO_T * PIX_T / 256, NPT = SH::NPT, CPT = SH::CPT;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // first channel (within the block) this thread stages: NPT is a multiple of 64, so it is the same for a whole wave --
    // kept in an SGPR, the BatchNorm scale / shift of the staged channels are scalar loads (as per-lane loads they were
    // 2 * CPT vector loads per tile on the serial store phase)
    static_assert(NPT % 64 == 0, "a wave stages one channel slice");
    const int cb = (wave / (NPT / 64)) * CPT;
    const int half = lane >> 5, l32 = lane & 31;
    const int wco = wave & 1, wci = (wave >> 1) % SH::WCI, wpx = (wave >> 1) / SH::WCI;

    const int lgTW = GEO ? 4 : p.lgTW, lgTH = GEO ? 2 : p.lgTH, lgTB = GEO ? 0 : p.lgTB;
    const int TW = 1 << lgTW, TH = 1 << lgTH, TB = 1 << lgTB;
    const int PW = (TW - 1) * S + KW, PH = (TH - 1) * S + KH, PLANE = PH * PW;
    const int GPITCH = PIX_T + 1, XPITCH = (TB * PLANE) | 1;
    float* const g_s = smem;
    float* const x_s = smem + CO_T * GPITCH;

    const int co0 = blockIdx.x * CO_T;                  // in the gradient tensor (all groups)
    const int grp = co0 / p.Cout;
    const int co_end = (grp + 1) * p.Cout;
    const int cx0 = grp * p.gin;                        // the group's first channel in x
    const int ci_blocks = (p.Cin + CI_T - 1) / CI_T;
    const int ci0 = (blockIdx.y % ci_blocks) * CI_T;
    const int pass = blockIdx.y / ci_blocks;            // tap row for ROWPASS kernels, else 0
    const int tap0 = SH::ROWPASS ? pass * KW : 0;
    const int nci = min(CI_T, p.Cin - ci0);
    const size_t HW = (size_t)p.H * p.W, src_plane = (size_t)p.Hs * p.Ws;

    f32x16 acc[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // plane positions this thread stages for every channel of every tile (relative to the tile origin)
    int xs_rem[KX], xs_r[KX], xs_c[KX], xs_tb[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int e = (tid % NPT) + NPT * k;
        xs_rem[k] = -1; xs_r[k] = 0; xs_c[k] = 0; xs_tb[k] = 0;
        if (e < TB * PLANE) {
            const int tb = e / PLANE, pidx = e - tb * PLANE;
            xs_rem[k] = e; xs_tb[k] = tb; xs_r[k] = pidx / PW; xs_c[k] = pidx - (pidx / PW) * PW;
        }
    }

    // prefetch registers: the next tile's global loads are in flight during the current tile's MFMAs
    float xg[KX * CPT], gg[NG];
    unsigned xok = 0;   // bit k: plane position k of the prefetched tile is inside the image
    unsigned gok = 0;   // bit i: gradient element i of the prefetched tile is inside the tensor

#define SPK_WG_PREFETCH(tile_)                                                                              \
    {                                                                                                       \
        int bx_ = (tile_);                                                                                  \
        const int tx_ = bx_ % p.tiles_x; bx_ /= p.tiles_x;                                                  \
        const int ty_ = bx_ % p.tiles_y;                                                                    \
        const int b0_ = (bx_ / p.tiles_y) << lgTB, y0_ = ty_ << lgTH, x0_ = tx_ << lgTW;                    \
        gok = 0;                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NG; ++i) {                                                    \
            const int e = tid + 256 * i;                                                                    \
            const int co = e / PIX_T, pix = e % PIX_T;                                                      \
            const int px = pix & (TW - 1), py = (pix >> lgTW) & (TH - 1), tb = pix >> (lgTW + lgTH);        \
            const int b = b0_ + tb, yy = y0_ + py, xx = x0_ + px;                                           \
            const bool ok = tb < TB && b < p.B && yy < p.H && xx < p.W && co0 + co < co_end;                \
            const size_t off = ok ? ((size_t)b * p.Cy + co0 + co) * HW + (size_t)yy * p.W + xx : 0;         \
            gg[i] = p.g[off];          /* masked at store time: no wait on the load here */                \
            if (ok) gok |= 1u << i;                                                                         \
        }                                                                                                   \
        xok = 0;                                                                                            \
        _Pragma("unroll") for (int k = 0; k < KX; ++k) {                                                    \
            const int uy = y0_ * S + xs_r[k] - PAD, ux = x0_ * S + xs_c[k] - PAD, b = b0_ + xs_tb[k];       \
            const bool ok = xs_rem[k] >= 0 && uy >= 0 && uy < p.Hs && ux >= 0 && ux < p.Ws && b < p.B;      \
            if (ok) xok |= 1u << k;                                                                         \
            const float* src = p.x + ((size_t)(ok ? b : 0) * p.Cx + cx0 + ci0) * src_plane +                \
                               (ok ? (size_t)uy * p.Ws + ux : 0);                                           \
            _Pragma("unroll") for (int ci = 0; ci < CPT; ++ci)                                              \
                xg[k * CPT + ci] = src[(size_t)(cb + ci < nci ? cb + ci : 0) * src_plane];                  \
        }                                                                                                   \
    }

#define SPK_WG_STORE()                                                                                      \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < NG; ++i) {                                                    \
            const int e = tid + 256 * i;                                                                    \
            g_s[(e / PIX_T) * GPITCH + (e % PIX_T)] = ((gok >> i) & 1u) ? gg[i] : 0.f;                      \
        }                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < KX; ++k) {                                                    \
            if (xs_rem[k] >= 0) {                                                                           \
                const bool ok = (xok >> k) & 1u;                                                            \
                float* dst = x_s + xs_rem[k];                                                               \
                _Pragma("unroll") for (int ci = 0; ci < CPT; ++ci) {                                        \
                    float v = xg[k * CPT + ci];                                                             \
                    const int cc = cb + ci;                                                                 \
                    if (AFF) v = fmaxf(v * p.in_scale[cx0 + ci0 + (cc < nci ? cc : 0)] +                    \
                                       p.in_shift[cx0 + ci0 + (cc < nci ? cc : 0)], 0.f);                   \
                    dst[cc * XPITCH] = (ok && cc < nci) ? v : 0.f;                                          \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
    }

    int tile = blockIdx.z;
#ifdef WGRAD_LAB
    long long lab_t[5] = {0, 0, 0, 0, 0};
    long long lab_a = __builtin_readcyclecounter();
    const long long lab_begin = lab_a;
    const unsigned long long lab_r0 = __builtin_amdgcn_s_memrealtime();
#define LAB_MARK(i_) { const long long n_ = __builtin_readcyclecounter(); lab_t[i_] += n_ - lab_a; lab_a = n_; }
#else
#define LAB_MARK(i_)
#endif
    if (tile < p.n_tiles) SPK_WG_PREFETCH(tile);
    LAB_MARK(0)
    for (; tile < p.n_tiles; tile += gridDim.z) {
        __syncthreads();          // previous tile fully consumed
        LAB_MARK(1)
        SPK_WG_STORE();
        __syncthreads();
        LAB_MARK(2)
        if (tile + (int)gridDim.z < p.n_tiles) SPK_WG_PREFETCH(tile + (int)gridDim.z);
        LAB_MARK(3)
        // ---- k-steps over this wave's pixel range: A read once per step, reused by every tap ----
        const float* ga = g_s + (wco * 32 + l32) * GPITCH;
        // B-fragment row of this lane: its input channel, or (PACK) its (kx, ci) pair: channel row + kx columns
        const float* xb = SH::PACK ? x_s + (l32 % nci) * XPITCH + min(l32 / nci, KW - 1)
                                   : x_s + (wci * 32 + l32) * XPITCH;
        constexpr int STEPS = PIX_T / 2 / SH::WPX;
        // Fragments are read one k-step ahead of the MFMAs that use them (two register sets, static indices after full
        // unrolling; sched_group_barrier pins the ds_read / MFMA interleave): with 144 accumulator registers the kernel
        // runs one wave per SIMD, so nothing else would hide the LDS latency.
        float fa[2], fb[2][TP];
#define SPK_WG_FRAG(st_, slot_)                                                                             \
    {                                                                                                       \
        const int pix = wpx * (PIX_T / SH::WPX) + 2 * (st_) + half;                                         \
        const int px = pix & (TW - 1), py = (pix >> lgTW) & (TH - 1);                                       \
        const int tb = min(pix >> (lgTW + lgTH), TB - 1);       /* idle pixel groups hold zeros in g_s */   \
        fa[slot_] = ga[pix];                                                                                \
        const float* xp = xb + tb * PLANE + (py * S) * PW + px * S;                                         \
        _Pragma("unroll") for (int t = 0; t < TP; ++t) {                                                    \
            const int tap = tap0 + t;                                                                       \
            const int ky = SH::PACK ? t : tap / KW, kx = SH::PACK ? 0 : tap % KW;   /* PACK: kx sits in the lane */ \
            fb[slot_][t] = xp[ky * PW + kx];                                                                \
        }                                                                                                   \
    }
        SPK_WG_FRAG(0, 0);
        wg_static_for<0, STEPS>([&](auto s_) {
            constexpr int st = decltype(s_)::value;
            if constexpr (st + 1 < STEPS) {
                SPK_WG_FRAG(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, TP + 1, 0);
            }
#pragma unroll
            for (int t = 0; t < TP; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1], fb[st & 1][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, TP, 0);
        });
#undef SPK_WG_FRAG
        LAB_MARK(4)
    }
#ifdef WGRAD_LAB
    if (KH == 3 && blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 3 && (tid & 63) == 0)
        printf("GEO %d wave %d: first-prefetch %lld  barrier-wait %lld  store+barrier %lld  prefetch-issue %lld  mfma %lld  total %lld (tiles %d) in %llu ticks of 10 ns\n",
               GEO, wave, lab_t[0], lab_t[1], lab_t[2], lab_t[3], lab_t[4], (long long)__builtin_readcyclecounter() - lab_begin,
               (p.n_tiles - (int)blockIdx.z + (int)gridDim.z - 1) / (int)gridDim.z,
               (unsigned long long)(__builtin_amdgcn_s_memrealtime() - lab_r0));
#endif
#undef SPK_WG_PREFETCH
#undef SPK_WG_STORE

    // ---- partial block -> slab [slab][co][tap][ci] (ci contiguous: 128-B stores per half wave) ----
    const int slab = blockIdx.z * SH::WPX + wpx;
    float* out = p.slabs + (size_t)slab * p.Cy * TAPS * p.Cin;
    // D column of this lane: an input channel, or (PACK) the (kx, ci) pair l32 = kx * nci + ci
    const int ci = SH::PACK ? l32 % nci : ci0 + wci * 32 + l32;
    const int kx_l = SH::PACK ? l32 / nci : 0;
    const bool col_ok = SH::PACK ? l32 < KW * nci : ci < p.Cin;
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int tap = SH::PACK ? t * KW + kx_l : tap0 + t;
            if (co < co_end && col_ok) out[((size_t)co * TAPS + tap) * p.Cin + ci] = acc[t][r];
        }
}

// ---- 3x3 stride-1 weight gradient, staging interleaved with the MFMAs ------------------------------------------------
// Same block (64co x 64ci x 9 taps, 4 waves, 16x4-pixel tiles) and the same arithmetic order as wgrad_kernel<3,3,1>, but
// the LDS tile is double-buffered and the next tile's staging rides in the shadow of this tile's MFMAs: k-steps 0-15
// issue its 48 global loads per thread (3 per step), k-steps 16-31 store them to the other buffer (3 per step), and one
// barrier ends the tile.  In the serial form a tile cost 18.6K cycles of MFMA + 8.6K of load issue, LDS stores and two
// barriers (profiles/r01_m_core_clock_under_load.txt).  The tile shape is a compile-time constant, so every LDS offset is
// an immediate and a staging piece is a clamp, a load or store and a select -- the per-piece index arithmetic is what
// made an earlier pipelined attempt slower than the serial kernel.
constexpr int WP_TW = 16, WP_PW = WP_TW + 2, WP_PLANE = 6 * WP_PW, WP_GPITCH = 65, WP_XPITCH = WP_PLANE | 1;
constexpr int WP_BUF = 64 * WP_GPITCH + 64 * WP_XPITCH + 4;      // floats per buffer (+ a dump slot for idle lanes)

__global__ __launch_bounds__(256) void wgrad3x3_pipe_kernel(const WgradArgs p) {
    constexpr int TAPS = 9, CI_T = 64, CO_T = 64, TW = WP_TW, PW = WP_PW, PLANE = WP_PLANE;
    constexpr int GPITCH = WP_GPITCH, XPITCH = WP_XPITCH, BUF = WP_BUF;
    constexpr int NGL = 16, NXL = 32, STEPS = 32, PPS = (NGL + NXL) / (STEPS / 2);   // 3 staging pieces per k-step
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wco = wave & 1, wci = wave >> 1;

    const int co0 = blockIdx.x * CO_T;
    const int grp = co0 / p.Cout;
    const int co_end = (grp + 1) * p.Cout;
    const int cx0 = grp * p.gin;
    const int ci0 = blockIdx.y * CI_T;
    const int nci = min(CI_T, p.Cin - ci0);
    const size_t HW = (size_t)p.H * p.W, src_plane = (size_t)p.Hs * p.Ws;

    // staging roles.  Gradient tile: thread -> pixel tid % 64, output channels tid / 64 + 4 i.  Input tile: thread -> plane
    // position tid % 128 (108 of them exist), channels 32 (tid / 128) + i (a wave-uniform slice).
    const int gpix = tid & 63, gpx = gpix & (TW - 1), gpy = gpix >> 4, gco = wave;
    const int xpos = tid & 127, cbw = (wave >> 1) * NXL;
    const bool xpos_ok = xpos < PLANE;
    const int xr = xpos / PW, xc = xpos - xr * PW;
    const int g_dst = gco * GPITCH + gpix;                                      // + 4 i GPITCH
    const int x_dst = xpos_ok ? CO_T * GPITCH + cbw * XPITCH + xpos : BUF - 1;  // + i XPITCH (idle lanes: the dump slot)
    const int x_dst_step = xpos_ok ? XPITCH : 0;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float gg[NGL], xg[NXL];
    bool g_ok = false, x_ok = false;          // the tile being staged: this thread's pixel / plane position is inside the image
    const float* g_ptr = p.g;                 // ... and the address of its element in channel 0 of the block
    const float* x_ptr = p.x;

    // tile -> pointers and validity of this thread's elements (elements outside the image read the tensor's first float)
    auto aim = [&](int tile) {
        const int tx = tile % p.tiles_x;
        const int q = tile / p.tiles_x;
        const int ty = q % p.tiles_y, b = q / p.tiles_y;
        const int yy = ty * 4 + gpy, xx = tx * TW + gpx;
        const bool live = tile < p.n_tiles;
        g_ok = live && yy < p.H && xx < p.W;
        g_ptr = p.g + (g_ok ? (size_t)b * p.Cy * HW + (size_t)yy * p.W + xx : 0);
        const int uy = ty * 4 + xr - 1, ux = tx * TW + xc - 1;
        x_ok = live && xpos_ok && uy >= 0 && uy < p.Hs && ux >= 0 && ux < p.Ws;
        x_ptr = p.x + (x_ok ? ((size_t)b * p.Cx + cx0 + ci0) * src_plane + (size_t)uy * p.Ws + ux : 0);
    };
    // staging pieces (j static): 0..15 the gradient element j, 16..47 the input channel j - 16
    auto load_piece = [&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (j < NGL) {
            const int c = min(co0 + gco + 4 * j, co_end - 1);      // rows past the group's last channel: clamped (see the store)
            gg[j] = g_ptr[g_ok ? (size_t)c * HW : 0];
        } else {
            const int c = min(cbw + (j - NGL), nci - 1);
            xg[j - NGL] = x_ptr[x_ok ? (size_t)c * src_plane : 0];
        }
    };
    auto store_piece = [&](float* buf, auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (j < NGL) {
            // only the PIXEL axis (the contraction) needs zeros; rows of channels past the block's last one hold the
            // clamped row's values and feed accumulator rows / columns the epilogue never writes (a uniform test here
            // would also split the loop body into branches)
            buf[g_dst + 4 * j * GPITCH] = g_ok ? gg[j] : 0.f;
        } else {
            buf[x_dst + (j - NGL) * x_dst_step] = x_ok ? xg[j - NGL] : 0.f;
        }
    };

    int tile = blockIdx.z;
    int cur = 0;
    if (tile < p.n_tiles) {                  // first tile: staged the serial way
        aim(tile);
        wg_static_for<0, NGL + NXL>([&](auto j_) { load_piece(j_); });
        wg_static_for<0, NGL + NXL>([&](auto j_) { store_piece(smem, j_); });
    }
    __syncthreads();

    // one tile's 32 k-steps out of buffer `cur`; PIPE: the next tile is staged into the other buffer on the way
    auto run_tile = [&](auto pipe_) {
        constexpr bool PIPE = decltype(pipe_)::value;
        const float* cbuf = smem + cur * BUF;
        float* nbuf = smem + (cur ^ 1) * BUF;
        const volatile wg_lds_f32* ga = (const volatile wg_lds_f32*)(cbuf + (wco * 32 + l32) * GPITCH + half);              // pixel 2 st + half of this lane's channel row
        const volatile wg_lds_f32* xb = (const volatile wg_lds_f32*)(cbuf + CO_T * GPITCH + (wci * 32 + l32) * XPITCH + half);
        float fa[2], fb[2][TAPS];
#define SPK_WP_FRAG(st_, slot_)                                                                               \
    {                                                                                                         \
        constexpr int px_ = (2 * (st_)) & (TW - 1), py_ = (2 * (st_)) >> 4;                                   \
        fa[slot_] = ga[2 * (st_)];                                                                            \
        _Pragma("unroll") for (int t = 0; t < TAPS; ++t) fb[slot_][t] = xb[(py_ + t / 3) * PW + px_ + t % 3];  \
    }
        SPK_WP_FRAG(0, 0);
        wg_static_for<0, STEPS>([&](auto s_) {
            constexpr int st = decltype(s_)::value;
            if constexpr (PIPE && st == STEPS / 2) __builtin_amdgcn_sched_barrier(0);      // see wgrad3x3_wide_kernel
            if constexpr (st + 1 < STEPS) SPK_WP_FRAG(st + 1, (st + 1) & 1);
            if constexpr (PIPE && st < STEPS / 2)
                wg_static_for<PPS * st, PPS * st + PPS>([&](auto j_) { load_piece(j_); });
            if constexpr (PIPE && st >= STEPS / 2)
                wg_static_for<PPS * (st - STEPS / 2), PPS * (st - STEPS / 2) + PPS>([&](auto j_) { store_piece(nbuf, j_); });
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1], fb[st & 1][t], acc[t], 0, 0, 0);
            // order: next step's fragments, then the MFMAs with one staging access behind every third
            if constexpr (st + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, TAPS + 1, 0);
            if constexpr (PIPE) {
#pragma unroll
                for (int k = 0; k < PPS; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x8, TAPS / PPS, 0);
                    __builtin_amdgcn_sched_group_barrier(st < STEPS / 2 ? 0x20 : 0x200, 1, 0);
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x8, TAPS, 0);
            }
        });
#undef SPK_WP_FRAG
    };

    // ONE loop body: after the last tile the staging still runs, with every piece masked off (it reads the tensors' first
    // floats and fills the idle buffer with zeros).  With a second, staging-free copy of the tile for that case the
    // accumulators cross the if / else merge in VGPRs: 144 v_accvgpr_write before and 144 v_accvgpr_read after EVERY tile's
    // 288 MFMAs (and twice the code).
    for (; tile < p.n_tiles; tile += gridDim.z) {
        aim(tile + (int)gridDim.z);
        run_tile(std::true_type{});
        __syncthreads();                      // every wave is done with `cur`, and the other buffer is complete
        cur ^= 1;
    }

    // ---- partial block -> slab [slab][co][tap][ci] (ci contiguous: 128-B stores per half wave) ----
    float* out = p.slabs + (size_t)blockIdx.z * p.Cy * TAPS * p.Cin;
    const int ci = ci0 + wci * 32 + l32;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < co_end && ci < p.Cin) out[((size_t)co * TAPS + t) * p.Cin + ci] = acc[t][r];
        }
}

// ---- 3x3 stride-1 weight gradient with 16-byte global loads ---------------------------------------------------------------
// The kernels above are bound by the ISSUE of their loads, not by the MFMA pipe (profiles/r01_k_wgrad_phases.txt: 48 dword
// loads per thread and tile, ~1 000 line requests against 18.4K cycles of MFMA).  Here rows are loaded 16 bytes per lane: a
// gradient row of the tile is TW / 4 aligned float4, an input row the same plus the two halo columns as single dwords --
// 13 (TW = 16) load instructions per thread and tile of 288 MFMAs where the dword form issues 48.  (A second accumulator tile
// set per wave -- 128co x 64ci blocks, 1.5x the FLOPs per staged byte -- was built and dropped: MFMA accumulators live in the
// 256 AGPRs only, two 9-tap sets need 288.)
// Staging roles make everything but the tile origin a launch constant: piece i of the input tile is a plane ROW of 64
// channels, so its global offset is a per-thread base plus a wave-uniform term, its LDS address an immediate offset and its
// row test a scalar compare.  Same 64co x 64ci block, LDS images (odd row pitch), per-element accumulation order and slab
// layout as wgrad3x3_pipe_kernel: results are bitwise equal to its for equal pixel splits.
// Needs W % 4 == 0, W >= 8 and 16-byte aligned tensors (host-checked: wide_takes).
template <int TW_>
struct WideShapeT {
    static constexpr int CO_T = 64, CI_T = 64, NT = 256;
    static constexpr int TW = TW_, TH = 64 / TW, PW = TW + 2, PH = TH + 2, PLANE = PH * PW;
    static constexpr int GPITCH = TW * TH + 1, XPITCH = PLANE | 1;
    static constexpr int BUF = (CO_T * GPITCH + CI_T * XPITCH + 1 + 3) & ~3;       // floats per buffer (+ a dump slot: BUF - 1)
    static constexpr int NG4 = CO_T / 16;            // float4 pieces of the gradient tile per thread (16 channels per piece)
    static constexpr int NXK = TW / 4;               // float4 per plane row (the aligned columns 0 .. TW-1)
    static constexpr int RS = 256 / (CI_T * NXK);    // plane rows covered by one float4 piece (1 or 2)
    static constexpr int NX4 = (PH + RS - 1) / RS;   // float4 pieces of the input tile
    static constexpr int NXH = (PH + 1) / 2;         // halo-dword pieces (piece = two plane rows x 2 sides of 64 channels)
    static constexpr int NL = NG4 + NX4 + NXH, NS = 4 * NG4 + 4 * NX4 + NXH;
    static constexpr int STEPS = 32, HS = STEPS / 2;
    static constexpr int PL = (NL + HS - 1) / HS, PS = (NS + HS - 1) / HS;
    static constexpr int NMFMA = 9, NFRAG = 10;
    static_assert(XPITCH % 2 == 1, "odd pitch");
};
// (the shape the folded-upsample kernel below shares: the 16 x 4 tile)
template <int MCO, int MCI>
struct WideShape : WideShapeT<16> { static_assert(MCO == 1 && MCI == 1, "one accumulator tile set per wave"); };

// TW = 16: 16 x 4 pixel tiles (18 x 6 plane); TW = 8: 8 x 8 tiles (10 x 10 plane) for the 8^2 layers, one tile per image.
// SB (single buffer): ONE LDS tile (44.5 KB) and <= 256 registers, so that TWO workgroups share a CU -- the next tile's loads
// still fly behind the first half of the k-steps, but its LDS stores wait for a barrier after the last MFMA (the other
// workgroup's MFMAs fill that gap, and each other's first-tile latency, slab stores and barriers).
template <int TW_, int MODE, bool SB = false>
__global__ __launch_bounds__(256, SB ? 2 : 1) void wgrad3x3_wide_kernel(const WgradArgs p) {
    using SH = WideShapeT<TW_>;
    constexpr int TAPS = 9, CO_T = SH::CO_T, CI_T = SH::CI_T, TW = SH::TW, TH = SH::TH, PW = SH::PW, PH = SH::PH;
    constexpr int GPITCH = SH::GPITCH, XPITCH = SH::XPITCH, BUF = SH::BUF;
    constexpr int NG4 = SH::NG4, NXK = SH::NXK, RS = SH::RS, NX4 = SH::NX4, NXH = SH::NXH, NL = SH::NL, NS = SH::NS;
    constexpr int STEPS = SH::STEPS, HS = SH::HS, PL = SH::PL, PS = SH::PS;
    constexpr bool AFF = MODE == WG_AFFINE_RELU, BSC = MODE == WG_BATCH_SCALE;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wco = wave & 1, wci = wave >> 1;

    const int co0 = blockIdx.x * CO_T;
    const int grp = co0 / p.Cout;
    const int co_end = (grp + 1) * p.Cout;
    const int cx0 = grp * p.gin;
    const int ci0 = blockIdx.y * CI_T;
    const int nci = min(CI_T, p.Cin - ci0);

    // ---- staging roles (launch constants; only the tile origin moves) ----
    // gradient piece i: float4 gk of tile row grow of channel (tid >> 4) + 16 i
    const int gk = tid & (TW / 4 - 1), grow = (tid / (TW / 4)) & (TH - 1);
    int g_off[NG4], g_ch[NG4];
#pragma unroll
    for (int i = 0; i < NG4; ++i) {
        g_ch[i] = min(co0 + (tid >> 4) + 16 * i, co_end - 1);                                      // rows past the last channel: clamped (never written out)
        g_off[i] = (g_ch[i] * p.H + grow) * p.W + 4 * gk;
    }
    float g_sc[NG4];                                                                              // BSC: d'[b, g_ch[i]] of the tile being staged
#pragma unroll
    for (int i = 0; i < NG4; ++i) g_sc[i] = 1.f;
    const int g_dst = (tid >> 4) * GPITCH + grow * TW + 4 * gk;                                   // + 16 i GPITCH + j
    // input piece i: float4 xk of plane row RS i + xr of channel xc
    const int xk = tid & (NXK - 1), xc = (tid / NXK) & (CI_T - 1), xr = tid / (NXK * CI_T);
    const int x_chan = cx0 + ci0 + min(xc, nci - 1);
    const int x_base = (x_chan * p.Hs + xr - 1) * p.Ws + 4 * xk;                                  // + RS i Ws
    const int x_dst = CO_T * GPITCH + xc * XPITCH + xr * PW + 1 + 4 * xk;                         // + RS i PW + j
    float x_sc = 1.f, x_sh = 0.f;
    if (AFF) { x_sc = p.in_scale[x_chan]; x_sh = p.in_shift[x_chan]; }
    // halo piece i: column -1 / TW (hs) of plane row 2 i + hr of channel hc
    const int hs = tid & 1, hc = (tid >> 1) & (CI_T - 1), hr = tid >> 7;
    const int h_chan = cx0 + ci0 + min(hc, nci - 1);
    const int h_base = (h_chan * p.Hs + hr - 1) * p.Ws + (hs ? TW : -1);                          // + 2 i Ws
    const int h_dst = CO_T * GPITCH + hc * XPITCH + hr * PW + (hs ? TW + 1 : 0);                  // + 2 i PW
    float h_sc = 1.f, h_sh = 0.f;
    if (AFF) { h_sc = p.in_scale[h_chan]; h_sh = p.in_shift[h_chan]; }

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // prefetch registers and, for the tile being staged: its base pointers, origin row and which of this thread's columns exist
    f32x4 gq[NG4], xq[NX4];
    float hq[NXH];
    const float* gbase = p.g;
    const float* xbase = p.x;
    int ty0 = 0;
    bool g_ok = false, xcol_ok = false, hcol_ok = false;
    auto aim = [&](int tile) {
        const int tx = tile % p.tiles_x;
        const int q = tile / p.tiles_x;
        const int ty = q % p.tiles_y, b = q / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        ty0 = y0;
        gbase = p.g + ((size_t)b * p.Cy * p.H + y0) * p.W + x0;
        xbase = p.x + ((size_t)b * p.Cx * p.Hs + y0) * p.Ws + x0;
        const bool live = tile < p.n_tiles;
        g_ok = live && y0 + grow < p.H && x0 + 4 * gk < p.W;
        xcol_ok = live && x0 + 4 * xk < p.Ws;
        hcol_ok = live && (unsigned)(x0 + (hs ? TW : -1)) < (unsigned)p.Ws;
        if constexpr (BSC) {                     // the staged tile's image decides the modulation / demodulation factors
            const int bb = live ? b : 0;
            x_sc = p.in_scale[(size_t)bb * p.Cx + x_chan];
            h_sc = p.in_scale[(size_t)bb * p.Cx + h_chan];
#pragma unroll
            for (int i = 0; i < NG4; ++i) g_sc[i] = p.g_scale[(size_t)bb * p.Cy + g_ch[i]];
        }
    };
    // plane row r of the tile being staged lies inside the image (and inside the plane: the last piece of a two-row role)
    auto row_ok = [&](int r) { return r < PH && (unsigned)(ty0 + r - 1) < (unsigned)p.Hs; };
    auto load_piece = [&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (j < NG4) {
            gq[j] = *reinterpret_cast<const f32x4*>(g_ok ? gbase + g_off[j] : p.g);
        } else if constexpr (j < NG4 + NX4) {
            constexpr int i = j - NG4;
            const bool ok = xcol_ok && row_ok(RS * i + xr);
            xq[i] = *reinterpret_cast<const f32x4*>(ok ? xbase + (x_base + RS * i * p.Ws) : p.x);
        } else if constexpr (j < NL) {
            constexpr int i = j - NG4 - NX4;
            const bool ok = hcol_ok && row_ok(2 * i + hr);
            hq[i] = *(ok ? xbase + (h_base + 2 * i * p.Ws) : p.x);
        }
    };
    auto store_piece = [&](float* buf, auto s_) {
        constexpr int s = decltype(s_)::value;
        if constexpr (s < 4 * NG4) {
            constexpr int i = s / 4, j = s % 4;
            buf[g_dst + 16 * i * GPITCH + j] = g_ok ? (BSC ? gq[i][j] * g_sc[i] : gq[i][j]) : 0.f;   // only the PIXEL axis (the contraction) needs zeros
        } else if constexpr (s < 4 * NG4 + 4 * NX4) {
            constexpr int i = (s - 4 * NG4) / 4, j = (s - 4 * NG4) % 4;
            float v = xq[i][j];
            if (AFF) v = fmaxf(v * x_sc + x_sh, 0.f);
            if (BSC) v *= x_sc;
            // (a two-row role's last piece: plane row PH does not exist -- those lanes hit the dump slot, no branch)
            const int dst = (RS * i + RS - 1 < PH || RS * i + xr < PH) ? x_dst + RS * i * PW + j : BUF - 1;
            buf[dst] = (xcol_ok && row_ok(RS * i + xr)) ? v : 0.f;
        } else if constexpr (s < NS) {
            constexpr int i = s - 4 * NG4 - 4 * NX4;
            float v = hq[i];
            if (AFF) v = fmaxf(v * h_sc + h_sh, 0.f);
            if (BSC) v *= h_sc;
            const int dst = (2 * i + 1 < PH || 2 * i + hr < PH) ? h_dst + 2 * i * PW : BUF - 1;
            buf[dst] = (hcol_ok && row_ok(2 * i + hr)) ? v : 0.f;
        }
    };

    int tile = blockIdx.z;
    int cur = 0;
    if (tile < p.n_tiles) {                  // first tile: staged the serial way
        aim(tile);
        wg_static_for<0, NL>([&](auto j_) { load_piece(j_); });
        wg_static_for<0, NS>([&](auto s_) { store_piece(smem, s_); });
    }
    __syncthreads();

    // one tile's 32 k-steps out of buffer `cur`; the next tile is staged into the other buffer on the way (its loads behind the
    // first half of the k-steps, its LDS stores behind the second half)
    auto run_tile = [&]() {
        const float* cbuf = smem + (SB ? 0 : cur * BUF);
        float* nbuf = smem + (SB ? 0 : (cur ^ 1) * BUF);
        const volatile wg_lds_f32* ga = (const volatile wg_lds_f32*)(cbuf + (wco * 32 + l32) * GPITCH + half);
        const volatile wg_lds_f32* xb = (const volatile wg_lds_f32*)(cbuf + CO_T * GPITCH + (wci * 32 + l32) * XPITCH + half);
        float fa[2], fb[2][TAPS];
#define SPK_WW_FRAG(st_, slot_)                                                                               \
    {                                                                                                         \
        constexpr int px_ = (2 * (st_)) & (TW - 1), py_ = (2 * (st_)) / TW;                                   \
        fa[slot_] = ga[2 * (st_)];                                                                            \
        _Pragma("unroll") for (int t = 0; t < TAPS; ++t)                                                      \
            fb[slot_][t] = xb[(py_ + t / 3) * PW + px_ + t % 3];                                              \
    }
        SPK_WW_FRAG(0, 0);
        wg_static_for<0, STEPS>([&](auto s_) {
            constexpr int st = decltype(s_)::value;
            // Nothing may be scheduled across the load / store boundary: left alone, the scheduler hoists the stores' VALU
            // halves (the zero-select of out-of-image pieces, the folded BatchNorm affine) up to the loads they consume -- and
            // every load is then followed by s_waitcnt vmcnt(0), a full memory latency with the MFMA pipe idle (one wave per SIMD).
            if constexpr (st == HS) __builtin_amdgcn_sched_barrier(0);
            if constexpr (st + 1 < STEPS) SPK_WW_FRAG(st + 1, (st + 1) & 1);
            constexpr int nl = st < HS ? ((PL * st + PL <= NL) ? PL : (PL * st < NL ? NL - PL * st : 0)) : 0;
            constexpr int s0 = PS * (st - HS);
            constexpr int ns = (!SB && st >= HS) ? ((s0 + PS <= NS) ? PS : (s0 < NS ? NS - s0 : 0)) : 0;
            if constexpr (nl > 0) wg_static_for<PL * st, PL * st + nl>([&](auto j_) { load_piece(j_); });
            if constexpr (ns > 0) wg_static_for<s0, s0 + ns>([&](auto q_) { store_piece(nbuf, q_); });
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1], fb[st & 1][t], acc[t], 0, 0, 0);
            // order: next step's fragments first, then the MFMAs with the staging accesses spread between them
            if constexpr (st + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, SH::NFRAG, 0);
            constexpr int np = nl > 0 ? nl : ns;
            if constexpr (np > 0) {
                constexpr int per = TAPS / (np + 1);
                wg_static_for<0, np>([&](auto k_) {
                    __builtin_amdgcn_sched_group_barrier(0x8, per, 0);
                    __builtin_amdgcn_sched_group_barrier(nl > 0 ? 0x20 : 0x200, 1, 0);
                });
                __builtin_amdgcn_sched_group_barrier(0x8, TAPS - per * np, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x8, TAPS, 0);
            }
        });
#undef SPK_WW_FRAG
        if constexpr (SB) {
            __syncthreads();                  // every wave is done reading the tile
            wg_static_for<0, NS>([&](auto s_) { store_piece(nbuf, s_); });
        }
    };

    // ONE loop body: after the last tile the staging still runs, with every piece masked off (it reads the tensors' first
    // floats and fills the idle buffer with zeros).  With a second, staging-free copy of the tile for that case the
    // accumulators cross the if / else merge in VGPRs: 144 v_accvgpr_write before and 144 v_accvgpr_read after EVERY tile's
    // 288 MFMAs (and twice the code).
    for (; tile < p.n_tiles; tile += gridDim.z) {
        aim(tile + (int)gridDim.z);
        run_tile();
        __syncthreads();                      // every wave is done with `cur`, and the other buffer is complete
        cur ^= 1;
    }

    // ---- partial block -> slab [slab][co][tap][ci] (ci contiguous: 128-B stores per half wave) ----
    float* out = p.slabs + (size_t)blockIdx.z * p.Cy * TAPS * p.Cin;
    const int ci = ci0 + wci * 32 + l32;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < co_end && ci < p.Cin) out[((size_t)co * TAPS + t) * p.Cin + ci] = acc[t][r];
        }
}

// ---- 3x3 STRIDE-2 weight gradient (the trunk's downsampling convs, the discriminator's conv2 of every block) -----------
// dW[co][ci][ky][kx] = sum g[b,co,y,x] * in(x)[b,ci,2y+ky-1,2x+kx-1].  A stride-2 conv reads 4.6 input values per output pixel
// and channel where a stride-1 one reads 1.7, so the generic kernel above (64co x 32ci block, 64 dword gathers + 16 loads per
// thread for 144 MFMAs per wave, staged serially) spends its time issuing loads: 33-42 TFLOP/s on the trunk, 65 on the
// discriminator.  Here the input tile serves FOUR co tiles: block = 128co x 32ci, wave w owns co rows [32w, 32w+32) and all 64
// pixels of the tile (288 MFMAs per wave and tile), the (2 TH + 1) x (2 TW + 1) input plane is loaded as 16-byte row
// segments (columns 2 x0 .. 2 x0 + 2 TW - 1 are aligned; the left halo column is one dword per row), and -- as in
// wgrad3x3_wide_kernel -- the LDS tile is double-buffered with the next tile's loads behind k-steps 0-15 and its LDS stores
// behind k-steps 16-31: 19 load + 70 store instructions per thread against 288 MFMAs.
// TW = 16 (16 x 4 output pixels, 9 x 33 plane) or 8 (8 x 8, 17 x 17: the 8^2 outputs).  Needs W % 4 == 0, W >= TW / 2,
// Hin = 2 H, Win = 2 W, 16-byte aligned tensors (host-checked: s2_takes).
template <int TW_>
struct S2Shape {
    static constexpr int CO_T = 128, CI_T = 32, TW = TW_, TH = 64 / TW, PW = 2 * TW + 1, PH = 2 * TH + 1, PLANE = PH * PW;
    static constexpr int GPITCH = 65, XPITCH = PLANE | 1;
    static constexpr int BUF = (CO_T * GPITCH + CI_T * XPITCH + 1 + 3) & ~3;        // floats per buffer (+ a dump slot: BUF - 1)
    static constexpr int NG4 = CO_T / 16;                 // float4 pieces of the gradient tile per thread (16 channels per piece)
    static constexpr int NXK = 2 * TW / 4;                // float4 per plane row
    static constexpr int RS = 256 / (CI_T * NXK);         // plane rows covered by one piece (1 or 2)
    static constexpr int NX4 = (PH + RS - 1) / RS;        // float4 pieces of the input tile
    static constexpr int NXH = (PH + 7) / 8;              // halo-dword pieces (piece = 8 plane rows x 32 channels)
    static constexpr int NL = NG4 + NX4 + NXH, NS = 4 * NG4 + 4 * NX4 + NXH;
    static constexpr int STEPS = 32, HS = STEPS / 2;
    static constexpr int PL = (NL + HS - 1) / HS, PS = (NS + HS - 1) / HS;
    static_assert(XPITCH % 2 == 1, "odd pitch: the 32 channels of a fragment hit 32 banks");
};

// LAB (builds with -DSPK_WGRAD_S2_LAB only): knock-outs for timing -- 1 = no global loads, 2 = no LDS stores of the staging,
// 4 = no MFMAs, 8 = no fragment reads
template <int TW_, int MODE, int LAB = 0>
__global__ __launch_bounds__(256) void wgrad3x3_s2_kernel(const WgradArgs p) {
    using SH = S2Shape<TW_>;
    constexpr int TAPS = 9, CO_T = SH::CO_T, CI_T = SH::CI_T, TW = SH::TW, TH = SH::TH, PW = SH::PW, PH = SH::PH;
    constexpr int GPITCH = SH::GPITCH, XPITCH = SH::XPITCH, BUF = SH::BUF;
    constexpr int NG4 = SH::NG4, NXK = SH::NXK, RS = SH::RS, NX4 = SH::NX4, NXH = SH::NXH, NL = SH::NL, NS = SH::NS;
    constexpr int STEPS = SH::STEPS, HS = SH::HS, PL = SH::PL, PS = SH::PS;
    constexpr bool AFF = MODE == WG_AFFINE_RELU;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;

    const int co0 = blockIdx.x * CO_T;
    const int grp = co0 / p.Cout;
    const int co_end = (grp + 1) * p.Cout;
    const int cx0 = grp * p.gin;
    const int ci0 = blockIdx.y * CI_T;
    const int nci = min(CI_T, p.Cin - ci0);

    // ---- staging roles (launch constants; only the tile origin moves) ----
    // gradient piece i: float4 gk of tile row grow of channel (tid >> 4) + 16 i
    const int gk = tid & (TW / 4 - 1), grow = (tid / (TW / 4)) & (TH - 1);
    int g_off[NG4];
#pragma unroll
    for (int i = 0; i < NG4; ++i)
        g_off[i] = (min(co0 + (tid >> 4) + 16 * i, co_end - 1) * p.H + grow) * p.W + 4 * gk;     // rows past the last channel: clamped (never written out)
    const int g_dst = (tid >> 4) * GPITCH + grow * TW + 4 * gk;                                    // + 16 i GPITCH + j
    // input piece i: float4 xk of plane row RS i + xr of channel xc
    const int xk = tid & (NXK - 1), xc = (tid / NXK) & (CI_T - 1), xr = tid / (NXK * CI_T);
    const int x_chan = cx0 + ci0 + min(xc, nci - 1);
    const int x_base = (x_chan * p.Hs + xr - 1) * p.Ws + 4 * xk;                                   // + RS i Ws
    const int x_dst = CO_T * GPITCH + xc * XPITCH + xr * PW + 1 + 4 * xk;                          // + RS i PW + j
    float x_sc = 1.f, x_sh = 0.f;
    if (AFF) { x_sc = p.in_scale[x_chan]; x_sh = p.in_shift[x_chan]; }
    // halo piece i: plane column 0 of plane row hr + 8 i of channel hc
    const int hc = tid & (CI_T - 1), hr = tid >> 5;
    const int h_chan = cx0 + ci0 + min(hc, nci - 1);
    const int h_base = (h_chan * p.Hs + hr - 1) * p.Ws - 1;                                        // + 8 i Ws
    const int h_dst = CO_T * GPITCH + hc * XPITCH + hr * PW;                                       // + 8 i PW
    float h_sc = 1.f, h_sh = 0.f;
    if (AFF) { h_sc = p.in_scale[h_chan]; h_sh = p.in_shift[h_chan]; }

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    f32x4 gq[NG4], xq[NX4];
    float hq[NXH];
    if constexpr (LAB & 1) {
#pragma unroll
        for (int i = 0; i < NG4; ++i) gq[i] = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int i = 0; i < NX4; ++i) xq[i] = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int i = 0; i < NXH; ++i) hq[i] = 1.f;
    }
    const float* gbase = p.g;
    const float* xbase = p.x;
    int sy0 = 0;                       // first input row of the tile being staged (2 y0)
    bool g_ok = false, xcol_ok = false, hcol_ok = false;
    auto aim = [&](int tile) {
        const int tx = tile % p.tiles_x;
        const int q = tile / p.tiles_x;
        const int ty = q % p.tiles_y, b = q / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        sy0 = 2 * y0;
        gbase = p.g + ((size_t)b * p.Cy * p.H + y0) * p.W + x0;
        xbase = p.x + ((size_t)b * p.Cx * p.Hs + 2 * y0) * p.Ws + 2 * x0;
        const bool live = tile < p.n_tiles;
        g_ok = live && y0 + grow < p.H && x0 + 4 * gk < p.W;
        xcol_ok = live && 2 * x0 + 4 * xk < p.Ws;
        hcol_ok = live && x0 > 0;
    };
    // plane row r of the tile being staged lies inside the image (and inside the plane: the last piece of a two-row role)
    auto row_ok = [&](int r) { return r < PH && (unsigned)(sy0 + r - 1) < (unsigned)p.Hs; };
    auto load_piece = [&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (LAB & 1) return;
        if constexpr (j < NG4) {
            gq[j] = *reinterpret_cast<const f32x4*>(g_ok ? gbase + g_off[j] : p.g);
        } else if constexpr (j < NG4 + NX4) {
            constexpr int i = j - NG4;
            const bool ok = xcol_ok && row_ok(RS * i + xr);
            xq[i] = *reinterpret_cast<const f32x4*>(ok ? xbase + (x_base + RS * i * p.Ws) : p.x);
        } else if constexpr (j < NL) {
            constexpr int i = j - NG4 - NX4;
            const bool ok = hcol_ok && row_ok(hr + 8 * i);
            hq[i] = *(ok ? xbase + (h_base + 8 * i * p.Ws) : p.x);
        }
    };
    auto store_piece = [&](float* buf, auto s_) {
        constexpr int s = decltype(s_)::value;
        if constexpr (LAB & 2) return;
        if constexpr (s < 4 * NG4) {
            constexpr int i = s / 4, j = s % 4;
            buf[g_dst + 16 * i * GPITCH + j] = g_ok ? gq[i][j] : 0.f;         // only the PIXEL axis (the contraction) needs zeros
        } else if constexpr (s < 4 * NG4 + 4 * NX4) {
            constexpr int i = (s - 4 * NG4) / 4, j = (s - 4 * NG4) % 4;
            float v = xq[i][j];
            if (AFF) v = fmaxf(v * x_sc + x_sh, 0.f);
            // (a two-row role's last piece: plane row PH does not exist -- those lanes hit the dump slot, no branch)
            const int dst = (RS * i + RS - 1 < PH || RS * i + xr < PH) ? x_dst + RS * i * PW + j : BUF - 1;
            buf[dst] = (xcol_ok && row_ok(RS * i + xr)) ? v : 0.f;
        } else if constexpr (s < NS) {
            constexpr int i = s - 4 * NG4 - 4 * NX4;
            float v = hq[i];
            if (AFF) v = fmaxf(v * h_sc + h_sh, 0.f);
            const int dst = (8 * i + 7 < PH || hr + 8 * i < PH) ? h_dst + 8 * i * PW : BUF - 1;
            buf[dst] = (hcol_ok && row_ok(hr + 8 * i)) ? v : 0.f;
        }
    };

    int tile = blockIdx.z;
    int cur = 0;
    if (tile < p.n_tiles) {                  // first tile: staged the serial way
        aim(tile);
        wg_static_for<0, NL>([&](auto j_) { load_piece(j_); });
        wg_static_for<0, NS>([&](auto s_) { store_piece(smem, s_); });
    }
    __syncthreads();

    auto run_tile = [&](auto pipe_) {
        constexpr bool PIPE = decltype(pipe_)::value;
        const float* cbuf = smem + cur * BUF;
        float* nbuf = smem + (cur ^ 1) * BUF;
        const volatile wg_lds_f32* ga = (const volatile wg_lds_f32*)(cbuf + (wave * 32 + l32) * GPITCH + half);
        const volatile wg_lds_f32* xb = (const volatile wg_lds_f32*)(cbuf + CO_T * GPITCH + l32 * XPITCH + 2 * half);
        float fa[2], fb[2][TAPS];
#define SPK_W2_FRAG(st_, slot_)                                                                               \
    {                                                                                                         \
        constexpr int px_ = (2 * (st_)) & (TW - 1), py_ = (2 * (st_)) / TW;                                   \
        fa[slot_] = (LAB & 8) ? 1.f : ga[2 * (st_)];                                                          \
        _Pragma("unroll") for (int t = 0; t < TAPS; ++t)                                                      \
            fb[slot_][t] = (LAB & 8) ? 1.f : xb[(2 * py_ + t / 3) * PW + 2 * px_ + t % 3];                    \
    }
        SPK_W2_FRAG(0, 0);
        wg_static_for<0, STEPS>([&](auto s_) {
            constexpr int st = decltype(s_)::value;
            if constexpr (PIPE && st == HS) __builtin_amdgcn_sched_barrier(0);             // see wgrad3x3_wide_kernel
            if constexpr (st + 1 < STEPS) SPK_W2_FRAG(st + 1, (st + 1) & 1);
            constexpr int nl = (PIPE && st < HS) ? ((PL * st + PL <= NL) ? PL : (PL * st < NL ? NL - PL * st : 0)) : 0;
            constexpr int s0 = PS * (st - HS);
            constexpr int ns = (PIPE && st >= HS) ? ((s0 + PS <= NS) ? PS : (s0 < NS ? NS - s0 : 0)) : 0;
            if constexpr (nl > 0) wg_static_for<PL * st, PL * st + nl>([&](auto j_) { load_piece(j_); });
            if constexpr (ns > 0) wg_static_for<s0, s0 + ns>([&](auto q_) { store_piece(nbuf, q_); });
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                if constexpr (LAB & 4) acc[t][st & 15] += fa[st & 1] * fb[st & 1][t];
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1], fb[st & 1][t], acc[t], 0, 0, 0);
            }
            if constexpr (LAB != 0 && LAB != 32) return;
            if constexpr (st + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, TAPS + 1, 0);
            if constexpr (LAB == 32) return;
            constexpr int np = nl > 0 ? nl : ns;
            if constexpr (np > 0) {
                constexpr int per = TAPS / (np + 1) > 0 ? TAPS / (np + 1) : 1;
                constexpr int groups = np < TAPS ? np : TAPS - 1;
                wg_static_for<0, groups>([&](auto k_) {
                    constexpr int k = decltype(k_)::value;
                    // pieces beyond the MFMA count share the last slots
                    constexpr int cnt = (k + 1 == groups) ? np - (groups - 1) : 1;
                    __builtin_amdgcn_sched_group_barrier(0x8, per, 0);
                    __builtin_amdgcn_sched_group_barrier(nl > 0 ? 0x20 : 0x200, cnt, 0);
                });
                __builtin_amdgcn_sched_group_barrier(0x8, TAPS - per * groups, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x8, TAPS, 0);
            }
        });
#undef SPK_W2_FRAG
    };

    // ONE loop body: after the last tile the staging still runs, with every piece masked off (it reads the tensors' first
    // floats and fills the idle buffer with zeros).  With a second, staging-free copy of the tile for that case the
    // accumulators cross the if / else merge in VGPRs: 144 v_accvgpr_write before and 144 v_accvgpr_read after EVERY tile's
    // 288 MFMAs (and twice the code).
    for (; tile < p.n_tiles; tile += gridDim.z) {
        aim(tile + (int)gridDim.z);
        run_tile(std::true_type{});
        __syncthreads();                      // every wave is done with `cur`, and the other buffer is complete
        cur ^= 1;
    }

    // ---- partial block -> slab [slab][co][tap][ci] (ci contiguous: 128-B stores per half wave) ----
    float* out = p.slabs + (size_t)blockIdx.z * p.Cy * TAPS * p.Cin;
    const int ci = ci0 + l32;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < co_end && ci < p.Cin) out[((size_t)co * TAPS + t) * p.Cin + ci] = acc[t][r];
        }
}

// ---- 3x3 stride-1 weight gradient of a conv that reads the BILINEAR x2 UPSAMPLING of x (styleganv1.py:624-625) ---------
// dW[co][ci][tap] = sum g[b,co,y,x] * up(x)[b,ci,y+ky-1,x+kx-1] without up(x) ever existing in HBM (round 1 materialised it:
// 268 MB written and read back at [8,128,256,256]).  Same block, tile, LDS images and arithmetic order as
// wgrad3x3_wide_kernel<1,1>; what differs is where the 18 x 6 input plane comes from: the tile's low-resolution SOURCE patch
// (4 x 10 pixels per channel, 2.7x fewer bytes than the plane) is loaded with 16-byte row loads two tiles ahead into a small
// fp32 LDS buffer, and the plane of the NEXT tile is interpolated LDS -> LDS while this tile's MFMAs run -- one element per
// thread and k-step, its tap offsets and weights wave-uniform scalars (a wave owns a quarter of the plane's positions, a lane
// one channel).  An earlier attempt formed the four bilinear taps with global gathers inside the staging and lost 2x to the
// materialised form: those 4 loads per element could not be prefetched; LDS reads need no prefetch.
constexpr int US_PW = 10, US_PH = 4, US_PITCH = US_PW * US_PH + 1, US_FLOATS = 64 * US_PITCH;

// MODE = WG_BATCH_SCALE: the StyleGAN2 variant's x2 layers -- the upsampling is upfirdn2d(up = 2, [1,3,3,1]), i.e. the SAME parity
// taps (.25 / .75) over a ZERO-bordered source (the bilinear form replicates the edge), the source is x * s[b,ci] and the
// gradient g * d'[b,co]: all three differences live in the staging (zeros instead of clamped loads, two multiplies).
template <int MODE>
__global__ __launch_bounds__(256) void wgrad3x3_up_kernel(const WgradArgs p) {
    constexpr bool BSC = MODE == WG_BATCH_SCALE;
    using SH = WideShape<1, 1>;
    constexpr int TAPS = 9, CO_T = 64, CI_T = 64, PW = SH::PW, PLANE = SH::PLANE, GPITCH = SH::GPITCH, XPITCH = SH::XPITCH, BUF = SH::BUF;
    constexpr int NG4 = 4, STEPS = 32, HS = 16, EPT = CI_T * PLANE / 256;       // 27 plane elements per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sbuf = smem + 2 * BUF;                  // two source-patch buffers behind the two tile buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wco = wave & 1, wci = wave >> 1;

    const int co0 = blockIdx.x * CO_T;
    const int grp = co0 / p.Cout;
    const int co_end = (grp + 1) * p.Cout;
    const int cx0 = grp * p.gin;
    const int ci0 = blockIdx.y * CI_T;
    const int nci = min(CI_T, p.Cin - ci0);

    // ---- staging roles ----
    const int gk = tid & 3, grow = (tid >> 2) & 3;
    int g_off[NG4], g_ch[NG4];
    float g_sc[NG4];
#pragma unroll
    for (int i = 0; i < NG4; ++i) {
        g_ch[i] = min(co0 + (tid >> 4) + 16 * i, co_end - 1);
        g_off[i] = (g_ch[i] * p.H + grow) * p.W + 4 * gk;
        g_sc[i] = 1.f;
    }
    const int g_dst = (tid >> 4) * GPITCH + grow * 16 + 4 * gk;
    // source patch: piece i in {0, 1} = source row rh + 2 i of channel sc: float4 sk (columns x0/2 + 4 sk ..) and the halo column
    // x0/2 - 1 (sk = 0) / x0/2 + 8 (sk = 1)
    const int sk = tid & 1, sc = (tid >> 1) & 63, rh = tid >> 7;
    const int s_ch = cx0 + ci0 + min(sc, nci - 1);
    const int s_chan = s_ch * p.Hs;
    const int s_dst4 = sc * US_PITCH + rh * US_PW + 1 + 4 * sk;      // + 2 i US_PW + j
    const int s_dsth = sc * US_PITCH + rh * US_PW + (sk ? 9 : 0);    // + 2 i US_PW

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // per-tile state: (g) the gradient tile being staged; (s) the source patch being staged; (i) the tile being interpolated
    f32x4 gq[NG4], sq[2];
    float sh[2];
    const float* gbase = p.g;
    const float* sbase = p.x;
    bool g_ok = false;
    int s_y0 = 0, i_y0 = 0, i_x0 = 0;
    auto origin = [&](int tile, int& b, int& y0, int& x0) {
        const int tx = tile % p.tiles_x;
        const int q = tile / p.tiles_x;
        b = q / p.tiles_y; y0 = (q % p.tiles_y) * 4; x0 = tx * 16;
    };
    auto aim_g = [&](int tile) {
        int b, y0, x0;
        origin(tile, b, y0, x0);
        gbase = p.g + ((size_t)b * p.Cy * p.H + y0) * p.W + x0;
        g_ok = tile < p.n_tiles && y0 + grow < p.H && x0 + 4 * gk < p.W;
        if constexpr (BSC) {
            const int bb = tile < p.n_tiles ? b : 0;
#pragma unroll
            for (int i = 0; i < NG4; ++i) g_sc[i] = p.g_scale[(size_t)bb * p.Cy + g_ch[i]];
        }
    };
    // The source patch is REPLICATE-padded (rows / columns outside the image take the nearest edge pixel: loads with clamped
    // coordinates): torch's bilinear clamps its second tap at the border, which is the plain parity weights (.75, .25) applied
    // to a replicated neighbour -- so every plane element is interpolated with position-independent weights and no branch; what
    // remains of the border is the conv's ZERO padding of the x2 image, a per-tile row / column mask.
    int s_x0 = 0;
    bool s4_out = false;
    float s_sc = 1.f;                                                 // BSC: s[b, channel] of the patch being staged
    auto aim_s = [&](int tile) {
        int b, y0, x0;
        origin(tile, b, y0, x0);
        s_y0 = (y0 >> 1) - 1;                                         // first source row of the patch (may be -1)
        s_x0 = x0 >> 1;
        sbase = p.x + ((size_t)b * p.Cx * p.Hs) * p.Ws;
        s4_out = s_x0 + 4 * sk >= p.Ws;                                // this thread's vector lies right of the image: replicate
        if constexpr (BSC) s_sc = p.in_scale[(size_t)b * p.Cx + s_ch];  // (aim_s is only ever called with a live tile)
    };
    unsigned i_rows = 0, i_cols = 0;                                  // plane rows / columns inside the x2 image (bit r / bit c)
    auto aim_i = [&](int tile) {
        int b;
        origin(tile, b, i_y0, i_x0);
        i_rows = 0; i_cols = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) i_rows |= ((unsigned)(i_y0 - 1 + r) < (unsigned)p.H ? 1u : 0u) << r;
#pragma unroll
        for (int c = 0; c < PW; ++c) i_cols |= ((unsigned)(i_x0 - 1 + c) < (unsigned)p.W ? 1u : 0u) << c;
    };
    auto s_row = [&](int i) { return min(max(s_y0 + rh + 2 * i, 0), p.Hs - 1); };
    auto load_g = [&](auto j_) { constexpr int j = decltype(j_)::value; gq[j] = *reinterpret_cast<const f32x4*>(g_ok ? gbase + g_off[j] : p.g); };
    auto store_g = [&](float* buf, auto s_) {
        constexpr int s = decltype(s_)::value, i = s / 4, j = s % 4;
        buf[g_dst + 16 * i * GPITCH + j] = g_ok ? (BSC ? gq[i][j] * g_sc[i] : gq[i][j]) : 0.f;
    };
    auto load_s = [&](auto j_) {              // j: 0, 1 = float4 of piece j; 2, 3 = halo of piece j - 2
        constexpr int j = decltype(j_)::value, i = j & 1;
        const float* row = sbase + (size_t)(s_chan + s_row(i)) * p.Ws;
        if constexpr (j < 2) sq[i] = *reinterpret_cast<const f32x4*>(row + min(s_x0 + 4 * sk, p.Ws - 4));     // Ws % 4 == 0
        else sh[i] = row[min(max(s_x0 + (sk ? 8 : -1), 0), p.Ws - 1)];
    };
    auto store_s = [&](float* sb, auto s_) {  // s: 0..7 = element s % 4 of float4 piece s / 4; 8, 9 = halo of piece s - 8
        constexpr int s = decltype(s_)::value;
        if constexpr (s < 8) {
            constexpr int i = s / 4, j = s % 4;
            if constexpr (BSC) {     // zero border: a source row / vector outside the image contributes nothing
                const bool in = (unsigned)(s_y0 + rh + 2 * i) < (unsigned)p.Hs && !s4_out;
                sb[s_dst4 + 2 * i * US_PW + j] = in ? sq[i][j] * s_sc : 0.f;
            } else {
                sb[s_dst4 + 2 * i * US_PW + j] = s4_out ? sq[i][3] : sq[i][j];     // (the clamped load fetched the image's last vector)
            }
        } else {
            constexpr int i = s - 8;
            if constexpr (BSC) {
                const bool in = (unsigned)(s_y0 + rh + 2 * i) < (unsigned)p.Hs && (unsigned)(s_x0 + (sk ? 8 : -1)) < (unsigned)p.Ws;
                sb[s_dsth + 2 * i * US_PW] = in ? sh[i] * s_sc : 0.f;
            } else {
                sb[s_dsth + 2 * i * US_PW] = sh[i];
            }
        }
    };
    // plane element e = wave * 27 + i of channel `lane` (a wave owns a quarter of the 108 positions): everything but the
    // channel is wave-uniform, so the tap offsets and weights live in scalar registers
    const float* const s_lane_base = sbuf + lane * US_PITCH;          // this lane's channel in source buffer 0
    // an element's four source taps are READ one k-step before they are combined and written (as the MFMA fragments are): with
    // read, combine and write in one step every wave waited ~200 cycles on its own LDS reads 27 times a tile
    float iq[4];
    auto interp_read = [&](int sbo, int i) {                           // sbo: float offset of the source buffer (0 / US_FLOATS)
        const int e = wave * EPT + i;                                  // uniform
        const int r = (e * 3641) >> 16, c = e - r * PW;                 // e / 18 for e < 108
        // x2 pixel (y0 - 1 + r, x0 - 1 + c), y0 and x0 even: first tap at patch (r >> 1, c >> 1)
        const float* q = s_lane_base + sbo + (r >> 1) * US_PW + (c >> 1);
        iq[0] = q[0]; iq[1] = q[1]; iq[2] = q[US_PW]; iq[3] = q[US_PW + 1];
    };
    auto interp_write = [&](float* buf, int i) {
        const int e = wave * EPT + i;
        const int r = (e * 3641) >> 16, c = e - r * PW;
        const float ly1 = (r & 1) ? 0.75f : 0.25f, lx1 = (c & 1) ? 0.75f : 0.25f;    // second-tap weight .25 on even r / c
        const float t = (1.f - ly1) * ((1.f - lx1) * iq[0] + lx1 * iq[1]) + ly1 * ((1.f - lx1) * iq[2] + lx1 * iq[3]);
        const bool inside = ((i_rows >> r) & (i_cols >> c) & 1u) != 0;  // uniform
        buf[CO_T * GPITCH + lane * XPITCH + e] = inside ? t : 0.f;
    };
    auto interp = [&](float* buf, int sbo, int i) { interp_read(sbo, i); interp_write(buf, i); };
    const int stride = (int)gridDim.z;
    int tile = blockIdx.z;
    int cur = 0, par = 0;                      // tile buffer in use; source buffer holding the patch of tile `tile + stride`... see below
    // ---- prologue: S(k = 0) -> sbuf[0], S(k = 1) -> sbuf[1], G(0) -> buf 0; barrier; plane(0) by interpolation; barrier ----
    if (tile < p.n_tiles) {
        aim_s(tile);
        wg_static_for<0, 4>([&](auto j_) { load_s(j_); });
        wg_static_for<0, 10>([&](auto s_) { store_s(sbuf, s_); });
        if (tile + stride < p.n_tiles) {
            aim_s(tile + stride);
            wg_static_for<0, 4>([&](auto j_) { load_s(j_); });
            wg_static_for<0, 10>([&](auto s_) { store_s(sbuf + US_FLOATS, s_); });
        }
        aim_g(tile);
        wg_static_for<0, NG4>([&](auto j_) { load_g(j_); });
        wg_static_for<0, 4 * NG4>([&](auto s_) { store_g(smem, s_); });
        __syncthreads();
        aim_i(tile);
        for (int i = 0; i < EPT; ++i) interp(smem, 0, i);
    }
    __syncthreads();

    // iteration k computes tile k out of buf[cur]; meanwhile G(k+1) is loaded / stored into the other tile buffer, the plane of
    // tile k+1 is interpolated into it from sbuf[(k+1) & 1], and the source patch of tile k+2 is loaded and stored into
    // sbuf[k & 1] (whose previous content, the patch of tile k, was last read during iteration k - 1)
    auto run_tile = [&](auto has1_, auto has2_) {
        constexpr bool H1 = decltype(has1_)::value, H2 = decltype(has2_)::value;
        const float* cbuf = smem + cur * BUF;
        float* nbuf = smem + (cur ^ 1) * BUF;
        const int s_next = (par ^ 1) * US_FLOATS;                      // patch of tile k+1 (float offset behind sbuf)
        float* s_fill = sbuf + par * US_FLOATS;                       // patch of tile k+2 goes here
        const volatile wg_lds_f32* ga = (const volatile wg_lds_f32*)(cbuf + (wco * 32 + l32) * GPITCH + half);
        const volatile wg_lds_f32* xb = (const volatile wg_lds_f32*)(cbuf + CO_T * GPITCH + (wci * 32 + l32) * XPITCH + half);
        float fa[2], fb[2][TAPS];
#define SPK_WU_FRAG(st_, slot_)                                                                               \
    {                                                                                                         \
        constexpr int px_ = (2 * (st_)) & 15, py_ = (2 * (st_)) >> 4;                                         \
        fa[slot_] = ga[2 * (st_)];                                                                            \
        _Pragma("unroll") for (int t = 0; t < TAPS; ++t) fb[slot_][t] = xb[(py_ + t / 3) * PW + px_ + t % 3];  \
    }
        SPK_WU_FRAG(0, 0);
        wg_static_for<0, STEPS>([&](auto s_) {
            constexpr int st = decltype(s_)::value;
            if constexpr (H1 && st == HS) __builtin_amdgcn_sched_barrier(0);               // see wgrad3x3_wide_kernel
            if constexpr (st + 1 < STEPS) SPK_WU_FRAG(st + 1, (st + 1) & 1);
            if constexpr (H1 && st < NG4) load_g(std::integral_constant<int, st < NG4 ? st : 0>{});
            if constexpr (H2 && st >= NG4 && st < NG4 + 4) load_s(std::integral_constant<int, (st >= NG4 && st < NG4 + 4) ? st - NG4 : 0>{});
            if constexpr (H1 && st >= 1 && st <= EPT) interp_write(nbuf, st - 1);
            if constexpr (H1 && st < EPT) interp_read(s_next, st);
            if constexpr (H1 && st >= HS) store_g(nbuf, std::integral_constant<int, st >= HS ? st - HS : 0>{});
            if constexpr (H2 && st >= HS && st < HS + 10) store_s(s_fill, std::integral_constant<int, (st >= HS && st < HS + 10) ? st - HS : 0>{});
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1], fb[st & 1][t], acc[t], 0, 0, 0);
            // order: the LDS reads (next step's fragments, the next element's taps) first, then the MFMAs with the rest of the
            // step's staging (loads, LDS stores, the interpolation arithmetic) spread between them in three parts
            if constexpr (st + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, TAPS + 1 + ((H1 && st < EPT) ? 4 : 0), 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x2 | 0x4 | 0x20 | 0x200, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x2 | 0x4 | 0x20 | 0x200, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
        });
#undef SPK_WU_FRAG
    };

    for (; tile < p.n_tiles; tile += stride) {
        // one loop body (see wgrad3x3_wide_kernel): past the last tile the gradient pieces are masked off, the patch of the
        // last tile is fetched again and the idle buffers are filled with values nobody reads
        aim_g(tile + stride);
        aim_i(tile + stride);
        aim_s(min(tile + 2 * stride, p.n_tiles - 1));
        run_tile(std::true_type{}, std::true_type{});
        __syncthreads();
        cur ^= 1;
        par ^= 1;
    }

    // ---- partial block -> slab [slab][co][tap][ci] ----
    float* out = p.slabs + (size_t)blockIdx.z * p.Cy * TAPS * p.Cin;
    const int ci = ci0 + wci * 32 + l32;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < co_end && ci < p.Cin) out[((size_t)co * TAPS + t) * p.Cin + ci] = acc[t][r];
        }
}

// dW[co][ci][tap] (+)= scale * sum_slab slabs[slab][co][tap][ci]   (fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                          int n_slabs, int Cout_all, int Cin, int taps, float scale,
                                                          int accumulate, int fold) {
    // fold > 1: output channel co also takes the slabs' channels co + Cout, co + 2 Cout, ... (the same conv on other images)
    const int Cout = Cout_all / fold;
    const size_t total = (size_t)Cout * Cin * taps, slab_stride = (size_t)Cout_all * Cin * taps;
    // walk the slabs in THEIR order ([co][tap][ci]: coalesced reads of n_slabs x total floats) and scatter the
    // (n_slabs times smaller) result into [co][ci][tap]
    for (size_t src = (size_t)blockIdx.x * blockDim.x + threadIdx.x; src < total; src += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(src % Cin);
        const int tap = (int)((src / Cin) % taps);
        const int co = (int)(src / ((size_t)Cin * taps));
        // four interleaved partial sums (slab s goes to sum s % 4): still one fixed order, but four loads in flight
        float v = 0.f;
        for (int f = 0; f < fold; ++f) {
            const float* sl = slabs + (size_t)f * total + src;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            int s = 0;
            for (; s + 4 <= n_slabs; s += 4) {
                const float a0 = sl[(size_t)s * slab_stride], a1 = sl[(size_t)(s + 1) * slab_stride];
                const float a2 = sl[(size_t)(s + 2) * slab_stride], a3 = sl[(size_t)(s + 3) * slab_stride];
                v0 += a0; v1 += a1; v2 += a2; v3 += a3;
            }
            for (; s < n_slabs; ++s) v0 += sl[(size_t)s * slab_stride];
            v += (v0 + v1) + (v2 + v3);
        }
        v *= scale;
        const size_t idx = ((size_t)co * Cin + ci) * taps + tap;
        dw[idx] = accumulate ? dw[idx] + v : v;
    }
}

// The same sum, four input channels per thread (Cin % 4 == 0): 16-byte loads and EIGHT slabs in flight per thread.  The
// dword form above waits for every round of four loads before it issues the next -- with 16-42 slabs that is 4-10 memory
// latencies in a row, 2.3 TB/s on slabs that mostly still sit in the memory-side cache; this one runs at 13 -> ~7 us per
// trunk / decoder layer, ~65 (G step) to ~100 (D step) times per training step.  Same summation order as the dword form.
__global__ __launch_bounds__(256) void wgrad_reduce_vec_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                              int n_slabs, int Cout_all, int Cin, int taps, float scale,
                                                              int accumulate, int fold) {
    const int Cout = Cout_all / fold;
    const size_t total = (size_t)Cout * Cin * taps, slab_stride = (size_t)Cout_all * Cin * taps;
    const int cin4 = Cin >> 2;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total / 4; q += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(q % cin4) * 4;
        const int tap = (int)((q / cin4) % taps);
        const int co = (int)(q / ((size_t)cin4 * taps));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int f = 0; f < fold; ++f) {
            const float* sl = slabs + (size_t)f * total + 4 * q;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
            int s = 0;
            for (; s + 8 <= n_slabs; s += 8) {
                f32x4 a[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const f32x4*>(sl + (size_t)(s + j) * slab_stride);
                v0 += a[0]; v1 += a[1]; v2 += a[2]; v3 += a[3];
                v0 += a[4]; v1 += a[5]; v2 += a[6]; v3 += a[7];
            }
            if (s + 4 <= n_slabs) {
                f32x4 a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f32x4*>(sl + (size_t)(s + j) * slab_stride);
                v0 += a[0]; v1 += a[1]; v2 += a[2]; v3 += a[3];
                s += 4;
            }
            for (; s < n_slabs; ++s) v0 += *reinterpret_cast<const f32x4*>(sl + (size_t)s * slab_stride);
            v += (v0 + v1) + (v2 + v3);
        }
        v *= scale;
        const size_t idx = ((size_t)co * Cin + ci) * taps + tap;
        if (taps == 1) {
            f32x4* o = reinterpret_cast<f32x4*>(dw + idx);
            *o = accumulate ? *o + v : v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dw[idx + (size_t)j * taps] = accumulate ? dw[idx + (size_t)j * taps] + v[j] : v[j];
        }
    }
}

// MANY slabs of a SMALL gradient (the one-round grids of the Winograd weight gradient: 256 slabs of a 64 x 64 x 9 block, 37.7 MB for
// 36 workgroups of the kernel above -- 256 dependent rounds each): 16 threads share an output element group, thread (e, l) sums the
// slabs l, l + 16, ... (all of them in flight), the 16 partial sums meet in LDS and are added in lane order -- a fixed order, so the
// result stays bitwise reproducible (it is NOT the order of the kernels above: a layer always takes the same one of the two).
constexpr int RD_SL = 16, RD_EL = 16;
__global__ __launch_bounds__(256) void wgrad_reduce_deep_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int n_slabs,
                                                               int Cout_all, int Cin, int taps, float scale, int accumulate, int fold) {
    __shared__ f32x4 part[RD_SL][RD_EL];
    const int Cout = Cout_all / fold;
    const size_t total = (size_t)Cout * Cin * taps, slab_stride = (size_t)Cout_all * Cin * taps;
    const int cin4 = Cin >> 2;
    const int e = threadIdx.x & (RD_EL - 1), l = threadIdx.x / RD_EL;
    for (size_t q0 = (size_t)blockIdx.x * RD_EL; q0 < total / 4; q0 += (size_t)gridDim.x * RD_EL) {      // (uniform trip count: barriers inside)
        const size_t q = q0 + e;
        const bool live = q < total / 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (live)
            for (int f = 0; f < fold; ++f) {
                const float* sl = slabs + (size_t)f * total + 4 * q;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                int s_ = l;
                for (; s_ + 7 * RD_SL < n_slabs; s_ += 8 * RD_SL) {
                    f32x4 a[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const f32x4*>(sl + (size_t)(s_ + j * RD_SL) * slab_stride);
#pragma unroll
                    for (int j = 0; j < 8; j += 2) { v0 += a[j]; v1 += a[j + 1]; }
                }
                for (; s_ < n_slabs; s_ += RD_SL) v0 += *reinterpret_cast<const f32x4*>(sl + (size_t)s_ * slab_stride);
                v += v0 + v1;
            }
        part[l][e] = v;
        __syncthreads();
        if (l == 0 && live) {
            f32x4 t = part[0][e];
#pragma unroll
            for (int j = 1; j < RD_SL; ++j) t += part[j][e];
            t *= scale;
            const int ci = (int)(q % cin4) * 4;
            const int tap = (int)((q / cin4) % taps);
            const int co = (int)(q / ((size_t)cin4 * taps));
            const size_t idx = ((size_t)co * Cin + ci) * taps + tap;
            if (taps == 1) {
                f32x4* o = reinterpret_cast<f32x4*>(dw + idx);
                *o = accumulate ? *o + t : t;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dw[idx + (size_t)j * taps] = accumulate ? dw[idx + (size_t)j * taps] + t[j] : t[j];
            }
        }
        __syncthreads();
    }
}

// slabs -> dW on `stream`, after the kernel that wrote them
inline int launch_wgrad_reduce(hipStream_t stream, const float* slabs, float* dw, int n_slabs, int Cout_all, int Cin, int taps,
                               float scale, int accumulate, int fold) {
    const size_t slab_floats = (size_t)Cout_all * Cin * taps;
    const bool vec = Cin % 4 == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0;
    if (vec && n_slabs >= 32 && slab_floats / fold / 4 / 256 < 512) {
        const unsigned blocks = (unsigned)std::min<size_t>((slab_floats / fold / 4 + RD_EL - 1) / RD_EL, 4096);
        hipLaunchKernelGGL(wgrad_reduce_deep_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, stream, slabs, dw, n_slabs, Cout_all, Cin,
                           taps, scale, accumulate, fold);
    } else if (vec) {
        const unsigned blocks = (unsigned)std::min<size_t>((slab_floats / fold / 4 + 255) / 256, 4096);
        hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, stream, slabs, dw, n_slabs, Cout_all, Cin,
                           taps, scale, accumulate, fold);
    } else {
        const unsigned blocks = (unsigned)std::min<size_t>((slab_floats + 255) / 256, 2048);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slabs, dw, n_slabs, Cout_all, Cin, taps, scale,
                           accumulate, fold);
    }
    return spk::check_launch("wgrad_reduce_kernel");
}

struct WGeom {
    int TW, TH, TB, tiles_x, tiles_y, n_tiles, splits, n_slabs;
    size_t lds_bytes;
};

template <int KH, int KW, int S>
WGeom wgeom(int B, int Cin, int Cout, int H, int W, int want_splits) {
    using SH = WShape<KH, KW, S>;
    WGeom g;
    g.TW = std::min(32, spk::pow2_ceil(W));
    if (SH::CS > 1 && H >= 4) g.TW = std::min(g.TW, 16);     // 16x4 tiles: 108 staged positions (<= 128) instead of 136
    g.TH = std::min(SH::PIX_T / g.TW, spk::pow2_ceil(H));
    g.TB = SH::PIX_T / (g.TW * g.TH);
    // keep the staged input tile within LDS: shrink the image group first, then the rows
    auto lds = [&]() {
        const size_t plane = (size_t)((g.TH - 1) * S + KH) * ((g.TW - 1) * S + KW);
        return (SH::CO_T * (SH::PIX_T + 1) + SH::CI_T * ((g.TB * plane) | 1)) * sizeof(float);
    };
    auto plane_elems = [&]() { return g.TB * ((g.TH - 1) * S + KH) * ((g.TW - 1) * S + KW); };
    while ((lds() > 100 * 1024 || plane_elems() > SH::NPT * SH::KX) && g.TB > 1) g.TB >>= 1;   // idle pixel groups
    while (plane_elems() > SH::NPT * SH::KX && g.TH > 1) g.TH >>= 1;
    g.lds_bytes = lds();
    g.tiles_x = spk::ceil_div(W, g.TW);
    g.tiles_y = spk::ceil_div(H, g.TH);
    g.n_tiles = g.tiles_x * g.tiles_y * spk::ceil_div(B, g.TB);
    const int blocks = spk::ceil_div(Cout, SH::CO_T) * spk::ceil_div(Cin, SH::CI_T) * (SH::ROWPASS ? KH : 1);
    int sp = want_splits > 0 ? want_splits : std::max(1, 512 / blocks);     // 2 rounds of one workgroup per CU
    g.splits = std::max(1, std::min(sp, g.n_tiles));
    g.n_slabs = g.splits * SH::WPX;
    return g;
}

// geometry of the wide form: 16 x 4 (or, tw = 8, 8 x 8) pixel tiles, one workgroup per CU
struct WideGeom { int tiles_x, tiles_y, n_tiles, splits, n_slabs; };
// (SPK_WGRAD_WIDE_SB = 1: the single-buffer form of wgrad3x3_wide_kernel, two workgroups per CU, for the plain and the
// BatchNorm-folded modes; its split rule aims at SPK_WGRAD_WIDE_SB_TARGET workgroups)
inline bool wide_sb_on() {
    static const bool on = [] { const char* e = getenv("SPK_WGRAD_WIDE_SB"); return e && atoi(e) != 0; }();
    return on;
}
inline WideGeom wide_geom(int co_t, int ci_t, int B, int Cin, int Cout_all, int H, int W, int want_splits, int tw = 16, bool sb = false) {
    WideGeom g;
    g.tiles_x = spk::ceil_div(W, tw);
    g.tiles_y = spk::ceil_div(H, 64 / tw);
    g.n_tiles = g.tiles_x * g.tiles_y * B;
    const int blocks = spk::ceil_div(Cout_all, co_t) * spk::ceil_div(Cin, ci_t);
    static const int target1 = [] { const char* e = getenv("SPK_WGRAD_WIDE_TARGET"); return e ? atoi(e) : 256; }();
    static const int target2 = [] { const char* e = getenv("SPK_WGRAD_WIDE_SB_TARGET"); return e ? atoi(e) : 512; }();
    const int target = sb ? target2 : target1;
    const int sp = want_splits > 0 ? want_splits : std::max(1, target / blocks);
    g.splits = std::max(1, std::min(sp, g.n_tiles));
    g.n_slabs = g.splits;
    return g;
}
// which block shape the wide form uses for a problem: 0 = 128co x 64ci, 1 = 64co x 128ci, 2 = 64co x 64ci,
// -1 = not applicable (Cout per group must fill whole co blocks when grouped)
inline int wide_variant(int groups, int Cout, int Cin) {
    const int G = groups > 1 ? groups : 1;
    // (the two-tile-set shapes hold 288 accumulator registers per lane; hipcc keeps MFMA accumulators in the 256 AGPRs only
    // and spills the rest, so they are not instantiated: 0 and 1 are never returned)
    if (Cout % 64 == 0 || G == 1) return 2;
    return -1;
}
inline bool wide_takes(const spk_wgrad_desc* d) {
    static const bool allow = [] { const char* e = getenv("SPK_WGRAD_WIDE"); return !e || atoi(e) != 0; }();
    const int G = d->groups > 1 ? d->groups : 1;
    const auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return allow && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->W >= 8 && d->W % 4 == 0 && d->H >= 4 && aligned(d->g) && aligned(d->x) &&
           (long long)G * d->Cout * d->H * d->W < (1ll << 31) && ((long long)d->group_in_stride * (G - 1) + d->Cin) * d->H * d->W < (1ll << 31) &&
           wide_variant(d->groups, d->Cout, d->Cin) >= 0;
}

template <int TW, int MODE>
int run_wgrad_wide(const spk_wgrad_desc* d, hipStream_t stream) {
    using SH = WideShapeT<TW>;
    const int G = d->groups > 1 ? d->groups : 1;
    constexpr bool HAS_SB = MODE == WG_PLAIN || MODE == WG_AFFINE_RELU;      // (the modulated form needs > 256 registers)
    const bool sb = HAS_SB && wide_sb_on();
    const WideGeom g = wide_geom(SH::CO_T, SH::CI_T, d->B, d->Cin, G * d->Cout, d->H, d->W, d->splits, TW, sb);
    const size_t slab_floats = (size_t)G * d->Cout * d->Cin * 9;
    SPK_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= g.n_slabs * slab_floats * sizeof(float),
                "wgrad: needs a %zu-byte workspace (see spk_conv2d_wgrad_workspace_bytes)", g.n_slabs * slab_floats * sizeof(float));
    WgradArgs a;
    a.g = d->g; a.x = d->x; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.g_scale = d->g_scale; a.slabs = static_cast<float*>(d->workspace);
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W; a.Hs = d->Hin; a.Ws = d->Win;
    a.gin = G > 1 ? d->group_in_stride : d->Cin;
    a.Cx = a.gin * (G - 1) + d->Cin;
    a.Cy = G * d->Cout;
    a.lgTW = spk::ilog2(TW); a.lgTH = spk::ilog2(64 / TW); a.lgTB = 0;
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.n_tiles = g.n_tiles;
    auto kern = &wgrad3x3_wide_kernel<TW, MODE, false>;
    if constexpr (HAS_SB) {
        if (sb) kern = &wgrad3x3_wide_kernel<TW, MODE, true>;
    }
    static bool raised[2] = {false, false};
    if (!raised[sb]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        raised[sb] = true;
    }
    dim3 grid((unsigned)spk::ceil_div(G * d->Cout, SH::CO_T), (unsigned)spk::ceil_div(d->Cin, SH::CI_T), (unsigned)g.splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), (sb ? 1 : 2) * SH::BUF * sizeof(float), stream, a);
    int rc = spk::check_launch("wgrad3x3_wide_kernel");
    if (rc != SPK_OK) return rc;
    return launch_wgrad_reduce(stream, a.slabs, d->dw, g.n_slabs, G * d->Cout, d->Cin, 9, d->scale, d->accumulate ? 1 : 0,
                               d->fold > 1 ? d->fold : 1);
}

// the upsample-folded form: x is the LOW-resolution tensor [B, Cin, H/2, W/2]
inline bool up_takes(const spk_wgrad_desc* d) {
    const int G = d->groups > 1 ? d->groups : 1;
    const auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->W >= 16 && d->W % 8 == 0 && d->H >= 4 && d->H % 2 == 0 && aligned(d->g) &&
           aligned(d->x) && (long long)G * d->Cout * d->H * d->W < (1ll << 31) && (G == 1 || d->Cout % 64 == 0) &&
           ((long long)d->group_in_stride * (G - 1) + d->Cin) * d->Hin * d->Win < (1ll << 31);
}

template <int MODE>
int run_wgrad_up(const spk_wgrad_desc* d, hipStream_t stream) {
    using SH = WideShape<1, 1>;
    const int G = d->groups > 1 ? d->groups : 1;
    const WideGeom g = wide_geom(SH::CO_T, SH::CI_T, d->B, d->Cin, G * d->Cout, d->H, d->W, d->splits);
    const size_t slab_floats = (size_t)G * d->Cout * d->Cin * 9;
    SPK_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= g.n_slabs * slab_floats * sizeof(float),
                "wgrad: needs a %zu-byte workspace (see spk_conv2d_wgrad_workspace_bytes)", g.n_slabs * slab_floats * sizeof(float));
    WgradArgs a;
    a.g = d->g; a.x = d->x; a.in_scale = MODE == WG_BATCH_SCALE ? d->in_scale : nullptr; a.in_shift = nullptr;
    a.g_scale = MODE == WG_BATCH_SCALE ? d->g_scale : nullptr; a.slabs = static_cast<float*>(d->workspace);
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W; a.Hs = d->Hin; a.Ws = d->Win;
    a.gin = G > 1 ? d->group_in_stride : d->Cin;
    a.Cx = a.gin * (G - 1) + d->Cin;
    a.Cy = G * d->Cout;
    a.lgTW = 4; a.lgTH = 2; a.lgTB = 0;
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.n_tiles = g.n_tiles;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_up_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        raised = true;
    }
    dim3 grid((unsigned)spk::ceil_div(G * d->Cout, SH::CO_T), (unsigned)spk::ceil_div(d->Cin, SH::CI_T), (unsigned)g.splits);
    hipLaunchKernelGGL(wgrad3x3_up_kernel<MODE>, grid, dim3(256), (2 * SH::BUF + 2 * US_FLOATS) * sizeof(float), stream, a);
    int rc = spk::check_launch("wgrad3x3_up_kernel");
    if (rc != SPK_OK) return rc;
    return launch_wgrad_reduce(stream, a.slabs, d->dw, g.n_slabs, G * d->Cout, d->Cin, 9, d->scale, d->accumulate ? 1 : 0,
                               d->fold > 1 ? d->fold : 1);
}

// the stride-2 form: 128co x 32ci blocks, 16x4 or 8x8 output-pixel tiles
inline bool s2_takes(const spk_wgrad_desc* d) {
    static const bool allow = [] { const char* e = getenv("SPK_WGRAD_S2"); return !e || atoi(e) != 0; }();
    const int G = d->groups > 1 ? d->groups : 1;
    const auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return allow && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->W >= 8 && d->W % 4 == 0 && d->H >= 2 && d->Hin == 2 * d->H &&
           d->Win == 2 * d->W && aligned(d->g) && aligned(d->x) && d->Cout >= 96 && (G == 1 || d->Cout % 128 == 0) &&
           (long long)G * d->Cout * d->H * d->W < (1ll << 31) &&
           ((long long)d->group_in_stride * (G - 1) + d->Cin) * d->Hin * d->Win < (1ll << 31);
}
inline WideGeom s2_geom(int TW, int B, int Cin, int Cout_all, int H, int W, int want_splits) {
    WideGeom g;
    g.tiles_x = spk::ceil_div(W, TW);
    g.tiles_y = spk::ceil_div(H, 64 / TW);
    g.n_tiles = g.tiles_x * g.tiles_y * B;
    const int blocks = spk::ceil_div(Cout_all, 128) * spk::ceil_div(Cin, 32);
    static const int target = [] { const char* e = getenv("SPK_WGRAD_S2_TARGET"); return e ? atoi(e) : 256; }();
    const int sp = want_splits > 0 ? want_splits : std::max(1, target / blocks);
    g.splits = std::max(1, std::min(sp, g.n_tiles));
    g.n_slabs = g.splits;
    return g;
}

template <int TW, int MODE>
int run_wgrad_s2(const spk_wgrad_desc* d, hipStream_t stream) {
    using SH = S2Shape<TW>;
    const int G = d->groups > 1 ? d->groups : 1;
    const WideGeom g = s2_geom(TW, d->B, d->Cin, G * d->Cout, d->H, d->W, d->splits);
    const size_t slab_floats = (size_t)G * d->Cout * d->Cin * 9;
    SPK_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= g.n_slabs * slab_floats * sizeof(float),
                "wgrad: needs a %zu-byte workspace (see spk_conv2d_wgrad_workspace_bytes)", g.n_slabs * slab_floats * sizeof(float));
    WgradArgs a;
    a.g = d->g; a.x = d->x; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.g_scale = d->g_scale; a.slabs = static_cast<float*>(d->workspace);
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W; a.Hs = d->Hin; a.Ws = d->Win;
    a.gin = G > 1 ? d->group_in_stride : d->Cin;
    a.Cx = a.gin * (G - 1) + d->Cin;
    a.Cy = G * d->Cout;
    a.lgTW = spk::ilog2(TW); a.lgTH = spk::ilog2(64 / TW); a.lgTB = 0;
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.n_tiles = g.n_tiles;
    auto kern = &wgrad3x3_s2_kernel<TW, MODE>;
#ifdef SPK_WGRAD_S2_LAB
    if constexpr (TW == 16 && MODE == WG_PLAIN) {
        const char* e = getenv("SPK_WG_LAB");
        switch (e ? atoi(e) : 0) {
            case 16: kern = &wgrad3x3_s2_kernel<TW, MODE, 16>; break;
            case 32: kern = &wgrad3x3_s2_kernel<TW, MODE, 32>; break;
            default: break;
        }
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        raised = true;
    }
    dim3 grid((unsigned)spk::ceil_div(G * d->Cout, SH::CO_T), (unsigned)spk::ceil_div(d->Cin, SH::CI_T), (unsigned)g.splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), 2 * SH::BUF * sizeof(float), stream, a);
    int rc = spk::check_launch("wgrad3x3_s2_kernel");
    if (rc != SPK_OK) return rc;
    return launch_wgrad_reduce(stream, a.slabs, d->dw, g.n_slabs, G * d->Cout, d->Cin, 9, d->scale, d->accumulate ? 1 : 0,
                               d->fold > 1 ? d->fold : 1);
}

template <int MODE>
int run_wgrad_wide_any(const spk_wgrad_desc* d, hipStream_t stream) {
    return d->W >= 16 ? run_wgrad_wide<16, MODE>(d, stream) : run_wgrad_wide<8, MODE>(d, stream);
}

template <int KH, int KW, int S, int MODE>
int run_wgrad(const spk_wgrad_desc* d, hipStream_t stream) {
    using SH = WShape<KH, KW, S>;
    const int G = d->groups > 1 ? d->groups : 1;
    if constexpr (KH == 3 && S == 1 && (MODE == WG_PLAIN || MODE == WG_AFFINE_RELU)) {
        if (wide_takes(d)) return run_wgrad_wide_any<MODE>(d, stream);
    }
    if constexpr (KH == 3 && S == 2 && (MODE == WG_PLAIN || MODE == WG_AFFINE_RELU)) {
        if (s2_takes(d)) return d->W >= 16 ? run_wgrad_s2<16, MODE>(d, stream) : run_wgrad_s2<8, MODE>(d, stream);
    }
    SPK_REQUIRE(!SH::PACK || KW * d->Cin <= 32, "wgrad: the %dx%d kernel packs (kx, ci) into 32 lanes: Cin <= %d", KH, KW, 32 / KW);
    SPK_REQUIRE(G == 1 || d->Cout % SH::CO_T == 0, "wgrad: grouped launches need Cout (per group) to be a multiple of %d", SH::CO_T);
    const WGeom g = wgeom<KH, KW, S>(d->B, d->Cin, G * d->Cout, d->H, d->W, d->splits);
    SPK_REQUIRE(g.lds_bytes <= 160 * 1024, "wgrad: input tile does not fit LDS");
    const size_t slab_floats = (size_t)G * d->Cout * d->Cin * SH::TAPS;
    SPK_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= g.n_slabs * slab_floats * sizeof(float),
                "wgrad: needs a %zu-byte workspace (see spk_conv2d_wgrad_workspace_bytes)", g.n_slabs * slab_floats * sizeof(float));
    WgradArgs a;
    a.g = d->g; a.x = d->x; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.g_scale = d->g_scale; a.slabs = static_cast<float*>(d->workspace);
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W; a.Hs = d->Hin; a.Ws = d->Win;
    a.gin = G > 1 ? d->group_in_stride : d->Cin;
    a.Cx = a.gin * (G - 1) + d->Cin;
    a.Cy = G * d->Cout;
    a.lgTW = spk::ilog2(g.TW); a.lgTH = spk::ilog2(g.TH); a.lgTB = spk::ilog2(g.TB);
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.n_tiles = g.n_tiles;
    // the compile-time tile shape (two workgroups per CU) wherever the geometry picked it
    // (measured, profiles/r01_k_wgrad_phases.txt: +8% on the trunk's BatchNorm-folded layers, whose store phase carries
    // the affine; -4% on the decoder's plain layers, which stay on the one-wave form)
    constexpr bool HAS_FIXED = KH == 3 && S == 1 && MODE == WG_AFFINE_RELU;
    static const bool allow_fixed = [] { const char* e = getenv("SPK_WGRAD_FIXED"); return !e || atoi(e) != 0; }();
    const bool fixed = HAS_FIXED && allow_fixed && g.TW == 16 && g.TH == 4 && g.TB == 1;
    // plain 3x3 stride-1 layers on that tile shape: the form whose staging is interleaved with the MFMAs
    static const bool allow_pipe = [] { const char* e = getenv("SPK_WGRAD_PIPE"); return !e || atoi(e) != 0; }();
    if (KH == 3 && S == 1 && MODE == WG_PLAIN && allow_pipe && g.TW == 16 && g.TH == 4 && g.TB == 1) {
        static bool pipe_raised = false;
        if (!pipe_raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
            pipe_raised = true;
        }
        dim3 pgrid((unsigned)spk::ceil_div(G * d->Cout, SH::CO_T), (unsigned)spk::ceil_div(d->Cin, SH::CI_T), (unsigned)g.splits);
        hipLaunchKernelGGL(wgrad3x3_pipe_kernel, pgrid, dim3(256), 2 * WP_BUF * sizeof(float), stream, a);
        int prc = spk::check_launch("wgrad3x3_pipe_kernel");
        if (prc != SPK_OK) return prc;
        return launch_wgrad_reduce(stream, a.slabs, d->dw, g.n_slabs, G * d->Cout, d->Cin, SH::TAPS, d->scale, d->accumulate ? 1 : 0,
                                   d->fold > 1 ? d->fold : 1);
    }
    auto kern = fixed ? &wgrad_kernel<KH, KW, S, MODE, HAS_FIXED ? 1 : 0> : &wgrad_kernel<KH, KW, S, MODE, 0>;
    if (g.lds_bytes > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
            raised = true;
        }
    }
    dim3 grid((unsigned)spk::ceil_div(G * d->Cout, SH::CO_T),
              (unsigned)(spk::ceil_div(d->Cin, SH::CI_T) * (SH::ROWPASS ? KH : 1)), (unsigned)g.splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), g.lds_bytes, stream, a);
    int rc = spk::check_launch("wgrad_kernel");
    if (rc != SPK_OK) return rc;
    return launch_wgrad_reduce(stream, a.slabs, d->dw, g.n_slabs, G * d->Cout, d->Cin, SH::TAPS, d->scale, d->accumulate ? 1 : 0,
                               d->fold > 1 ? d->fold : 1);
}

// ---- 1x1 weight gradient as a GEMM: dW[co][ci] = sum_{b,pix} g[b,co,pix] * in(x)[b,ci,pix*S] --------------------------------
// The tap-per-tile kernel above keeps ONE accumulator tile per wave for a 1x1 (32 MFMAs per 64-pixel tile against ~80 staged
// elements per thread: staging-bound, a third of the G step's weight-gradient time).  Here a workgroup owns a 128co x 128ci
// block, each of its 2x2 waves a 64x64 quarter (2x2 MFMA tiles sharing their fragments), and walks 32-pixel k-tiles:
// 64 MFMAs per wave against 8 float4 loads per thread, LDS double-buffered (one barrier per k-tile), fragments read one
// k-step ahead.  Pixels are the contraction: rows of G and X are contiguous in memory (16-byte loads, stride 1), and
// LDS rows have pitch 33 so that the 32 lanes of a fragment (32 channels, one pixel) hit 32 banks.
constexpr int G1_KT = 32, G1_PITCH = G1_KT + 1;

// MT x NT = MFMA tiles per wave along co / ci (1 or 2): the block is (64 MT) co x (64 NT) ci.  2 x 2 is the general form;
// the narrow forms serve the trunk's layer1 shapes (64 -> 256, 256 -> 64, 64 -> 64 at 64^2), where a 128-wide block would
// multiply half a block of zeros.
template <int S, int MODE, int MT, int NT>
__global__ __launch_bounds__(256) void wgrad1x1_kernel(const WgradArgs p, int tiles_per_split) {
    constexpr bool AFF = MODE == WG_AFFINE_RELU;
    constexpr int BM = 64 * MT, BN = 64 * NT;             // block rows of G / of X
    constexpr int RG = BM / 32, RX = BN / 32;             // staging rounds: row = tid/8 + 32 i
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const gs = smem;                               // [2][BM][33]
    float* const xs = smem + 2 * BM * G1_PITCH;           // [2][BN][33]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;

    const int co0 = blockIdx.x * BM;                      // in the gradient tensor (all groups)
    const int grp = co0 / p.Cout;
    const int co_end = (grp + 1) * p.Cout;
    const int ci0 = blockIdx.y * BN;                      // within the group
    const int cx0 = grp * p.gin + ci0;                    // first x channel of this block
    const size_t HW = (size_t)p.H * p.W, src_plane = (size_t)p.Hs * p.Ws;
    const int tpi = (int)(HW / G1_KT);                    // k-tiles per image (HW % 32 == 0: checked on the host)
    const int t_begin = blockIdx.z * tiles_per_split, t_end = min(p.n_tiles, t_begin + tiles_per_split);

    // staging roles: row = tid/8 + 32*i, pixels 4*(tid%8) .. +3 of the k-tile
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    float4 gq[RG], xq[RX];
    float a_sc[RX], a_sh[RX];
    bool g_ok[RG], x_ok[RX];
#pragma unroll
    for (int i = 0; i < RG; ++i) g_ok[i] = co0 + srow + 32 * i < co_end;
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int r = srow + 32 * i;
        x_ok[i] = ci0 + r < p.Cin;
        a_sc[i] = 1.f; a_sh[i] = 0.f;
        if (AFF && x_ok[i]) { a_sc[i] = p.in_scale[cx0 + r]; a_sh[i] = p.in_shift[cx0 + r]; }
    }

    // Vector-instruction diet (the f32 MFMA shares the SIMD's vector ALU, tools/mfma_valu_coexec.hip: the first version of this
    // loop spent 150 vector instructions per 64 MFMAs -- a uniform integer division, 64-bit address products per load, a zero
    // select per staged value):
    //   * a load is a SCALAR base (image / k-tile: advanced with scalar adds) + the lane's constant 32-bit offset (row, pixels);
    //   * rows past the group / past Cin are loaded from a valid row and NOT zeroed: they only reach accumulator rows / columns
    //     that the final store skips;
    //   * stride 2 gathers keep their per-lane pixel arithmetic (they are three launches of the trunk).
    unsigned g_lane[RG], x_lane[RX];                       // element offsets of (row, scol) from the k-tile's first pixel
#pragma unroll
    for (int i = 0; i < RG; ++i) g_lane[i] = (unsigned)((g_ok[i] ? co0 + srow + 32 * i : co0) * HW) + (unsigned)scol;
#pragma unroll
    for (int i = 0; i < RX; ++i) x_lane[i] = (unsigned)((x_ok[i] ? cx0 + srow + 32 * i : cx0) * src_plane) + (S == 1 ? (unsigned)scol : 0u);
    // (uniform) position of k-tile t: image b, first pixel q of the tile inside the image
    int tb = t_begin / tpi, tq = (t_begin - tb * tpi) * G1_KT;
#define SPK_G1_LOAD(tb_, tq_)                                                                                \
    {                                                                                                        \
        const float* gb_ = p.g + ((size_t)(tb_) * p.Cy * HW + (size_t)(tq_));                  /* uniform */  \
        _Pragma("unroll") for (int i = 0; i < RG; ++i) gq[i] = *reinterpret_cast<const float4*>(gb_ + g_lane[i]); \
        if (S == 1) {                                                                                        \
            const float* xb_ = p.x + ((size_t)(tb_) * p.Cx * src_plane + (size_t)(tq_));       /* uniform */  \
            _Pragma("unroll") for (int i = 0; i < RX; ++i) xq[i] = *reinterpret_cast<const float4*>(xb_ + x_lane[i]); \
        } else {                                                                                             \
            const float* xb_ = p.x + (size_t)(tb_) * p.Cx * src_plane;                                       \
            const int q0_ = (tq_) + scol;                                                                    \
            const int y_ = q0_ / p.W, x_ = q0_ - y_ * p.W;   /* 4 pixels of one row (W % 4 == 0) */           \
            _Pragma("unroll") for (int i = 0; i < RX; ++i) {                                                 \
                const float* xp = xb_ + x_lane[i] + (size_t)(y_ * S) * p.Ws + x_ * S;                        \
                xq[i] = make_float4(xp[0], xp[S], xp[2 * S], xp[3 * S]);                                     \
            }                                                                                                \
        }                                                                                                    \
    }
#define SPK_G1_STORE(buf_)                                                                                   \
    {                                                                                                        \
        float* gd = gs + (buf_) * BM * G1_PITCH;                                                             \
        float* xd = xs + (buf_) * BN * G1_PITCH;                                                             \
        _Pragma("unroll") for (int i = 0; i < RG; ++i) {                                                     \
            const int o = (srow + 32 * i) * G1_PITCH + scol;                                                 \
            gd[o] = gq[i].x; gd[o + 1] = gq[i].y; gd[o + 2] = gq[i].z; gd[o + 3] = gq[i].w;                  \
        }                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < RX; ++i) {                                                     \
            const int o = (srow + 32 * i) * G1_PITCH + scol;                                                 \
            float4 xv = xq[i];                                                                               \
            if (AFF) {                                                                                       \
                xv.x = fmaxf(xv.x * a_sc[i] + a_sh[i], 0.f); xv.y = fmaxf(xv.y * a_sc[i] + a_sh[i], 0.f);    \
                xv.z = fmaxf(xv.z * a_sc[i] + a_sh[i], 0.f); xv.w = fmaxf(xv.w * a_sc[i] + a_sh[i], 0.f);    \
            }                                                                                                \
            xd[o] = xv.x; xd[o + 1] = xv.y; xd[o + 2] = xv.z; xd[o + 3] = xv.w;                              \
        }                                                                                                    \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    if (t_begin < t_end) {
        SPK_G1_LOAD(tb, tq);
        SPK_G1_STORE(0);
    }
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        tq += G1_KT;                                        // (uniform) the next k-tile: same image, or the next one's first
        if (tq >= (int)HW) { tq = 0; ++tb; }
        if (more) SPK_G1_LOAD(tb, tq);
        const volatile wg_lds_f32* ga = (const volatile wg_lds_f32*)(gs + buf * BM * G1_PITCH + (wm * 32 * MT + l32) * G1_PITCH + half);
        const volatile wg_lds_f32* xb = (const volatile wg_lds_f32*)(xs + buf * BN * G1_PITCH + (wn * 32 * NT + l32) * G1_PITCH + half);
        float fa[2][MT], fb[2][NT];
#define SPK_G1_FRAG(ks_, slot_)                                                                              \
    {                                                                                                        \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) fa[slot_][m] = ga[m * 32 * G1_PITCH + 2 * (ks_)];     \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) fb[slot_][n] = xb[n * 32 * G1_PITCH + 2 * (ks_)];     \
    }
        SPK_G1_FRAG(0, 0);
        wg_static_for<0, G1_KT / 2>([&](auto s_) {
            constexpr int ks = decltype(s_)::value;
            if constexpr (ks + 1 < G1_KT / 2) {
                SPK_G1_FRAG(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks & 1][m], fb[ks & 1][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, MT * NT, 0);
        });
#undef SPK_G1_FRAG
        if (more) SPK_G1_STORE(buf ^ 1);
        __syncthreads();
    }
#undef SPK_G1_LOAD
#undef SPK_G1_STORE

    // partial block -> slab [slab][co][ci] (ci contiguous)
    float* out = p.slabs + (size_t)blockIdx.z * p.Cy * p.Cin;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int ci = ci0 + (wn * NT + n) * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < co_end && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[m][n][r];
            }
        }
}

// ---- the same GEMM, stride 1, with EVERY operand by LDS-DMA ---------------------------------------------------------------------
// The register-staged kernel above pays, per 32-pixel k-tile and thread, 8 float4 loads, 32 ds_write_b32 and (folded BatchNorm) 32
// fma / max -- about 45 vector instructions per 64 MFMAs after its diet, and on gfx950 every one of them is matrix time (the f32
// MFMA and the vector ALU do not overlap); its LDS pitch of 33 also rules out wide fragment reads.  Here:
//   * a k-tile row (one channel, 32 pixels = 128 B = eight 16-byte chunks) goes global -> LDS by `global_load_lds_dwordx4`, 8 rows
//     per instruction (scalar base + one constant lane offset), rows at a pitch of exactly 128 B.  Lane i of the instruction fetches
//     chunk (i & 7) ^ f(row) of its row, f(row) = (row >> 1) & 7: the XOR swizzle costs nothing (a lane may fetch any address);
//   * a fragment read is ONE ds_read_b128 per MFMA tile and FOUR k-steps: lane (channel l32, half) takes pixels
//     8 jq + 4 half .. + 3 -- the pixel sum may run in any order as long as G and X use the same one -- and with the swizzle the
//     16 lanes of every ds_read_b128 lane group ({0-3,12-15,20-27} ...) fall on 16 distinct 16-byte slots: conflict-free;
//   * two ring slots, one bare barrier per k-tile, two workgroups per CU (64 KB each): one's DMA wait, barrier and slab
//     store hide behind the other's MFMAs; the loop's only vector instructions are the folded BatchNorm + ReLU of the X
//     fragments (2 per value and NT tiles; none for plain layers).
// Needs whole blocks (Cout % BM == 0, Cin % BN == 0), HW % 32 == 0 and 16-byte aligned tensors: dma_takes().
typedef __attribute__((address_space(3))) f32x4 wg_lds_f32x4;
typedef __attribute__((address_space(3))) unsigned char wg_lds_u8;

template <int MODE, int MT, int NT>
__global__ __launch_bounds__(256, 2) void wgrad1x1_dma_kernel(const WgradArgs p, int tiles_per_split) {
    constexpr bool AFF = MODE == WG_AFFINE_RELU;
    constexpr int BM = 64 * MT, BN = 64 * NT, KT = G1_KT;
    constexpr int SLOT_B = (BM + BN) * KT * 4;            // bytes per ring slot: rows [0, BM) of G, then [BM, BM + BN) of X
    constexpr int NJ = (BM + BN) / 32;                    // DMA instructions per wave and k-tile (8 rows each)
    static_assert(KT == 32 && 2 * SLOT_B <= 65536, "immediate LDS offsets reach 64 KB");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;

    const int co0 = blockIdx.x * BM;                      // in the gradient tensor (all groups)
    const int grp = co0 / p.Cout;
    const int ci0 = blockIdx.y * BN;                      // within the group
    const int cx0 = grp * p.gin + ci0;                    // first x channel of this block
    const size_t HW = (size_t)p.H * p.W;
    const int tpi = (int)(HW / KT);
    const int t_begin = blockIdx.z * tiles_per_split, t_end = min(p.n_tiles, t_begin + tiles_per_split);

    // DMA role of a lane: row lane / 8 of the instruction's 8 rows; LDS chunk lane % 8 of that row receives global chunk
    // (lane % 8) ^ f(row).  Instruction j = 4 jj + wave covers rows 8 j .. 8 j + 7, so f(row) = (4 (j & 1) + lane / 16) & 7 and
    // j & 1 = wave & 1: ONE constant lane offset per wave.
    // (a BYTE offset added to a byte pointer: base + zext(u32) is the scalar-base form of the load; scaled by 4 it is not)
    unsigned dma_lane = ((unsigned)((lane >> 3) * HW) + (unsigned)((((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) & 7) * 4)) * 4u;
    const float* const g_w = p.g + (size_t)(co0 + 8 * wave) * HW;
    const float* const x_w = p.x + (size_t)(cx0 + 8 * wave) * HW;
    auto dma_tile = [&](int tb_, int tq_, int slot_) {
        const float* gb = g_w + ((size_t)tb_ * p.Cy * HW + (size_t)tq_);            // uniform
        const float* xb = x_w + ((size_t)tb_ * p.Cx * HW + (size_t)tq_);
        char* dst = reinterpret_cast<char*>(smem) + slot_ * SLOT_B + wave * 1024;
        asm volatile("" : "+v"(dma_lane));                // keep (scalar base, 32-bit lane offset): no per-lane 64-bit pointers
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const float* src = jj * 32 < BM ? gb + (size_t)(jj * 32) * HW : xb + (size_t)(jj * 32 - BM) * HW;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(src) + dma_lane, dst + jj * 4096, 16, 0, 0);
        }
    };

    // fragment byte addresses inside a slot, one per 8-pixel group jq (the XOR with the lane's f is not an immediate)
    const unsigned fl = (unsigned)(l32 >> 1) & 7u;
    unsigned a_addr[4], b_addr[4];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
        const unsigned posb = (((unsigned)(2 * jq + half)) ^ fl) * 16u;
        a_addr[jq] = (unsigned)(wm * 32 * MT + l32) * 128u + posb;
        b_addr[jq] = (unsigned)(BM + wn * 32 * NT + l32) * 128u + posb;
        asm volatile("" : "+v"(a_addr[jq]), "+v"(b_addr[jq]));      // eight registers, not a row + chunk add per read
    }
    float sc[NT], sh[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        sc[n] = 1.f; sh[n] = 0.f;
        if (AFF) { sc[n] = p.in_scale[cx0 + (wn * NT + n) * 32 + l32]; sh[n] = p.in_shift[cx0 + (wn * NT + n) * 32 + l32]; }
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    int tb = t_begin / tpi, tq = (t_begin - tb * tpi) * KT;    // (uniform) image and first pixel of the k-tile being fetched
    if (t_begin < t_end) dma_tile(tb, tq, 0);

    // one k-tile out of ring slot SLOT (compile-time: every LDS offset is an immediate)
    auto run_tile = [&](auto slot_c, bool more) {
        constexpr int SLOT = decltype(slot_c)::value;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0): this wave's rows of the k-tile have landed
        __builtin_amdgcn_s_barrier();                     // ... and everyone's; everyone is done reading the other slot
        __builtin_amdgcn_sched_barrier(0);
        tq += KT;
        if (tq >= (int)HW) { tq = 0; ++tb; }
        if (more) dma_tile(tb, tq, SLOT ^ 1);
        f32x4 fa[2][MT], fb[2][NT];
#define SPK_GD_FRAG(jq_, f_)                                                                                                   \
    {                                                                                                                          \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                         \
            fa[f_][m] = *(const wg_lds_f32x4*)((wg_lds_u8*)smem + (a_addr[jq_] + (unsigned)(SLOT * SLOT_B + m * 4096)));      \
        _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                                         \
            fb[f_][n] = *(const wg_lds_f32x4*)((wg_lds_u8*)smem + (b_addr[jq_] + (unsigned)(SLOT * SLOT_B + n * 4096)));      \
    }
        SPK_GD_FRAG(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wg_static_for<0, 4>([&](auto q_) {
            constexpr int jq = decltype(q_)::value;
            if constexpr (jq + 1 < 4) {
                SPK_GD_FRAG(jq + 1, (jq + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);          // the next group's reads first: 16 MFMAs of flight
                __builtin_amdgcn_s_waitcnt(0xC07F | ((MT + NT) << 8));      // lgkmcnt(MT + NT): this group's have landed, no more
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (AFF) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fb[jq & 1][n][i] = fmaxf(__builtin_fmaf(fb[jq & 1][n][i], sc[n], sh[n]), 0.f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[jq & 1][m][i], fb[jq & 1][n][i], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
#undef SPK_GD_FRAG
    };
    for (int t = t_begin; t < t_end; t += 2) {
        run_tile(std::integral_constant<int, 0>{}, t + 1 < t_end);
        if (t + 1 < t_end) run_tile(std::integral_constant<int, 1>{}, t + 2 < t_end);
    }

    // partial block -> slab [slab][co][ci] (ci contiguous)
    float* out = p.slabs + (size_t)blockIdx.z * p.Cy * p.Cin;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int ci = ci0 + (wn * NT + n) * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(size_t)co * p.Cin + ci] = acc[m][n][r];
            }
        }
}

// whether the LDS-DMA form takes a stride-1 problem of block shape (bm, bn)
inline bool dma_takes(const spk_wgrad_desc* d, int bm, int bn) {
    static const bool allow = [] { const char* e = getenv("SPK_WGRAD1X1_DMA"); return !e || atoi(e) != 0; }();
    const auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const long long HW = (long long)d->H * d->W;
    return allow && d->stride == 1 && d->Cout % bm == 0 && d->Cin % bn == 0 && HW % G1_KT == 0 && HW * 8 * 4 < (1ll << 31) &&
           aligned(d->g) && aligned(d->x) && d->Hin == d->H && d->Win == d->W;
}

// block shape of the GEMM form for a problem: tiles per wave along co / ci
inline void g1_shape(int groups, int Cout, int Cin, int* mt, int* nt) {
    const int G = groups > 1 ? groups : 1;
    *mt = (Cout % 128 == 0 || (G == 1 && Cout > 64)) ? 2 : 1;
    *nt = Cin > 64 ? 2 : 1;
}

// geometry of the GEMM form: pixel k-tiles and how many slabs (pixel splits) fill the chip
struct G1Geom { int n_tiles, splits, tiles_per_split; bool ok; };
inline G1Geom g1geom(int bm, int bn, int B, int Cin, int Cout_all, int H, int W, int want_splits) {
    G1Geom g;
    const long long HW = (long long)H * W;
    g.ok = HW % G1_KT == 0 && W % 4 == 0;
    g.n_tiles = (int)(B * HW / G1_KT);
    if (!g.ok || g.n_tiles == 0) { g.ok = false; g.splits = 1; g.tiles_per_split = 1; return g; }
    const int blocks = spk::ceil_div(Cout_all, bm) * spk::ceil_div(Cin, bn);
    int sp = want_splits > 0 ? want_splits : std::max(1, 512 / blocks);
    sp = std::max(1, std::min(sp, std::max(1, g.n_tiles / 4)));       // at least 4 k-tiles per workgroup
    g.tiles_per_split = spk::ceil_div(g.n_tiles, sp);
    g.splits = spk::ceil_div(g.n_tiles, g.tiles_per_split);
    return g;
}

template <int S, int MODE, int MT, int NT>
int run_wgrad1x1_shape(const spk_wgrad_desc* d, hipStream_t stream) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    const int G = d->groups > 1 ? d->groups : 1;
    const G1Geom g = g1geom(BM, BN, d->B, d->Cin, G * d->Cout, d->H, d->W, d->splits);
    const size_t slab_floats = (size_t)G * d->Cout * d->Cin;
    SPK_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= g.splits * slab_floats * sizeof(float),
                "wgrad: needs a %zu-byte workspace (see spk_conv2d_wgrad_workspace_bytes)", g.splits * slab_floats * sizeof(float));
    WgradArgs a;
    a.g = d->g; a.x = d->x; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.g_scale = d->g_scale; a.slabs = static_cast<float*>(d->workspace);
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W; a.Hs = d->Hin; a.Ws = d->Win;
    a.gin = G > 1 ? d->group_in_stride : d->Cin;
    a.Cx = a.gin * (G - 1) + d->Cin;
    a.Cy = G * d->Cout;
    a.lgTW = a.lgTH = a.lgTB = 0; a.tiles_x = a.tiles_y = 0; a.n_tiles = g.n_tiles;
    auto kern = &wgrad1x1_kernel<S, MODE, MT, NT>;
    size_t lds = 2 * (size_t)(BM + BN) * G1_PITCH * sizeof(float);
    bool dma = false;
    if constexpr (S == 1) {
        if (dma_takes(d, BM, BN)) {
            kern = &wgrad1x1_dma_kernel<MODE, MT, NT>;
            lds = 2 * (size_t)(BM + BN) * G1_KT * sizeof(float);
            dma = true;
        }
    }
    static bool raised[2] = {false, false};
    if (!raised[dma]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        raised[dma] = true;
    }
    SPK_REQUIRE((long long)a.Cy * d->H * d->W < (1ll << 31) && (long long)a.Cx * d->Hin * d->Win < (1ll << 31),
                "wgrad 1x1 GEMM form: an image's planes are addressed with 32-bit offsets");
    // a co block must not straddle two groups
    SPK_REQUIRE(G == 1 || d->Cout % BM == 0, "wgrad 1x1 GEMM form: grouped launches need Cout %% %d == 0 (use the tap kernel)", BM);
    dim3 grid((unsigned)spk::ceil_div(G * d->Cout, BM), (unsigned)spk::ceil_div(d->Cin, BN), (unsigned)g.splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a, g.tiles_per_split);
    int rc = spk::check_launch("wgrad1x1_kernel");
    if (rc != SPK_OK) return rc;
    return launch_wgrad_reduce(stream, a.slabs, d->dw, g.splits, G * d->Cout, d->Cin, 1, d->scale, d->accumulate ? 1 : 0,
                               d->fold > 1 ? d->fold : 1);
}

template <int S, int MODE>
int run_wgrad1x1(const spk_wgrad_desc* d, hipStream_t stream) {
    int mt, nt;
    g1_shape(d->groups, d->Cout, d->Cin, &mt, &nt);
    if (mt == 2) return nt == 2 ? run_wgrad1x1_shape<S, MODE, 2, 2>(d, stream) : run_wgrad1x1_shape<S, MODE, 2, 1>(d, stream);
    return nt == 2 ? run_wgrad1x1_shape<S, MODE, 1, 2>(d, stream) : run_wgrad1x1_shape<S, MODE, 1, 1>(d, stream);
}

// whether the GEMM form takes this 1x1 problem (else the tap kernel does)
inline bool g1_takes(int groups, int Cout, int H, int W) {
    const int G = groups > 1 ? groups : 1;
    return ((long long)H * W) % G1_KT == 0 && W % 4 == 0 && (G == 1 || Cout % 64 == 0);
}

// ---- the stem's weight gradient: 7x7 stride 2, Cin = 3, Cout = 64 per group --------------------------------------------------
// dW[co][k] = sum_{b, oy, ox} g[b, co, oy, ox] * x[b, ci, 2 oy + ky - 3, 2 ox + kx - 3], k = (ci, ky, kx): a GEMM of 64 x 147 outputs over
// B H W pixels.  The generic tap kernel runs it with 49 taps of a padded 4-channel chunk (46 TFLOP/s, 0.32 ms of a G step).  Here,
// as in conv7x7_stem.hip: a 4 x 32 output tile's input patch (13 rows x 69 columns x 3 channels) in LDS with its columns split by
// parity, the tile's gradients [64 co][128 px] beside it; MFMA 32x32x2 with A = gradients (lane = channel, half = one of two adjacent
// pixels) and B = patch values (lane = k: a per-lane constant offset (ci, ky, kx) + the pixel as an immediate); every wave owns one
// row of the tile and ALL 2 x 5 accumulator tiles (64 co x 160 k, 160 registers), workgroups are persistent over tiles, the four
// waves' sums meet in LDS at the end and leave as one slab per workgroup ([co][147], the layout of dW itself).
constexpr int SW_TH = 4, SW_TW = 32, SW_PR = 2 * SW_TH + 5, SW_PC = 2 * SW_TW + 5, SW_PH = SW_TW + 3, SW_PROW = 2 * SW_PH;
constexpr int SW_K = 147, SW_NT = 5;
constexpr int SW_GP = SW_TH * SW_TW + 1;                       // pitch of a channel's gradients (odd: 32 channels, 32 banks)
constexpr int SW_G_FL = 64 * SW_GP, SW_P_FL = 3 * SW_PR * SW_PROW;
constexpr int SW_LDS_FL = SW_G_FL + SW_P_FL;                   // 8256 + 2730 floats = 43.9 KB: two workgroups per CU
constexpr int SW_TASKS = 3 * SW_PR, SW_TPW = (SW_TASKS + 3) / 4;
static_assert(SW_LDS_FL * 4 >= 4 * 2 * 16 * 64 * 4, "the final cross-wave sum of one n-tile fits the same LDS");

__global__ __launch_bounds__(256, 2) void wgrad_stem_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int grp = blockIdx.y;
    const float* gb = p.g + (size_t)grp * 64 * p.H * p.W;             // + b Cy H W
    const float* xb = p.x + (size_t)grp * p.gin * p.Hs * p.Ws;       // + b Cx Hs Ws

    // staging roles: gradients -- float4 f = tid + 256 i of the tile's [64 co][4 rows][8 float4]; patch -- as conv7x7_stem.hip
    float4 gq[8];
    float pa[SW_TPW], pb[SW_TPW];
    const int g_co = tid >> 5, g_row = (tid >> 3) & 3, g_c4 = tid & 7;                     // + 8 i channels
    const unsigned g_dst = (unsigned)(g_co * SW_GP + g_row * SW_TW + 4 * g_c4) * 4u;      // + 8 i SW_GP floats
    const unsigned p_dst_a = (unsigned)(SW_G_FL + wave * SW_PROW + (lane & 1) * SW_PH + (lane >> 1)) * 4u;
    const unsigned p_dst_b = p_dst_a + 32 * 4;
    auto load_tile = [&](int t) {
        const bool live = t < p.n_tiles;
        const int tt = live ? t : 0;
        const int tx = tt % p.tiles_x, q = tt / p.tiles_x;
        const int ty = q % p.tiles_y, b = q / p.tiles_y;
        const int oy0 = ty * SW_TH, ox0 = tx * SW_TW;
        const float* gt = gb + (size_t)b * p.Cy * p.H * p.W;
        const bool g_ok = live && oy0 + g_row < p.H && ox0 + 4 * g_c4 < p.W;            // (W % 4 == 0: host-checked)
        const unsigned g_off = (unsigned)((min(oy0 + g_row, p.H - 1)) * p.W + min(ox0 + 4 * g_c4, p.W - 4)) * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(gt + (size_t)(g_co + 8 * i) * p.H * p.W) + g_off);
            gq[i] = g_ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);      // zero gradients outside: the pixel sum ignores them
        }
        const float* xt = xb + (size_t)b * p.Cx * p.Hs * p.Ws;
        const int ix_a = 2 * ox0 - 3 + lane, ix_b = ix_a + 64;
        const bool ok_a = (unsigned)ix_a < (unsigned)p.Ws, ok_b = lane < SW_PC - 64 && (unsigned)ix_b < (unsigned)p.Ws;
        const unsigned off_a = (unsigned)min(max(ix_a, 0), p.Ws - 1) * 4u, off_b = (unsigned)min(max(ix_b, 0), p.Ws - 1) * 4u;
#pragma unroll
        for (int j = 0; j < SW_TPW; ++j) {
            const int qq = min(wave + 4 * j, SW_TASKS - 1);                  // (uniform)
            const int ci = qq / SW_PR, r = qq - ci * SW_PR;
            const int iy = 2 * oy0 - 3 + r;
            const bool row_ok = (unsigned)iy < (unsigned)p.Hs;
            const char* row = reinterpret_cast<const char*>(xt + ((size_t)ci * p.Hs + min(max(iy, 0), p.Hs - 1)) * p.Ws);
            const float a = *reinterpret_cast<const float*>(row + off_a);
            const float c = *reinterpret_cast<const float*>(row + off_b);
            pa[j] = (row_ok && ok_a) ? a : 0.f;
            pb[j] = (row_ok && ok_b) ? c : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            volatile wg_lds_f32* d = (volatile wg_lds_f32*)((wg_lds_u8*)smem + (g_dst + (unsigned)(8 * i * SW_GP * 4)));
            d[0] = gq[i].x; d[1] = gq[i].y; d[2] = gq[i].z; d[3] = gq[i].w;
        }
#pragma unroll
        for (int j = 0; j < SW_TPW; ++j) {
            if (wave + 4 * j < SW_TASKS) {                                   // (uniform)
                *(volatile wg_lds_f32*)((wg_lds_u8*)smem + (p_dst_a + (unsigned)(j * 4 * SW_PROW * 4))) = pa[j];
                if (lane < SW_PC - 64) *(volatile wg_lds_f32*)((wg_lds_u8*)smem + (p_dst_b + (unsigned)(j * 4 * SW_PROW * 4))) = pb[j];
            }
        }
    };

    // fragment addresses (bytes): A -- channel m 32 + l32, pixel row = wave, pixels 2 s + half; B -- k = 32 j + l32 (clamped: columns
    // >= 147 are never written out), the same pixels
    unsigned a_addr = (unsigned)(l32 * SW_GP + wave * SW_TW + half) * 4u;
    unsigned b_addr[SW_NT];
#pragma unroll
    for (int j = 0; j < SW_NT; ++j) {
        const int k = min(32 * j + l32, SW_K - 1);
        const int ci = k / 49, ky = (k % 49) / 7, kx = k % 7;
        b_addr[j] = (unsigned)(SW_G_FL + ((ci * SW_PR + 2 * wave + ky) * 2 + (kx & 1)) * SW_PH + (kx >> 1) + half) * 4u;
        asm volatile("" : "+v"(b_addr[j]));
    }
    asm volatile("" : "+v"(a_addr));

    f32x16 acc[2][SW_NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < SW_NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    int tile = blockIdx.x;
    load_tile(tile);
    store_tile();
    __syncthreads();
    for (; tile < p.n_tiles; tile += gridDim.x) {
        load_tile(tile + (int)gridDim.x);                   // the next tile's loads fly behind this tile's MFMAs
        float fa[2][2], fb[2][SW_NT];
#define SPK_SW_FRAG(s_, f_)                                                                                                   \
    {                                                                                                                         \
        fa[f_][0] = *(const volatile wg_lds_f32*)((wg_lds_u8*)smem + (a_addr + (unsigned)(2 * (s_) * 4)));                    \
        fa[f_][1] = *(const volatile wg_lds_f32*)((wg_lds_u8*)smem + (a_addr + (unsigned)((32 * SW_GP + 2 * (s_)) * 4)));     \
        _Pragma("unroll") for (int j = 0; j < SW_NT; ++j)                                                                     \
            fb[f_][j] = *(const volatile wg_lds_f32*)((wg_lds_u8*)smem + (b_addr[j] + (unsigned)(2 * (s_) * 4)));             \
    }
        SPK_SW_FRAG(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wg_static_for<0, SW_TW / 2>([&](auto s_) {
            constexpr int st = decltype(s_)::value;
            if constexpr (st + 1 < SW_TW / 2) {
                SPK_SW_FRAG(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < SW_NT; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1][m], fb[st & 1][j], acc[m][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
#undef SPK_SW_FRAG
        __syncthreads();                                    // every wave is done reading the tile
        store_tile();
        __syncthreads();
    }

    // ---- the four waves' sums, one n-tile at a time through LDS, -> slab [slab][co_all][147] ----
    float* out = p.slabs + ((size_t)blockIdx.x * p.Cy + (size_t)grp * 64) * SW_K;
#pragma unroll
    for (int j = 0; j < SW_NT; ++j) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) smem[((wave * 2 + m) * 16 + r) * 64 + lane] = acc[m][j][r];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = tid + 256 * e;                  // (m, r, lane) of 2 x 16 x 64
            const int ln = idx & 63, r = (idx >> 6) & 15, m = idx >> 10;
            const float v = (smem[idx] + smem[2048 + idx]) + (smem[4096 + idx] + smem[6144 + idx]);
            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), k = 32 * j + (ln & 31);
            if (k < SW_K) out[co * SW_K + k] = v;
        }
        __syncthreads();
    }
}

inline bool stem_wgrad_takes(const spk_wgrad_desc* d) {
    static const bool allow = [] { const char* e = getenv("SPK_WGRAD_STEM"); return !e || atoi(e) != 0; }();
    const int G = d->groups > 1 ? d->groups : 1;
    const auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return allow && d->kh == 7 && d->kw == 7 && d->stride == 2 && d->Cin == 3 && d->Cout == 64 && d->W % 4 == 0 && aligned(d->g) &&
           (d->flags & ~0u) == 0 && (long long)G * 64 * d->H * d->W < (1ll << 29) && (long long)3 * G * d->Hin * d->Win < (1ll << 29) &&
           (G == 1 || d->group_in_stride == 0 || d->group_in_stride >= 3);
}
// slabs (= persistent workgroups per group) of the stem form
inline int stem_wgrad_slabs(int B, int H, int W, int G) {
    const long long tiles = (long long)B * spk::ceil_div(H, SW_TH) * spk::ceil_div(W, SW_TW);
    return (int)std::max(1ll, std::min(tiles, (long long)std::max(1, 512 / std::max(G, 1))));
}

int run_wgrad_stem(const spk_wgrad_desc* d, hipStream_t stream) {
    const int G = d->groups > 1 ? d->groups : 1;
    const int n_slabs = d->splits > 0 ? std::min(d->splits, 512) : stem_wgrad_slabs(d->B, d->H, d->W, G);
    const size_t slab_floats = (size_t)G * 64 * SW_K;
    SPK_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= n_slabs * slab_floats * sizeof(float),
                "wgrad: needs a %zu-byte workspace (see spk_conv2d_wgrad_workspace_bytes)", n_slabs * slab_floats * sizeof(float));
    WgradArgs a;
    a.g = d->g; a.x = d->x; a.in_scale = nullptr; a.in_shift = nullptr; a.g_scale = nullptr; a.slabs = static_cast<float*>(d->workspace);
    a.B = d->B; a.Cin = 3; a.Cout = 64; a.H = d->H; a.W = d->W; a.Hs = d->Hin; a.Ws = d->Win;
    a.gin = G > 1 ? d->group_in_stride : 3;
    a.Cx = a.gin * (G - 1) + 3;
    a.Cy = G * 64;
    a.lgTW = a.lgTH = a.lgTB = 0;
    a.tiles_x = spk::ceil_div(d->W, SW_TW); a.tiles_y = spk::ceil_div(d->H, SW_TH); a.n_tiles = a.tiles_x * a.tiles_y * d->B;
    const int wgs = std::min(n_slabs, a.n_tiles);
    hipLaunchKernelGGL(wgrad_stem_kernel, dim3((unsigned)wgs, (unsigned)G), dim3(256), SW_LDS_FL * sizeof(float), stream, a);
    int rc = spk::check_launch("wgrad_stem_kernel");
    if (rc != SPK_OK) return rc;
    return launch_wgrad_reduce(stream, a.slabs, d->dw, wgs, G * 64, SW_K, 1, d->scale, d->accumulate ? 1 : 0, d->fold > 1 ? d->fold : 1);
}

template <int KH, int KW, int S>
int by_mode(int mode, const spk_wgrad_desc* d, hipStream_t s) {
    if (mode == WG_AFFINE_RELU) return run_wgrad<KH, KW, S, WG_AFFINE_RELU>(d, s);
    return run_wgrad<KH, KW, S, WG_PLAIN>(d, s);
}

bool wg_supported(int kh, int kw, int stride) {
    if (kh == 4 && kw == 4) return stride == 2;
    return kh == kw && (kh == 1 || kh == 3 || kh == 7) && (stride == 1 || stride == 2) && !(kh == 7 && stride == 1);
}

}  // namespace

extern "C" {

int64_t spk_conv2d_wgrad_workspace_bytes(int kh, int kw, int stride, int splits, int B, int Cin, int Cout, int H, int W) {
    if (!wg_supported(kh, kw, stride) || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return -1;
    WGeom g;
    if (kh == 1) {
        // Cout here is the channel count of the gradient tensor (groups * Cout for a grouped launch); the GEMM form
        // needs no more than the tap kernel whenever it applies, so the larger of the two is always enough
        g = stride == 1 ? wgeom<1, 1, 1>(B, Cin, Cout, H, W, splits) : wgeom<1, 1, 2>(B, Cin, Cout, H, W, splits);
        int g1_splits = 0;       // the group count is not known here: the largest over the GEMM form's block shapes
        bool g1_ok = false;
        for (int bm = 64; bm <= 128; bm += 64)
            for (int bn = 64; bn <= 128; bn += 64) {
                const G1Geom g1 = g1geom(bm, bn, B, Cin, Cout, H, W, splits);
                g1_ok = g1.ok;
                g1_splits = std::max(g1_splits, g1.splits);
            }
        const int64_t tap = (int64_t)g.n_slabs * Cout * Cin * (int64_t)sizeof(float);
        const int64_t gemm = g1_ok ? (int64_t)g1_splits * Cout * Cin * (int64_t)sizeof(float) : 0;
        return std::max(tap, gemm);
    }
    else if (kh == 3) g = stride == 1 ? wgeom<3, 3, 1>(B, Cin, Cout, H, W, splits) : wgeom<3, 3, 2>(B, Cin, Cout, H, W, splits);
    else if (kh == 4) g = wgeom<4, 4, 2>(B, Cin, Cout, H, W, splits);
    else g = wgeom<7, 7, 2>(B, Cin, Cout, H, W, splits);
    int n_slabs = g.n_slabs;
    if (kh == 3 && stride == 1) {     // the 16-byte-load form may take the problem
        n_slabs = std::max(n_slabs, wide_geom(64, 64, B, Cin, Cout, H, W, splits, W >= 16 ? 16 : 8, wide_sb_on()).n_slabs);
    }
    if (kh == 3 && stride == 2 && W >= 8) n_slabs = std::max(n_slabs, s2_geom(W >= 16 ? 16 : 8, B, Cin, Cout, H, W, splits).n_slabs);
    if (kh == 7 && Cin == 3 && Cout % 64 == 0)       // the stem form (Cout here = groups * 64): one slab per persistent workgroup
        n_slabs = std::max(n_slabs, splits > 0 ? std::min(splits, 512) : stem_wgrad_slabs(B, H, W, Cout / 64));
    return (int64_t)n_slabs * Cout * Cin * kh * kw * (int64_t)sizeof(float);
}

int spk_conv2d_wgrad_up_supported(int B, int Cin, int Cout, int H, int W) {
    // Measured (tools/bench_wgrad.py --upsample, B = 8): the folded form beats "upsample, then the plain kernel" at 256^2
    // (744 vs 775 us, and 268 MB less written and read back), is level with it at 128^2 (668 vs 668, 134 MB less) and 2 %
    // behind at 32^2..64^2, whose x2 images are small (8..67 MB): by default it takes planes at least 128 wide
    // (SPK_WGRAD_UP_MIN_W overrides).
    static const int min_w = [] { const char* e = getenv("SPK_WGRAD_UP_MIN_W"); return e ? atoi(e) : 128; }();
    if (W < min_w) return 0;
    return (B > 0 && Cin > 0 && Cout > 0 && W >= 16 && W % 8 == 0 && H >= 4 && H % 2 == 0 && (long long)Cout * H * W < (1ll << 31) &&
            (long long)Cin * H * W / 4 < (1ll << 31)) ? 1 : 0;
}

int spk_conv2d_wgrad_mod_supported(int B, int Cin, int Cout, int H, int W, int upsample) {
    // shapes (OUTPUT size H x W) whose modulated weight gradient runs fused (SPK_CONV_IN_BATCH_SCALE); 16-byte aligned tensors
    // are a further, per-call condition.  Everything else: rescale the operands first (tiny layers only: 4^2).
    if (B <= 0 || Cin <= 0 || Cout <= 0 || (long long)Cout * H * W >= (1ll << 31) || (long long)Cin * H * W >= (1ll << 31)) return 0;
    if (upsample) return (W >= 16 && W % 8 == 0 && H >= 4 && H % 2 == 0) ? 1 : 0;
    return (W >= 8 && W % 4 == 0 && H >= 4) ? 1 : 0;
}

int spk_wgrad_reduce_slabs(const float* slabs, float* dw, int n_slabs, int Cout, int Cin, int taps, float scale, int accumulate, int fold,
                           void* stream) {
    SPK_REQUIRE(slabs && dw && n_slabs > 0 && Cout > 0 && Cin > 0 && taps > 0, "wgrad reduce: bad arguments");
    return launch_wgrad_reduce((hipStream_t)stream, slabs, dw, n_slabs, Cout, Cin, taps, scale, accumulate ? 1 : 0, fold > 1 ? fold : 1);
}

int spk_conv2d_wgrad(const spk_wgrad_desc* d, void* stream) {
    SPK_REQUIRE(d && d->g && d->x && d->dw, "wgrad: null pointer");
    if (d->flags & SPK_CONV_WINOGRAD) return spk_conv2d_wgrad_wino(d, stream);
    SPK_REQUIRE(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->H > 0 && d->W > 0 && d->Hin > 0 && d->Win > 0, "wgrad: bad shape");
    SPK_REQUIRE(wg_supported(d->kh, d->kw, d->stride), "wgrad: unsupported kernel %dx%d stride %d", d->kh, d->kw, d->stride);
    SPK_REQUIRE(d->fold <= 1 || (d->groups > 1 && d->groups % d->fold == 0), "wgrad: fold %d must divide groups %d", d->fold, d->groups);
    const bool ups = d->flags & SPK_CONV_UPSAMPLE2X, aff = d->flags & SPK_CONV_IN_AFFINE_RELU, bsc = d->flags & SPK_CONV_IN_BATCH_SCALE;
    SPK_REQUIRE(!(ups && aff) && !(bsc && aff), "wgrad: IN_AFFINE_RELU excludes UPSAMPLE2X and IN_BATCH_SCALE");
    SPK_REQUIRE(!ups || (d->kh == 3 && d->stride == 1), "wgrad: UPSAMPLE2X needs a 3x3 stride-1 kernel");
    SPK_REQUIRE(!aff || (d->in_scale && d->in_shift), "wgrad: IN_AFFINE_RELU without in_scale/in_shift");
    const int pad = (d->kh - 1) / 2;
    if (ups) SPK_REQUIRE(d->H == 2 * d->Hin && d->W == 2 * d->Win, "wgrad: upsampled size mismatch");
    else SPK_REQUIRE(d->H == (d->Hin + 2 * pad - d->kh) / d->stride + 1 && d->W == (d->Win + 2 * pad - d->kw) / d->stride + 1,
                     "wgrad: output size %dx%d does not match input %dx%d", d->H, d->W, d->Hin, d->Win);
    hipStream_t s = (hipStream_t)stream;
    const int mode = aff ? WG_AFFINE_RELU : WG_PLAIN;
    // SPK_CONV_UPSAMPLE2X: x is the low-resolution tensor; the x2 plane is interpolated LDS -> LDS from a source patch
    // (wgrad3x3_up_kernel).  Shapes it does not take (W % 8 != 0, planes under 16 x 4): callers upsample with
    // spk_upsample2x_bilinear_fwd first -- spk_conv2d_wgrad_up_supported tells which.
    if (bsc) {
        // the modulated convolution's weight gradient (StyleGAN2 variant): x * s[b,ci] and g * d'[b,co] formed while staging
        SPK_REQUIRE(d->in_scale && d->g_scale, "wgrad: IN_BATCH_SCALE needs in_scale = s[B,Cin] and g_scale = d'[B,Cout]");
        SPK_REQUIRE(d->kh == 3 && d->stride == 1 && d->groups <= 1, "wgrad: IN_BATCH_SCALE is built for ungrouped 3x3 stride-1 convs");
        SPK_REQUIRE(!ups || (d->flags & SPK_CONV_UP_FIR1331), "wgrad: IN_BATCH_SCALE with UPSAMPLE2X is the upfirdn2d [1,3,3,1] form (SPK_CONV_UP_FIR1331)");
        if (ups) {
            if (up_takes(d)) return run_wgrad_up<WG_BATCH_SCALE>(d, s);
        } else if (wide_takes(d)) {
            return run_wgrad_wide_any<WG_BATCH_SCALE>(d, s);
        }
        return spk::fail(SPK_EUNSUPPORTED, "wgrad: IN_BATCH_SCALE does not take this shape (spk_conv2d_wgrad_mod_supported): pass rescaled tensors");
    }
    SPK_REQUIRE(!(d->flags & SPK_CONV_UP_FIR1331), "wgrad: UP_FIR1331 goes with IN_BATCH_SCALE");
    if (ups) {
        if (up_takes(d)) return run_wgrad_up<WG_PLAIN>(d, s);
        return spk::fail(SPK_EUNSUPPORTED, "wgrad: SPK_CONV_UPSAMPLE2X does not take this shape: pass the upsampled input (spk_upsample2x_bilinear_fwd)");
    }
    if (d->kh == 1 && g1_takes(d->groups, d->Cout, d->H, d->W)) {
        if (d->stride == 1) return aff ? run_wgrad1x1<1, WG_AFFINE_RELU>(d, s) : run_wgrad1x1<1, WG_PLAIN>(d, s);
        return aff ? run_wgrad1x1<2, WG_AFFINE_RELU>(d, s) : run_wgrad1x1<2, WG_PLAIN>(d, s);
    }
    if (d->kh == 1) return d->stride == 1 ? by_mode<1, 1, 1>(mode, d, s) : by_mode<1, 1, 2>(mode, d, s);
    if (d->kh == 3) return d->stride == 1 ? by_mode<3, 3, 1>(mode, d, s) : by_mode<3, 3, 2>(mode, d, s);
    if (d->kh == 4) {
        SPK_REQUIRE(!aff, "wgrad: the 4x4 stride-2 form takes a plain input");
        return run_wgrad<4, 4, 2, WG_PLAIN>(d, s);
    }
    if (!aff && stem_wgrad_takes(d)) return run_wgrad_stem(d, s);
    return by_mode<7, 7, 2>(mode, d, s);
}

}  // extern "C"
