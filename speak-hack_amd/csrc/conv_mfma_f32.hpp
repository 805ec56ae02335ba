// 2-D convolution (KHxKW in {1x1, 3x3, 7x7}, stride 1 or 2, zero pad (K-1)/2; 2x2 = window rows y, y+1, see the epilogue) as an LDS-staged implicit
// GEMM on the gfx950 f32 MFMA pipe.  One kernel template serves the StyleGAN decoder (3x3 s1 with the
// bilinear x2 upsampling folded into staging and the bias/noise/LeakyReLU/style epilogue) and the
// ResNet-50 trunk (1x1 / 3x3 / 7x7, stride 1/2, the producer's BatchNorm+ReLU folded into staging as a
// per-channel affine, BatchNorm batch statistics accumulated in the epilogue).
//
// GEMM view (per image group):  D[co][pix] = sum_k A[co][k] * Bm[k][pix],  k = (tap, ci)
//   A  = weights, pre-packed [co_tile][ci_chunk][tap][ci][co]  (co contiguous -> conflict-free
//        ds_read_b32 of the A fragment: lane l reads A[i = l&31][k = l>>5])
//   Bm = input window; the tile's input planes (with halo) sit in LDS as [ci][tb][PH][PW];
//        lane l reads Bm[k = l>>5][j = l&31] = 32 consecutive output pixels (stride S apart in LDS)
//   D  : v_mfma_f32_32x32x2_f32, col = lane&31 = pixel (x-contiguous -> 128-B coalesced NCHW
//        stores), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = output channel.
// One workgroup = WM x WN waves; each wave owns MT x NT accumulator tiles of 32co x 32pix.
//
// Pipeline: a three-slot LDS ring.  During chunk i a wave -- between its KH*KW*CI_T/2 MFMA k-steps out of slot i%3 --
// stores the prefetch registers (chunk i+1) to slot (i+1)%3, passes the chunk's single barrier, issues the global
// loads of chunk i+2 (weights: 16-B loads of one contiguous packed block; input: per-lane gathers whose offsets,
// validity and interpolation weights were tabulated once at kernel start) and already reads chunk i+1's first
// fragments.  Nothing serial happens at a chunk boundary, so a wave's MFMA stream is continuous from the first chunk
// to the last; the f32 MFMA is 64 cycles/SIMD per instruction (an exact fmaf chain) and every staging piece is 1-3
// branch-free instructions pinned behind one of them (see the main loop for the protocol and its hazards).
//
// Split-K: gridDim.z slices the ci-chunk range; slices write raw partial sums to a workspace and
// splitk_epilogue_kernel reduces them in a fixed order and applies the epilogue.
#pragma once
#include "spk_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace spkconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32_t;        // LDS-typed accesses: volatile ones stay ds_read / ds_write

// MODE_PLAIN_STATS: MODE_PLAIN of a 3x3 stride-1 conv with the BatchNorm-statistics epilogue compiled in (see HAS_STATS in
// the kernel); every other kernel shape carries that epilogue in all its modes.
enum Mode { MODE_PLAIN = 0, MODE_UPSAMPLE = 1, MODE_AFFINE_RELU = 2, MODE_BATCH_SCALE = 3, MODE_UPSAMPLE_BATCH_SCALE = 4, MODE_PLAIN_STATS = 5 };

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct ConvArgs {
    const float* x;
    const float* wp;
    const float* bias;
    const float* noise_w;
    const float* noise;
    const float* style;
    const float* in_scale;   // MODE_AFFINE_RELU: per input channel; MODE_BATCH_SCALE: [B,Cin] modulation
    const float* out_scale_bc;  // optional [B,Cout] factor applied right after the contraction (demodulation)
    const float* in_shift;
    double* stats;           // [stats_slots][2*Cy] sum / sum of squares of y (SPK_EPI_STATS)
    int stats_slots;         // copies: workgroup (pixel tile) i adds into copy i % stats_slots; >= gridDim.x: plain stores
    float* y;                // output, or the split-K workspace [ksplit][B][Cout][H][W]
    float* y_pre;            // optional: value before the style stage (kept for backward)
    int B, Cin, Cout, H, W;  // output spatial size
    int Hs, Ws;              // source tensor spatial size
    int lgTW, lgTH, lgTB;    // log2 of the pixel-tile geometry
    int tiles_x, tiles_y;
    int n_chunks;            // ceil(Cin / CI_T)
    int chunks_per_split;
    int style_stride;
    unsigned flags;
    float slope, out_scale, act_gain;
    const float* out_scale_dev;   // optional device scalar multiplied into out_scale (spk_conv2d_desc.out_scale_dev)
    unsigned magic_plane, magic_pw;   // ceil(2^32 / PLANE), ceil(2^32 / PW): exact division of the small tile indices
    // Grouped convolution (G independent convs of the same shape in one launch -- the three IRFD encoders): Cin / Cout
    // above are PER GROUP; x has Cx channels, group g reading [g*gin, g*gin + Cin) (gin = 0: every group reads the same
    // input); y, bias, stats, ... have Cy = G*Cout channels; blockIdx.y = g * co_tiles_g + co tile.  G = 1: Cx = gin =
    // Cin, Cy = Cout.
    int G, Cx, Cy, gin, co_tiles_g;
    int Hd, Wd;              // 2x2 (parity) kernels: spatial size of the interleaved destination
    int pshift;              // 2x2 (parity) kernels: 1 = window rows y-1, y / destination pixel (2y-1+py, 2x-1+px) -- the
                             // ConvTranspose2d(4, stride 2, pad 1) form; 0 = rows y, y+1 / (2y+py, 2x+px) -- the 3x3 s2 data gradient
    int staged;              // epilogue through LDS with 16-byte stores (set by the host when the tile / tensors allow it)
};

template <int WM_, int WN_, int MT_, int NT_, int CIT_>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_, CI_T = CIT_;
    static constexpr int NW = WM * WN, NTHREADS = NW * 64;
    static constexpr int CO_T = WM * MT * 32, PIX_T = WN * NT * 32;
    // fragment prefetch distance in k-steps: about 256 MFMA cycles of cover for the ds_read latency
    static constexpr int PD = MT * NT >= 4 ? 2 : 4;   // wanted distance; the kernel rounds it up so that PD + 1 divides the step count
};

// compile-time shape of one (config, kernel size, stride) instantiation
template <class C, int KH, int KW, int S>
struct Shape {
    static constexpr int TAPS = KH * KW, PAD = (KH - 1) / 2;
    static constexpr int W_FLOATS = TAPS * C::CI_T * C::CO_T;
    static constexpr int WV = (W_FLOATS / 4 + C::NTHREADS - 1) / C::NTHREADS;  // 16-B loads per thread per chunk
    static constexpr int STEPS = TAPS * (C::CI_T / 2);
    // largest input plane of a tile (TW = 32): register slots (64 elements each) a wave prefetches
    // stride of the staged tile in LDS: a 1x1 kernel stages only the pixels it samples, so its tile is dense
    static constexpr int SL = KH * KW == 1 ? 1 : S;
    static constexpr int PH_MAX = (C::PIX_T / 32 - 1) * SL + KH, PW_MAX = 31 * SL + KW;
    static constexpr int PPW = C::CI_T / C::NW > 0 ? C::CI_T / C::NW : 1;
    static constexpr int NSLOT = (PPW * PH_MAX * PW_MAX + 63) / 64;
};

// staging pieces per k-step so that a store phase of p_s pieces and a load phase of p_l pieces fit `steps` k-steps
constexpr int pieces_per_step(int p_s, int p_l, int steps) {
    for (int per = 1;; ++per)
        if ((p_s + per - 1) / per + (p_l + per - 1) / per <= steps) return per;
}

// fragment prefetch distance: the ring size PD + 1 must divide the number of k-steps of a chunk (so that ring indices
// line up across chunks) and leave room for the barrier before the cross-chunk reads; prefer the smallest such
// distance >= want, else the largest one below it
constexpr int pick_pd(int want, int steps) {
    for (int pd = want; pd + 1 <= steps / 2; ++pd)
        if (steps % (pd + 1) == 0) return pd;
    for (int pd = want - 1; pd >= 1; --pd)
        if (steps % (pd + 1) == 0) return pd;
    return 1;
}

__device__ __forceinline__ float w1_of(unsigned code) { return code == 1 ? 0.25f : (code == 2 ? 0.75f : 0.f); }
__device__ __forceinline__ float w0_of(unsigned code) { return code == 0 ? 1.f : (code == 2 ? 0.25f : 0.75f); }

// FG (fixed geometry): the pixel tile is the 32-wide one of every layer at least 32 pixels wide -- TW = 32, TH = PIX_T / 32,
// TB = 1 -- as compile-time constants, so that every LDS offset of the main loop (tap rows, channel planes, k-steps) is an
// instruction immediate off one base register per operand and chunk.  On gfx950 the f32 MFMA shares the SIMD's vector ALU
// (tools/mfma_valu_coexec.hip): each v_add_u32 that forms an LDS address costs ~5 of an MFMA's 64 cycles, and the runtime-
// geometry loop carries 32-50 of them per 72 MFMAs.  The host picks the FG instantiation when the layer's geometry is this one.
template <class C, int KH, int KW, int S, int MODE, bool FG = false>
__global__ __launch_bounds__(C::NTHREADS) void conv_kernel(const ConvArgs p) {
    using SH = Shape<C, KH, KW, S>;
    constexpr bool UPS = MODE == MODE_UPSAMPLE || MODE == MODE_UPSAMPLE_BATCH_SCALE, AFF = MODE == MODE_AFFINE_RELU;
    constexpr bool BSC = MODE == MODE_BATCH_SCALE || MODE == MODE_UPSAMPLE_BATCH_SCALE;
    constexpr int NSLOT = SH::NSLOT, W_FLOATS = SH::W_FLOATS, WV = SH::WV, PAD = SH::PAD;
    static_assert(!UPS || S == 1, "upsample folding needs stride 1");
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef SPK_LAB_CLOCK   // core clock under this kernel's load: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime)
    const long long lab_c0 = __builtin_readcyclecounter();
    const unsigned long long lab_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave / C::WN, wn = wave % C::WN;

    const int lgTW = FG ? 5 : p.lgTW, lgTH = FG ? (C::PIX_T == 256 ? 3 : (C::PIX_T == 128 ? 2 : (C::PIX_T == 64 ? 1 : 0))) : p.lgTH;
    const int lgTB = FG ? 0 : p.lgTB;
    static_assert(!FG || (C::PIX_T >= 32 && C::PIX_T <= 256), "fixed geometry: 32 x (PIX_T / 32) pixel tiles");
    const int TW = 1 << lgTW, TH = 1 << lgTH, TB = 1 << lgTB;
    constexpr int SL = SH::SL;
    const int PW = (TW - 1) * SL + KW, PLANE = ((TH - 1) * SL + KH) * PW;
    // FG: every wave stages ONE input plane (CI_T = waves, TB = 1) and the planes sit NSLOT * 64 floats apart, so that lane l's
    // element of gather slot s lives at plane + s * 64 + l for every lane (lanes past the plane write into the pitch's padding):
    // an LDS store is one base register + an immediate, and needs no dump slot
    static_assert(!FG || (C::CI_T == C::NW && (S == 1 || S == 2) && KH == 3), "fixed geometry: 3x3 stride 1 / 2, one plane per wave");
    const int PP = FG ? NSLOT * 64 : PLANE;
    const int IN_FLOATS = C::CI_T * TB * PP;
    const int BUF_FLOATS = W_FLOATS + ((IN_FLOATS + 3) & ~3) + 4;  // +4: dump slot for lanes without an element
    // XCD-aware tile order: the hardware deals workgroup i to XCD i % 8 (profiles/r01_e_workgroup_placement.txt), so in
    // launch order neighbouring pixel tiles land on different L2s and each re-fetches the halo rows they share.  Remap
    // so that every XCD walks ONE contiguous run of tiles (a bijection of [0, gridDim.x)): halos and the rows a tile's
    // successor needs are L2 hits.
    int bx;
    {
        const int n = (int)gridDim.x, q = n >> 3, r = n & 7;
        const int xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
        bx = xcd * q + min(xcd, r) + k;
    }
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int tbi = bx / p.tiles_y;
    const int b0 = tbi << lgTB, y0 = ty << lgTH, x0 = tx << lgTW;
    const int grp = (int)blockIdx.y / p.co_tiles_g;
    const int co_tile0 = ((int)blockIdx.y - grp * p.co_tiles_g) * C::CO_T;   // within the group
    const int c_begin = blockIdx.z * p.chunks_per_split;
    const int c_end = min(p.n_chunks, c_begin + p.chunks_per_split);
    // extent of the (virtual) input image the taps index: the x2-upsampled image in MODE_UPSAMPLE
    const int Hv = UPS ? 2 * p.Hs : p.Hs, Wv = UPS ? 2 * p.Ws : p.Ws;

    // ---- per-lane gather slots: wave w owns planes [w*ppw, (w+1)*ppw), contiguous in LDS.  Everything a chunk's
    // staging needs per slot is tabulated once, so that a staging piece is one or two instructions in the main loop:
    //   s_xoff / s_xoffl  byte offset of the element from the chunk's first input plane (always inside the tensor); the
    //                     *l variant serves the zero-padded last chunk, whose missing channels fold onto its channel 0
    //   s_vc              2: element exists in every chunk, 1: in every chunk but a ragged last one, 0: zero padding
    //   s_dst             byte offset of the element in an LDS ring slot (the dump float for lanes without an element)
    //   s_dxb/s_dyb, s_l* (MODE_UPSAMPLE) byte steps to the second column / row tap and the four tap weights
    //   s_sc / s_scl      (AFFINE_RELU, BATCH_SCALE) byte offset of the element's scale in in_scale (+ in_shift)
    const int ppw = (C::CI_T * TB) / C::NW;
    const int wave_elems = ppw * PLANE;
    const size_t src_plane = (size_t)p.Hs * p.Ws;
    const int ci_left_last = p.Cin - (p.n_chunks - 1) * C::CI_T;
    unsigned s_xoff[NSLOT], s_xoffl[FG ? 1 : NSLOT], s_dst[FG ? 1 : NSLOT], s_vc[FG ? 1 : NSLOT];
    float s_mask[(FG && !UPS) ? NSLOT : 1];               // FG: 1 / 0 (zero padding) -- multiplied in, or folded into the bilinear weights
    const unsigned dst_lane = (unsigned)(W_FLOATS + wave * PP + lane) * 4u;
    unsigned s_dxb[(UPS && !FG) ? NSLOT : 1], s_dyb[(UPS && !FG) ? NSLOT : 1];
    float s_lx0[UPS ? NSLOT : 1], s_lx1[UPS ? NSLOT : 1], s_ly0[UPS ? NSLOT : 1], s_ly1[UPS ? NSLOT : 1];
    unsigned s_sc[(AFF || BSC) ? NSLOT : 1], s_scl[(AFF || BSC) ? NSLOT : 1];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        s_xoff[s] = 0;
        bool live = false;                                    // the element exists (inside the image)
        if constexpr (!FG) {
            s_xoffl[s] = 0;
            s_vc[s] = 0;
            s_dst[s] = (unsigned)(BUF_FLOATS - 1) * 4u;
        }
        if (UPS) { if constexpr (!FG) s_dxb[s] = s_dyb[s] = 0; s_lx0[s] = s_ly0[s] = 1.f; s_lx1[s] = s_ly1[s] = 0.f; }
        if (AFF || BSC) s_sc[s] = s_scl[s] = 0;
        const int e = s * 64 + lane;
        if (e < wave_elems) {
            // e < 2^16 and PLANE, PW < 2^16: __umulhi with ceil(2^32/d) is the exact quotient (no integer-divide sequences)
            const int pl = FG ? e / PLANE : (p.magic_plane ? (int)__umulhi((unsigned)e, p.magic_plane) : e), pidx = e - pl * PLANE;
            const int q = wave * ppw + pl;
            const int ci = q >> lgTB, tb = q & (TB - 1);
            const int r = FG ? pidx / PW : (p.magic_pw ? (int)__umulhi((unsigned)pidx, p.magic_pw) : pidx), c = pidx - r * PW;
            const int psh = KH == 2 ? p.pshift : 0;
            const int uy = (y0 * SL + r) * (S / SL) - PAD - psh, ux = (x0 * SL + c) * (S / SL) - PAD - psh;
            const bool past = ci >= ci_left_last;
            if constexpr (!FG) s_dst[s] = (unsigned)(W_FLOATS + wave * wave_elems + e) * 4u;
            unsigned off = (unsigned)(grp * p.gin + ci) * (unsigned)src_plane;  // padding elements read any address inside the tensor
            if (uy >= 0 && uy < Hv && ux >= 0 && ux < Wv && b0 + tb < p.B) {
                live = true;
                if constexpr (!FG) s_vc[s] = past ? 1u : 2u;
                int goff;
                if (!UPS) {
                    goff = uy * p.Ws + ux;
                } else {
                    // torch area_pixel_compute_source_index(scale=0.5, align_corners=False):
                    // src = max(0.5*(dst+0.5)-0.5, 0); lambdas are exactly 0, 0.25 or 0.75
                    // SPK_CONV_UP_FIR1331: the same two taps, but a neighbour outside the image counts as zero
                    // (upfirdn2d, up = 2, FIR [1,3,3,1], pad (2,1)) instead of being clamped.
                    const bool zb = p.flags & SPK_CONV_UP_FIR1331;
                    const int iy0 = uy == 0 ? 0 : (uy - 1) >> 1, ix0 = ux == 0 ? 0 : (ux - 1) >> 1;
                    unsigned ly = uy == 0 ? (zb ? 3u : 0u) : ((uy & 1) ? 1u : 2u);
                    unsigned lx = ux == 0 ? (zb ? 3u : 0u) : ((ux & 1) ? 1u : 2u);
                    int gy = iy0, gx = ix0;
                    float wx0, wx1, wy0, wy1;
                    if constexpr (FG) {
                        // FG: the second column / row tap is ALWAYS one element / one row further (an instruction immediate and
                        // a second scalar base: no per-lane address arithmetic).  Where bilinear clamps that tap onto the first
                        // (last column / row), the pair moves one back and the whole weight goes to ITS second tap -- the same
                        // value, every read inside the plane (host: Hs, Ws >= 2 for this build).
                        // (zero border: the clamped-away neighbour counts as zero, the tap keeps its own weight)
                        wx0 = w0_of(lx); wx1 = w1_of(lx); wy0 = w0_of(ly); wy1 = w1_of(ly);
                        if (ix0 + 1 >= p.Ws) { gx = ix0 - 1; wx1 = zb ? wx0 : wx0 + wx1; wx0 = 0.f; }
                        if (iy0 + 1 >= p.Hs) { gy = iy0 - 1; wy1 = zb ? wy0 : wy0 + wy1; wy0 = 0.f; }
                    } else {
                        if (iy0 + 1 < p.Hs) s_dyb[FG ? 0 : s] = (unsigned)p.Ws * 4u; else if (zb) ly = 3u;
                        if (ix0 + 1 < p.Ws) s_dxb[FG ? 0 : s] = 4u; else if (zb) lx = 3u;
                        wx0 = w0_of(lx); wx1 = w1_of(lx); wy0 = w0_of(ly); wy1 = w1_of(ly);
                    }
                    s_lx0[s] = wx0; s_lx1[s] = wx1;
                    s_ly0[s] = wy0; s_ly1[s] = wy1;
                    goff = gy * p.Ws + gx;
                }
                off = (unsigned)((size_t)(tb * p.Cx + grp * p.gin + ci) * src_plane) + (unsigned)goff;
            }
            s_xoff[s] = off * 4u;
            if constexpr (!FG) s_xoffl[s] = (off - (past ? (unsigned)ci * (unsigned)src_plane : 0u)) * 4u;   // stays >= the group's first plane
            if (BSC) {   // modulation s[b,ci]: the per-sample input scale of a modulated convolution
                const unsigned b_ = (unsigned)min(b0 + tb, p.B - 1);
                s_sc[s] = (b_ * (unsigned)p.Cx + (unsigned)(grp * p.gin + ci)) * 4u;
                s_scl[s] = (b_ * (unsigned)p.Cx + (unsigned)(grp * p.gin) + (past ? 0u : (unsigned)ci)) * 4u;
            }
            if (AFF) { s_sc[s] = (unsigned)(grp * p.gin + ci) * 4u; s_scl[s] = (unsigned)(grp * p.gin + (past ? 0 : ci)) * 4u; }
        }
        if constexpr (FG) {
            if (UPS) {
                // the four tap weights as products (1 mul + 3 fma per element instead of 3 mul + 3 fma), zero for padding:
                // s_lx0 = w00, s_lx1 = w01 (row iy0), s_ly0 = w10, s_ly1 = w11 (row iy0 + 1)
                const float m = live ? 1.f : 0.f, x0 = s_lx0[s], x1 = s_lx1[s], y0w = s_ly0[s] * m, y1w = s_ly1[s] * m;
                s_lx0[s] = y0w * x0; s_lx1[s] = y0w * x1; s_ly0[s] = y1w * x0; s_ly1[s] = y1w * x1;
            } else {
                s_mask[s] = live ? 1.f : 0.f;
            }
        }
    }
    // weight vectors: lane offsets inside a chunk's packed block; the ragged last vector is clamped
    const unsigned w_tid16 = (unsigned)tid * 16u;
    const unsigned w_last16 = (unsigned)(min(tid + (WV - 1) * C::NTHREADS, W_FLOATS / 4 - 1) - (WV - 1) * C::NTHREADS) * 16u;
    const float* xblk = p.x + (size_t)b0 * p.Cx * src_plane;

    // ---- per-lane fragment addresses ----
    int b_off[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int pt = (wn * C::NT + n) * 32 + l32;
        const int px = pt & (TW - 1), py = (pt >> lgTW) & (TH - 1);
        const int tb = min(pt >> (lgTW + lgTH), TB - 1);  // pixel groups beyond the tile idle (results dropped)
        b_off[n] = W_FLOATS + half * TB * PP + tb * PP + py * SL * PW + px * SL;
    }
    const int a_off = half * C::CO_T + wm * C::MT * 32 + l32;
    const int ci_stride2 = 2 * TB * PP;

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int n = 0; n < C::NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wp) + ((size_t)blockIdx.y * p.n_chunks + c_begin) * (W_FLOATS / 4);

    // prefetch registers: the global loads of one chunk (weights: WV 16-B vectors; input: NSLOT gather slots)
    f32x4 wreg[WV];
    float xin[UPS ? 4 * NSLOT : NSLOT];
    float xsc[(AFF || BSC) ? NSLOT : 1], xsh[AFF ? NSLOT : 1];

    // Staging is written as macros over compile-time piece indices (not lambdas / functions over arrays) so that the
    // prefetch arrays stay in registers.  Every piece is branch-free: surplus weight lanes re-load / re-store the
    // last vector, lanes without an input element read a valid address and write the buffer's dump slot.  The chunk
    // loop body is therefore a single basic block and the pieces can be placed between individual MFMAs.
#define SPK_BYTES(ptr_, off_) (reinterpret_cast<const char*>(ptr_) + (off_))
#define SPK_LOAD_W(chunk_, i_)                                                                                \
    {                                                                                                         \
        const f32x4* wc_ = wsrc + (size_t)((chunk_) - c_begin) * (W_FLOATS / 4) + (i_) * C::NTHREADS; /* uniform */ \
        wreg[i_] = *reinterpret_cast<const f32x4*>(SPK_BYTES(wc_, (i_) == WV - 1 ? w_last16 : w_tid16));      \
    }
#define SPK_LOAD_X(chunk_, s_)                                                                                \
    {                                                                                                         \
        /* loads from always-valid addresses; masking happens at LDS-store time */                            \
        const float* xc = xblk + (size_t)(chunk_) * C::CI_T * src_plane; /* uniform */                        \
        const bool lastc_ = (chunk_) == p.n_chunks - 1;                                                       \
        /* FG: the offset register is made opaque per chunk (in place), so that (base + offset) is not hoisted as a 64-bit \
           per-lane pointer that then costs a 64-bit vector add per load: the loads keep the scalar-base + 32-bit-offset form */ \
        if constexpr (FG) asm volatile("" : "+v"(s_xoff[s_]));                                                \
        const unsigned vo_ = FG ? s_xoff[s_] : (lastc_ ? s_xoffl[FG ? 0 : (s_)] : s_xoff[s_]);                  \
        if (!UPS) {                                                                                           \
            xin[s_] = *reinterpret_cast<const float*>(SPK_BYTES(xc, vo_));                                    \
        } else if constexpr (FG) {   /* second taps: +1 element (an immediate), +1 row (a second scalar base) */ \
            const float* xr = xc + p.Ws;                                                                      \
            xin[4 * (s_) + 0] = *reinterpret_cast<const float*>(SPK_BYTES(xc, vo_));                          \
            xin[4 * (s_) + 1] = *reinterpret_cast<const float*>(SPK_BYTES(xc, vo_) + 4);                      \
            xin[4 * (s_) + 2] = *reinterpret_cast<const float*>(SPK_BYTES(xr, vo_));                          \
            xin[4 * (s_) + 3] = *reinterpret_cast<const float*>(SPK_BYTES(xr, vo_) + 4);                      \
        } else {                                                                                              \
            xin[4 * (s_) + 0] = *reinterpret_cast<const float*>(SPK_BYTES(xc, vo_));                          \
            xin[4 * (s_) + 1] = *reinterpret_cast<const float*>(SPK_BYTES(xc, vo_ + s_dxb[s_]));              \
            xin[4 * (s_) + 2] = *reinterpret_cast<const float*>(SPK_BYTES(xc, vo_ + s_dyb[s_]));              \
            xin[4 * (s_) + 3] = *reinterpret_cast<const float*>(SPK_BYTES(xc, vo_ + s_dyb[s_] + s_dxb[s_]));  \
        }                                                                                                     \
        if (BSC || AFF) {                                                                                     \
            const unsigned so_ = FG ? s_sc[s_] : (lastc_ ? s_scl[s_] : s_sc[s_]);   /* FG: no ragged chunk */   \
            xsc[s_] = *reinterpret_cast<const float*>(SPK_BYTES(p.in_scale + (size_t)(chunk_) * C::CI_T, so_)); \
            if (AFF) xsh[s_] = *reinterpret_cast<const float*>(SPK_BYTES(p.in_shift + (size_t)(chunk_) * C::CI_T, so_)); \
        }                                                                                                     \
    }
#define SPK_STORE_W(buf_, i_)                                                                                 \
    {                                                                                                         \
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(buf_) + (i_) * C::NTHREADS * 16 +                   \
                                  ((i_) == WV - 1 ? w_last16 : w_tid16)) = wreg[i_];                          \
    }
#define SPK_STORE_X(buf_, lastc_, s_)                                                                         \
    {                                                                                                         \
        float v;                                                                                              \
        if (!UPS) {                                                                                           \
            v = xin[s_];                                                                                      \
        } else if constexpr (FG) {                                                                            \
            v = fmaf(s_ly1[s_], xin[4 * (s_) + 3], fmaf(s_ly0[s_], xin[4 * (s_) + 2],                         \
                     fmaf(s_lx1[s_], xin[4 * (s_) + 1], s_lx0[s_] * xin[4 * (s_)])));                         \
        } else {                                                                                              \
            v = s_ly0[s_] * (s_lx0[s_] * xin[4 * (s_)] + s_lx1[s_] * xin[4 * (s_) + 1]) +                     \
                s_ly1[s_] * (s_lx0[s_] * xin[4 * (s_) + 2] + s_lx1[s_] * xin[4 * (s_) + 3]);                  \
        }                                                                                                     \
        if (AFF) v = fmaxf(v * xsc[s_] + xsh[s_], 0.f);                                                       \
        if (BSC) v *= xsc[s_];                                                                                \
        if constexpr (FG) {                                                                                   \
            if (!UPS) v *= s_mask[UPS ? 0 : (s_)];          /* UPS: the mask sits in the row weights */             \
            *((volatile lds_f32_t*)(buf_) + ((dst_lane >> 2) + (s_) * 64)) = v;                                   \
        } else {                                                                                              \
            *reinterpret_cast<float*>(reinterpret_cast<char*>(buf_) + s_dst[s_]) = s_vc[s_] > ((lastc_) ? 1u : 0u) ? v : 0.f; \
        }                                                                                                     \
    }
    // Weights by LDS-DMA (global_load_lds_dwordx4: global -> LDS without touching VGPRs; each wave copies 64-vector
    // = 1 KB blocks of the chunk's packed image to the same offset of a ring slot) when the image is a whole number
    // of such blocks; waves beyond the last block repeat it (same bytes to the same place).  Completion is tracked by
    // vmcnt: the issuing wave waits vmcnt(0) before the chunk barrier that publishes the slot.
    constexpr bool ROWLOOP = KH * KW > 9;
    constexpr bool DMA = !ROWLOOP && (W_FLOATS / 4) % 64 == 0;
#define SPK_DMA_W(chunk_, buf_, i_)                                                                           \
    {                                                                                                         \
        const int blk_ = min((i_) * C::NW + wave, W_FLOATS / 4 / 64 - 1);                                     \
        const f32x4* src_ = wsrc + (size_t)((chunk_) - c_begin) * (W_FLOATS / 4) + blk_ * 64 + lane;          \
        __builtin_amdgcn_global_load_lds(src_, reinterpret_cast<char*>(buf_) + blk_ * 1024, 16, 0, 0);        \
    }
    // piece j of a chunk's loads: gather slots first (they miss L2: longest flight), then weight vectors; W goes to
    // ``wbuf_`` directly when DMA, else into the prefetch registers
#define SPK_LOAD_PIECE(chunk_, wbuf_, j_)                                                                     \
    {                                                                                                         \
        if constexpr ((j_) < NSLOT) { SPK_LOAD_X(chunk_, ((j_) < NSLOT ? (j_) : 0)); }                        \
        else if constexpr (DMA) { SPK_DMA_W(chunk_, wbuf_, ((j_) < NSLOT ? 0 : (j_) - NSLOT)); }              \
        else { SPK_LOAD_W(chunk_, ((j_) < NSLOT ? 0 : (j_) - NSLOT)); }                                       \
    }
    // piece j of a chunk's LDS stores out of the prefetch registers: (weight vectors unless DMA, then) gather slots
#define SPK_STORE_PIECE(buf_, lastc_, j_)                                                                     \
    {                                                                                                         \
        if constexpr (!DMA && (j_) < WV) { SPK_STORE_W(buf_, ((j_) < WV ? (j_) : 0)); }                       \
        else { SPK_STORE_X(buf_, lastc_, (DMA ? (j_) : ((j_) < WV ? 0 : (j_) - WV))); }                       \
    }
    constexpr int P_L = WV + NSLOT, P_S = DMA ? NSLOT : WV + NSLOT;
    // VMEM instructions of one load piece (for the scheduling groups)
    constexpr int X_LOADS = (UPS ? 4 : 1) + (BSC ? 1 : 0) + (AFF ? 2 : 0);

    // ---- prologue: chunk c_begin -> ring slot 0; chunk c_begin+1 -> registers ----
    const int n_my = c_end - c_begin;
    float* ring0 = smem;
    float* ring1 = smem + BUF_FLOATS;
    float* ring2 = smem + 2 * BUF_FLOATS;
    if (n_my > 0) {
        static_for<0, P_L>([&](auto j) { SPK_LOAD_PIECE(c_begin, ring0, decltype(j)::value); });
        const bool l0 = c_begin == p.n_chunks - 1;
        static_for<0, P_S>([&](auto j) { SPK_STORE_PIECE(ring0, l0, decltype(j)::value); });
        const int c1 = min(c_begin + 1, c_end - 1);
        if (!ROWLOOP || n_my > 1) {   // a single-chunk rolled-loop problem (the 7x7 stem: Cin = 3) lives in slot 0 alone
            static_for<0, P_L>([&](auto j) { SPK_LOAD_PIECE(c1, ring1, decltype(j)::value); });
        }
    }
    if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's LDS-DMA blocks have landed
    __syncthreads();

    // ---- main loop -------------------------------------------------------------------------------------------
    // Invariant at the top of chunk i: ring slot i%3 holds chunk i (complete, visible); the prefetch registers hold
    // chunk i+1; ring slot (i+1)%3 was last read during chunk i-2.  During chunk i, between its MFMAs:
    //   steps [0, NST_S)   the registers are stored to slot (i+1)%3 (ds_write pieces: weights first, gathers last)
    //   end of step SB     s_waitcnt lgkmcnt(0) [+ vmcnt(0)] + s_barrier: slot (i+1)%3 becomes readable; every wave is
    //                      past chunk i-1, so slot (i+2)%3 may be overwritten
    //   steps (SB, ...)    the global loads of chunk i+2 are issued (VMEM pieces: gathers first -- they are the ones
    //                      that miss L2 -- so that they get the longest flight time): gathers into the registers,
    //                      weights into the registers or, by LDS-DMA, straight into slot (i+2)%3 -- those land before
    //                      the barrier of chunk i+1, a whole chunk later
    //   last PD steps      the fragment ring already reads chunk i+1 from slot (i+1)%3
    // so a wave issues MFMAs without interruption from the first chunk to the last: there is no barrier, no load
    // burst and no LDS-latency bubble at the chunk boundary.  Three slots are what allows the single barrier to sit
    // in the middle of a chunk.
    constexpr int ROWS = ROWLOOP ? KH : 1;
    constexpr int STEPS = SH::STEPS / ROWS;
    constexpr int PD = pick_pd(C::PD, STEPS);          // (PD + 1) divides STEPS: ring indices line up across chunks
    constexpr int PER = pieces_per_step(P_S, P_L, STEPS);   // staging pieces carried by one k-step
    constexpr int NST_S = (P_S + PER - 1) / PER, NST_L = (P_L + PER - 1) / PER;   // steps of the store / load phase
    constexpr int SB = NST_S - 1;                      // the barrier closes the last store step
    static_assert(ROWLOOP || (NST_S + NST_L <= STEPS && SB <= STEPS - PD - 1), "staging pieces do not fit the k-steps");
    float fa[PD + 1][C::MT], fb[PD + 1][C::NT];

    // (FG: LDS-typed volatile reads -- one ds_read_b32 with a 16-bit immediate per value; merged ds_read2_b32 forms reach 1 KB
    // and cost a vector add per pair of k-steps)
#define SPK_LOAD_FRAG(abuf_, bbuf_, step_, slot_)                                                             \
    {                                                                                                         \
        constexpr int tap_ = (step_) / (C::CI_T / 2), kk_ = (step_) % (C::CI_T / 2);                          \
        const int tapoff_ = (tap_ / KW) * PW + (tap_ % KW);                                                   \
        _Pragma("unroll") for (int m = 0; m < C::MT; ++m) {                                                   \
            if constexpr (FG) fa[slot_][m] = *((const volatile lds_f32_t*)(abuf_) + (a_off + (tap_ * C::CI_T + 2 * kk_) * C::CO_T + m * 32)); \
            else fa[slot_][m] = (abuf_)[a_off + (tap_ * C::CI_T + 2 * kk_) * C::CO_T + m * 32];               \
        }                                                                                                     \
        _Pragma("unroll") for (int n = 0; n < C::NT; ++n) {                                                   \
            if constexpr (FG) fb[slot_][n] = *((const volatile lds_f32_t*)(bbuf_) + (b_off[n] + kk_ * ci_stride2 + tapoff_)); \
            else fb[slot_][n] = (bbuf_)[b_off[n] + kk_ * ci_stride2 + tapoff_];                               \
        }                                                                                                     \
    }
#define SPK_MFMA_STEP(slot_)                                                                                  \
    _Pragma("unroll") for (int m = 0; m < C::MT; ++m)                                                         \
        _Pragma("unroll") for (int n = 0; n < C::NT; ++n)                                                     \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot_][m], fb[slot_][n], acc[m][n], 0, 0, 0);

    if constexpr (!ROWLOOP) {
        if (n_my > 0) static_for<0, PD>([&](auto i) { SPK_LOAD_FRAG(ring0, ring0, decltype(i)::value, decltype(i)::value); });
    }
    float *cur = ring0, *nxt = ring1, *oth = ring2;
#ifdef SPK_LAB_STEPTIME
    unsigned long long lab_ts[STEPS > 1 ? STEPS : 1];
#pragma unroll
    for (int k = 0; k < STEPS; ++k) lab_ts[k] = 0;
#endif
    for (int i = 0; i < n_my; ++i) {
        const int chunk = c_begin + i;
        const int chunk2 = min(chunk + 2, c_end - 1);
        const bool l1 = min(chunk + 1, c_end - 1) == p.n_chunks - 1;   // the chunk held in registers is the ragged last one
        if constexpr (ROWLOOP) {
            // large kernels (7x7): the tap rows stay a runtime loop of KW*CI_T/2 static steps; staging is not interleaved
            if (n_my > 1) {
                static_for<0, P_S>([&](auto j) { SPK_STORE_PIECE(nxt, l1, decltype(j)::value); });
                __syncthreads();
                static_for<0, P_L>([&](auto j) { SPK_LOAD_PIECE(chunk2, oth, decltype(j)::value); });
            }
#pragma unroll 1
            for (int row = 0; row < ROWS; ++row) {
                const float* abuf = cur + row * (KW * C::CI_T * C::CO_T);
                const float* bbuf = cur + row * PW;
                static_for<0, PD>([&](auto s) { SPK_LOAD_FRAG(abuf, bbuf, decltype(s)::value, decltype(s)::value % (PD + 1)); });
                static_for<0, STEPS>([&](auto s) {
                    constexpr int st = decltype(s)::value;
                    if constexpr (st + PD < STEPS) {
                        SPK_LOAD_FRAG(abuf, bbuf, st + PD, (st + PD) % (PD + 1));
                        __builtin_amdgcn_sched_group_barrier(0x100, C::MT + C::NT, 0);
                    }
                    SPK_MFMA_STEP(st % (PD + 1));
                    __builtin_amdgcn_sched_group_barrier(0x8, C::MT * C::NT, 0);
                });
            }
        } else {
            static_for<0, STEPS>([&](auto s) {
                constexpr int st = decltype(s)::value;
                // fragments PD steps ahead; the last PD steps read the next chunk's slot (readable since step SB)
                if constexpr (st + PD < STEPS) { SPK_LOAD_FRAG(cur, cur, st + PD, (st + PD) % (PD + 1)); }
                else { SPK_LOAD_FRAG(nxt, nxt, st + PD - STEPS, (st + PD) % (PD + 1)); }
                SPK_MFMA_STEP(st % (PD + 1));
                constexpr int sj = st * PER, lj = (st - NST_S) * PER;
                if constexpr (st < NST_S) {
                    static_for<sj, (sj + PER < P_S ? sj + PER : P_S)>([&](auto j) { SPK_STORE_PIECE(nxt, l1, decltype(j)::value); });
                }
                if constexpr (st >= NST_S && st < NST_S + NST_L) {
                    static_for<lj, (lj + PER < P_L ? lj + PER : P_L)>([&](auto j) { SPK_LOAD_PIECE(chunk2, oth, decltype(j)::value); });
                }
                // pin the interleave: one LDS fragment read behind each of the first MFMAs, the staging piece behind the last
                static_for<0, C::MT * C::NT>([&](auto q) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    if constexpr (decltype(q)::value < C::MT + C::NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                });
                if constexpr (C::MT * C::NT < C::MT + C::NT) __builtin_amdgcn_sched_group_barrier(0x100, C::MT + C::NT - C::MT * C::NT, 0);
                if constexpr (st < NST_S) __builtin_amdgcn_sched_group_barrier(0x200, (sj + PER < P_S ? PER : P_S - sj), 0);
                if constexpr (st >= NST_S && st < NST_S + NST_L) {
                    constexpr int np = (lj + PER < P_L ? PER : P_L - lj);
                    __builtin_amdgcn_sched_group_barrier(0x20, np * (X_LOADS > 1 ? X_LOADS : 1), 0);
                }
                if constexpr (st == SB) {
                    if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the DMA blocks of slot (i+1)%3 have landed
                    __syncthreads();
                }
                // A store piece's VALU half (the zero-select of padding elements, the folded affine) must stay in its own step:
                // unfenced, the scheduler gathers the selects of ALL pieces behind the first MFMAs of the chunk, and the
                // s_waitcnt vmcnt(0) in front of them makes step 0 wait for the gather issued two steps earlier (a full memory
                // latency) instead of each step waiting for a load that has been in flight for half a chunk.
                if constexpr (st < NST_S) __builtin_amdgcn_sched_barrier(0);
#ifdef SPK_LAB_STEPTIME
                if (i == 10) lab_ts[st] = __builtin_readcyclecounter();
#endif
            });
        }
        float* t = cur; cur = nxt; nxt = oth; oth = t;
    }
    if constexpr (ROWLOOP) __syncthreads();
#ifdef SPK_LAB_STEPTIME
    if (p.stats && lane == 0 && blockIdx.x == 8 && blockIdx.y == 0) {
#pragma unroll
        for (int k = 0; k < STEPS; ++k) reinterpret_cast<unsigned long long*>(p.stats)[wave * 64 + k] = lab_ts[k];
    }
#endif
#undef SPK_MFMA_STEP
#undef SPK_LOAD_FRAG
#undef SPK_STORE_PIECE
#undef SPK_LOAD_PIECE
#undef SPK_DMA_W
#undef SPK_STORE_X
#undef SPK_BYTES
#undef SPK_STORE_W
#undef SPK_LOAD_X
#undef SPK_LOAD_W

    // ---- epilogue ----
    // 2x2 kernels exist for one purpose, the data gradient of a 3x3 stride-2 conv by output parity: the Cout channels
    // are 4 classes q = (py, px) of Cout/4 channels (class-major), and channel (q, c) at window (y, x) is the gradient
    // at pixel (2y+py, 2x+px) of channel c in a [B, G*Cout/4, Hd, Wd] tensor
    constexpr bool UNSH = KH == 2;
    const bool split = gridDim.z > 1;
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_noise = p.flags & SPK_EPI_NOISE;
    const bool f_lrelu = p.flags & SPK_EPI_LRELU, f_style = p.flags & SPK_EPI_STYLE;
    // The statistics epilogue is compiled OUT of the 3x3 stride-1 plain / upsample / batch-scale instantiations -- the StyleGAN
    // decoders' hot kernels, which never ask for it: with its (runtime-dead) code inside, the 64 -> 64 @256^2 layer ran 3 %
    // slower (308 -> 318 us, +4 VGPRs; headline 3.71 -> 3.77 ms per step).  A plain 3x3 stride-1 conv that wants statistics
    // is the separate instantiation MODE_PLAIN_STATS (conv_inst_3x3s1_stats.hip).
    constexpr bool HAS_STATS = !(KH == 3 && KW == 3 && S == 1) || MODE == MODE_AFFINE_RELU || MODE == MODE_PLAIN_STATS;
    const bool f_accum = p.flags & SPK_EPI_ACCUM, f_stats = HAS_STATS && (p.flags & SPK_EPI_STATS) && !split;
    const size_t HW = (size_t)p.H * p.W;
    const float osc = p.out_scale_dev ? p.out_scale * *p.out_scale_dev : p.out_scale;   // (uniform: one scalar load)
    if (p.staged) {
        // ---- staged form (host: no split-K, not the parity kernel, W % 4 == 0, 16-byte aligned tensors, TW >= 4, and an
        // LDS tile that does not cost a workgroup slot): the block goes through LDS and a thread finishes 4 consecutive
        // pixels of a channel at a time -- 16 vector stores per lane instead of 64 dword stores, the bias / style /
        // demodulation operands loaded once per vector instead of once per element.  Same arithmetic per element.
        // (Knock-out measurement, profiles/r01_m_core_clock_under_load.txt: the dword epilogue is 8.7 % of the conv time.)
        // p.staged == 2 (configs whose waves hold two MFMA tiles of channels, e.g. 64co x 256px on the 256^2 layers): the block
        // goes in two halves -- every wave's upper tiles, then its lower ones -- so that the LDS tile stays within the ring
        // and the kernel keeps its workgroups per CU.
        constexpr int OP = C::PIX_T + 4, F4 = C::PIX_T / 4, RPI = C::NTHREADS / F4;
        constexpr bool CAN_HALVE = C::MT == 2;
        float* const ot = smem;
        const int rounds = (CAN_HALVE && p.staged == 2) ? 2 : 1;
        float* const red_s = ot + (C::CO_T / rounds) * OP;    // [2][CO_T] row sums / sums of squares (SPK_EPI_STATS)
        const int f4 = tid % F4, row0 = tid / F4;
        const int pt = 4 * f4;
        const int px = pt & (TW - 1), py = (pt >> lgTW) & (TH - 1), tb = pt >> (lgTW + lgTH);
        const int b = b0 + tb, yy = y0 + py, xx = x0 + px;
        const bool pvv = tb < TB && b < p.B && yy < p.H && xx < p.W;      // W % 4 == 0: the vector is in or out as a whole
        const size_t pix = (size_t)yy * p.W + xx;
        const size_t o0 = pvv ? (size_t)b * p.Cy * HW + pix : 0;
        float4 nzv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f_noise && pvv) nzv = *reinterpret_cast<const float4*>(p.noise + (size_t)b * HW + pix);
        const float* stp = (f_style && pvv) ? p.style + (size_t)b * p.style_stride : nullptr;
        double* sp = f_stats ? p.stats + (size_t)((int)blockIdx.x % p.stats_slots) * 2 * p.Cy : nullptr;
        const bool own_slot = p.stats_slots >= (int)gridDim.x;
        for (int h = 0; h < rounds; ++h) {
        __syncthreads();                                  // every wave is done reading the ring / the previous half
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
            if (rounds == 2 && m != h) continue;
            const int mrow = rounds == 2 ? wm * 32 : (wm * C::MT + m) * 32;
#pragma unroll
            for (int n = 0; n < C::NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ot[(mrow + (r & 3) + 8 * (r >> 2) + 4 * half) * OP + (wn * C::NT + n) * 32 + l32] = acc[m][n][r];
        }
        __syncthreads();
        const int rows = C::CO_T / rounds;
#pragma unroll 2
        for (int i = 0; i < rows / RPI; ++i) {
            const int cl = row0 + RPI * i;
            // row cl of a half belongs to wave row cl / 32, whose tiles m = 0, 1 sit 32 channels apart in the block
            const int co = co_tile0 + (rounds == 2 ? (cl >> 5) * 64 + h * 32 + (cl & 31) : cl);
            const bool cv = co < p.Cout;
            const int cg = grp * p.Cout + (cv ? co : 0);
            float ssum = 0.f, ssq = 0.f;
            if (cv && pvv) {
                float4 v = *reinterpret_cast<const float4*>(ot + cl * OP + pt);
                float sc = osc;
                v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                if (BSC && p.out_scale_bc) {
                    const float d = p.out_scale_bc[(size_t)b * p.Cy + cg];
                    v.x *= d; v.y *= d; v.z *= d; v.w *= d;
                }
                const float bb = f_bias ? p.bias[cg] : 0.f;
                v.x += bb; v.y += bb; v.z += bb; v.w += bb;
                if (f_noise) {
                    const float nwc = p.noise_w[cg];
                    v.x += nwc * nzv.x; v.y += nwc * nzv.y; v.z += nwc * nzv.z; v.w += nwc * nzv.w;
                }
                if (f_lrelu) {
                    v.x = (v.x > 0.f ? v.x : v.x * p.slope) * p.act_gain; v.y = (v.y > 0.f ? v.y : v.y * p.slope) * p.act_gain;
                    v.z = (v.z > 0.f ? v.z : v.z * p.slope) * p.act_gain; v.w = (v.w > 0.f ? v.w : v.w * p.slope) * p.act_gain;
                }
                const size_t off = o0 + (size_t)cg * HW;
                if (p.y_pre) *reinterpret_cast<float4*>(p.y_pre + off) = v;
                if (f_style) {
                    const float s0 = stp[cg] + 1.f, s1 = stp[p.Cy + cg];
                    v.x = v.x * s0 + s1; v.y = v.y * s0 + s1; v.z = v.z * s0 + s1; v.w = v.w * s0 + s1;
                }
                float4* dst = reinterpret_cast<float4*>(p.y + off);
                if (f_accum) {
                    const float4 old = *dst;
                    v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                }
                *dst = v;
                ssum = (v.x + v.y) + (v.z + v.w);
                ssq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            if (f_stats) {          // the F4 lanes of a row are consecutive: DPP adds finish the row's sums in its last lane
                ssum = spk::lane_group_sum_hi<F4>(ssum);
                ssq = spk::lane_group_sum_hi<F4>(ssq);
                if (f4 == F4 - 1) {
                    red_s[co - co_tile0] = ssum;
                    red_s[C::CO_T + co - co_tile0] = ssq;
                }
            }
        }
        }
        if (f_stats) {
            // the block's row sums leave as two contiguous runs of doubles instead of one 8-byte store per row and wave
            __syncthreads();
            for (int t = tid; t < 2 * C::CO_T; t += C::NTHREADS) {
                const int cl = t % C::CO_T, co = co_tile0 + cl;
                if (co < p.Cout) {
                    double* dst = sp + (t >= C::CO_T ? p.Cy : 0) + grp * p.Cout + co;
                    if (own_slot) *dst = (double)red_s[t];
                    else atomicAdd(dst, (double)red_s[t]);
                }
            }
        }
        return;
    }
    float* ybase = p.y + (split ? (size_t)blockIdx.z * p.B * p.Cy * HW : 0);
    // per pixel group: validity, output offset, noise value, style row
    bool pv[C::NT];
    size_t poff[C::NT];
    int pb[C::NT], uy2[C::NT], ux2[C::NT];
    float nz[C::NT];
    const float* st[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; ++n) {
        const int pt = (wn * C::NT + n) * 32 + l32;
        const int px = pt & (TW - 1), py = (pt >> lgTW) & (TH - 1), tb = pt >> (lgTW + lgTH);
        const int b = b0 + tb, yy = y0 + py, xx = x0 + px;
        pv[n] = tb < TB && b < p.B && yy < p.H && xx < p.W;
        const size_t pix = (size_t)yy * p.W + xx;
        poff[n] = pv[n] ? (size_t)b * p.Cy * HW + pix : 0;
        pb[n] = pv[n] ? b : 0;
        uy2[n] = 2 * yy;
        ux2[n] = 2 * xx;
        nz[n] = (f_noise && pv[n] && !split) ? p.noise[(size_t)b * HW + pix] : 0.f;
        st[n] = (f_style && pv[n]) ? p.style + (size_t)b * p.style_stride : nullptr;
    }
    // BatchNorm sums: the waves' per-channel partial sums meet in LDS (the ring is idle by now), so that a workgroup
    // issues ONE add per channel and moment -- fp64 atomics are what limits this epilogue (~10-20 per ns chip-wide)
#ifdef SPK_LAB_NOEPI   // lab: how much of a launch is the epilogue?  (skips it unless an impossible value turns up)
    if (acc[0][0][0] != 123456.789f) return;
#endif
    float* const red = smem;                              // [WN][CO_T][2]
    if (f_stats) __syncthreads();                         // every wave is done reading the ring
#pragma unroll
    for (int m = 0; m < C::MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_tile0 + (wm * C::MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool cv = co < p.Cout;
            const int cg = grp * p.Cout + co;              // channel in the output tensor (all groups)
            float ssum = 0.f, ssq = 0.f;
            if (cv) {
                // parity kernels: the bias belongs to the interleaved destination's channel (one value for the 4 classes)
                const float bb = (f_bias && !split) ? p.bias[UNSH ? grp * (p.Cout >> 2) + co % (p.Cout >> 2) : cg] : 0.f;
                const float nwc = (f_noise && !split) ? p.noise_w[cg] : 0.f;
#pragma unroll
                for (int n = 0; n < C::NT; ++n) {
                    if (!pv[n]) continue;
                    float* dst = ybase + poff[n] + (size_t)cg * HW;
                    if constexpr (UNSH) {
                        const int c4 = p.Cout >> 2, q = co / c4;
                        const int Y = uy2[n] + (q >> 1) - p.pshift, X = ux2[n] + (q & 1) - p.pshift;
                        if (Y < 0 || X < 0 || Y >= p.Hd || X >= p.Wd) continue;
                        dst = p.y + (((size_t)pb[n] * (p.G * c4) + grp * c4 + (co - q * c4)) * p.Hd + Y) * p.Wd + X;
                    }
                    if (split) {  // raw partial sums; splitk_epilogue_kernel finishes
                        *dst = acc[m][n][r];
                        continue;
                    }
                    float v = acc[m][n][r] * osc;
                    if (BSC && p.out_scale_bc) v *= p.out_scale_bc[(size_t)pb[n] * p.Cy + cg];   // demodulation d[b,co]
                    v += bb;
                    if (f_noise) v += nwc * nz[n];
                    if (f_lrelu) v = (v > 0.f ? v : v * p.slope) * p.act_gain;
                    if (p.y_pre) p.y_pre[poff[n] + (size_t)cg * HW] = v;
                    if (f_style) v = v * (st[n][cg] + 1.f) + st[n][p.Cy + cg];
                    if (f_accum) v += *dst;
                    *dst = v;
                    ssum += v;
                    ssq += v * v;
                }
            }
            if (f_stats) {  // wave-uniform flag: every lane takes part
                ssum = spk::half_wave_sum_hi(ssum);
                ssq = spk::half_wave_sum_hi(ssq);
                if (l32 == 31) {
                    float* rp = red + (wn * C::CO_T + (co - co_tile0)) * 2;
                    rp[0] = ssum;
                    rp[1] = ssq;
                }
            }
        }
    }
    if (f_stats) {
        __syncthreads();
        const int co = co_tile0 + tid;
        if (tid < C::CO_T && co < p.Cout) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < C::WN; ++w) {
                s += (double)red[(w * C::CO_T + tid) * 2];
                q += (double)red[(w * C::CO_T + tid) * 2 + 1];
            }
            const int cg = grp * p.Cout + co;
            double* sp = p.stats + (size_t)((int)blockIdx.x % p.stats_slots) * 2 * p.Cy;
            if (p.stats_slots >= (int)gridDim.x) {        // this pixel tile owns its copy (the caller zeroed it)
                sp[cg] = s;
                sp[p.Cy + cg] = q;
            } else {
                atomicAdd(sp + cg, s);
                atomicAdd(sp + p.Cy + cg, q);
            }
        }
    }
#ifdef SPK_LAB_CLOCK
    if (KH == 3 && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) {
        const long long c = __builtin_readcyclecounter() - lab_c0;
        const unsigned long long r = __builtin_amdgcn_s_memrealtime() - lab_r0;
        printf("conv clock: wg %u of %u (Cin %d, %dx%d): %lld cycles in %llu ticks of 10 ns = %.0f MHz\n", blockIdx.x, gridDim.x, p.Cin,
               p.H, p.W, c, r, r ? (double)c / (double)r * 100.0 : 0.0);
    }
#endif
}

// ---- host side -------------------------------------------------------------------------------------
struct Geometry {
    int TW, TH, TB, PLANE, tiles_x, tiles_y, tiles_b, n_chunks, co_tiles;
    size_t lds_bytes;
    bool ok;
};

template <class C, int KH, int KW, int S>
Geometry geometry(int B, int Cin, int Cout, int H, int W) {
    using SH = Shape<C, KH, KW, S>;
    Geometry g;
    g.TW = std::min(32, spk::pow2_ceil(W));
    g.TH = std::min(C::PIX_T / g.TW, spk::pow2_ceil(H));
    g.TB = C::PIX_T / (g.TW * g.TH);
    auto plane = [&]() { return ((g.TH - 1) * SH::SL + KH) * ((g.TW - 1) * SH::SL + KW); };
    // the wave's share of the input tile must fit its prefetch slots
    auto slots = [&]() { return spk::ceil_div(C::CI_T * g.TB / C::NW * plane(), 64); };
    while (slots() > SH::NSLOT && g.TB > 1 && (C::CI_T * (g.TB / 2)) % C::NW == 0) g.TB >>= 1;  // idle pixel groups
    while (slots() > SH::NSLOT && g.TH > 1) g.TH >>= 1;
    g.PLANE = plane();
    g.ok = slots() <= SH::NSLOT && (C::CI_T * g.TB) % C::NW == 0 && C::CI_T <= 64;
    g.tiles_x = spk::ceil_div(W, g.TW);
    g.tiles_y = spk::ceil_div(H, g.TH);
    g.tiles_b = spk::ceil_div(B, g.TB);
    g.n_chunks = spk::ceil_div(Cin, C::CI_T);
    g.co_tiles = spk::ceil_div(Cout, C::CO_T);
    const size_t in_floats = ((size_t)C::CI_T * g.TB * g.PLANE + 3) & ~(size_t)3;
    const bool one_slot = KH * KW > 9 && g.n_chunks == 1;   // rolled-loop kernel with a single chunk: no ring needed
    g.lds_bytes = (one_slot ? 1 : 3) * (SH::W_FLOATS + in_floats + 4) * sizeof(float);   // three-slot ring
    if (g.lds_bytes > 160 * 1024) g.ok = false;
    return g;
}

// number of ci-chunk slices so that the grid fills the chip (about 2 workgroups per CU)
inline int pick_ksplit(const Geometry& g) {
    const long long tiles = (long long)g.tiles_x * g.tiles_y * g.tiles_b * g.co_tiles;
    static const int target = [] { const char* e = getenv("SPK_KSPLIT_TARGET"); return e ? atoi(e) : 512; }();
    int ks = 1;
    while (tiles * ks < target && g.n_chunks / (ks * 2) >= 4 && ks < 64) ks *= 2;
    return ks;
}

inline int resolve_ksplit(const Geometry& g, int requested, int* chunks_per_split) {
    int ks = requested > 0 ? requested : pick_ksplit(g);
    ks = std::max(1, std::min(ks, g.n_chunks));
    const int cps = spk::ceil_div(g.n_chunks, ks);
    if (chunks_per_split) *chunks_per_split = cps;
    return spk::ceil_div(g.n_chunks, cps);
}

int launch_splitk_epilogue(const ConvArgs& a, const float* ws, int ksplit, hipStream_t stream);

// Fill ConvArgs from the public descriptor, pick geometry / split-K, launch (and the split-K epilogue).
template <class C, int KH, int KW, int S, int MODE, bool TRY_FG = false>
int run(const spk_conv2d_desc* d, hipStream_t stream, int Hd = 0, int Wd = 0, int pshift = 0) {
    ConvArgs a;
    a.Hd = Hd; a.Wd = Wd; a.pshift = pshift;
    a.x = d->x; a.wp = d->w_packed; a.bias = d->bias; a.noise_w = d->noise_w; a.noise = d->noise;
    a.style = d->style; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.stats = d->stats; a.stats_slots = d->stats_slots > 1 ? d->stats_slots : 1; a.y = d->y; a.y_pre = d->y_pre; a.out_scale_bc = d->out_scale_bc;
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W; a.Hs = d->Hin; a.Ws = d->Win;
    a.G = d->groups > 1 ? d->groups : 1;
    a.gin = a.G > 1 ? d->group_in_stride : d->Cin;
    a.Cx = a.gin * (a.G - 1) + d->Cin;
    a.Cy = a.G * d->Cout;
    Geometry g = geometry<C, KH, KW, S>(d->B, d->Cin, d->Cout, d->H, d->W);
    a.co_tiles_g = g.co_tiles;
    g.co_tiles *= a.G;                                   // grid.y and the split-K fill heuristic count every group
    SPK_REQUIRE(g.ok, "conv2d: config %d does not fit this shape (%dx%d, B=%d)", d->config, d->H, d->W, d->B);
    // the staging tables hold 32-bit BYTE offsets relative to the image group's first plane
    SPK_REQUIRE((size_t)g.TB * a.Cx * a.Hs * a.Ws < (1ull << 30), "conv2d: image group too large for 32-bit byte offsets");
    a.lgTW = spk::ilog2(g.TW); a.lgTH = spk::ilog2(g.TH); a.lgTB = spk::ilog2(g.TB);
    {
        const unsigned pw = (unsigned)((g.TW - 1) * Shape<C, KH, KW, S>::SL + KW), plane = (unsigned)g.PLANE;
        a.magic_plane = plane > 1 ? (unsigned)(((1ull << 32) + plane - 1) / plane) : 0u;   // 0: divisor 1
        a.magic_pw = pw > 1 ? (unsigned)(((1ull << 32) + pw - 1) / pw) : 0u;
        SPK_REQUIRE((size_t)C::CI_T * g.TB * g.PLANE < 65536, "conv2d: tile too large for the index arithmetic");
    }
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
    a.n_chunks = g.n_chunks;
    a.style_stride = d->style_stride; a.flags = d->flags; a.slope = d->lrelu_slope; a.out_scale = d->out_scale; a.act_gain = d->act_gain != 0.f ? d->act_gain : 1.f;
    a.out_scale_dev = d->out_scale_dev;
    const int ksplit = resolve_ksplit(g, d->ksplit, &a.chunks_per_split);
    // the fixed-geometry build of the same kernel, where the layer's tile is that one (and no chunk is ragged)
    bool use_fg = false;
    if constexpr (TRY_FG) {
        static const bool fg_on = [] { const char* e = getenv("SPK_CONV_FG"); return !e || atoi(e) != 0; }();
        use_fg = fg_on && g.TW == 32 && g.TH == C::PIX_T / 32 && g.TB == 1 && d->Cin % C::CI_T == 0 && d->Hin >= 2 && d->Win >= 2;
        if (use_fg)                       // planes NSLOT * 64 floats apart (see the kernel)
            g.lds_bytes = 3 * ((size_t)Shape<C, KH, KW, S>::W_FLOATS + (size_t)C::CI_T * Shape<C, KH, KW, S>::NSLOT * 64 + 4) * sizeof(float);
    }
    // staged epilogue: whenever its LDS tile costs no workgroup slot and the vectors are whole and aligned
    a.staged = 0;
    {
        static const bool allow = [] { const char* e = getenv("SPK_CONV_STAGED_EPI"); return !e || atoi(e) != 0; }();
        const size_t tile_bytes = (size_t)C::CO_T * (C::PIX_T + 4) * sizeof(float);
        const size_t red_bytes = (d->flags & SPK_EPI_STATS) ? 2 * (size_t)C::CO_T * sizeof(float) : 0;   // row sums behind the tile
        const size_t lds_now = g.lds_bytes;
        const auto aligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        const auto same_slots = [&](size_t bytes) { return (160 * 1024) / std::max(bytes, lds_now) == (160 * 1024) / lds_now; };
        constexpr bool can_halve = C::MT == 2;
        const int rounds = same_slots(tile_bytes + red_bytes) ? 1 : ((can_halve && same_slots(tile_bytes / 2 + red_bytes)) ? 2 : 0);
        if (allow && rounds && KH != 2 && ksplit == 1 && g.TW >= 4 && d->W % 4 == 0 && C::NTHREADS % (C::PIX_T / 4) == 0 &&
            (C::CO_T / rounds) % (C::NTHREADS / (C::PIX_T / 4)) == 0 && aligned(d->y) && aligned(d->y_pre) && aligned(d->noise)) {
            a.staged = rounds;
            g.lds_bytes = std::max(lds_now, tile_bytes / rounds + red_bytes);
        }
    }
    const size_t out_floats = (size_t)d->B * a.Cy * d->H * d->W;
    if (ksplit > 1) {
        SPK_REQUIRE(d->workspace && (size_t)d->workspace_bytes >= ksplit * out_floats * sizeof(float),
                    "conv2d: split-K x%d needs a %zu-byte workspace (see spk_conv2d_workspace_bytes)", ksplit,
                    ksplit * out_floats * sizeof(float));
        a.y = static_cast<float*>(d->workspace);
    }
    auto kern = &conv_kernel<C, KH, KW, S, MODE>;
    if constexpr (TRY_FG) {
        if (use_fg) kern = &conv_kernel<C, KH, KW, S, MODE, true>;
    }
    if (g.lds_bytes > 64 * 1024) {  // dynamic LDS above 64 KiB needs the attribute raised (once per kernel)
        static const void* raised[2] = {nullptr, nullptr};
        const void* kp = reinterpret_cast<const void*>(kern);
        if (raised[0] != kp && raised[1] != kp) {
            hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
            raised[raised[0] ? 1 : 0] = kp;
        }
    }
    const long long gx = (long long)g.tiles_x * g.tiles_y * g.tiles_b;
    SPK_REQUIRE(gx < (1ll << 31), "conv2d: grid too large");
    dim3 grid((unsigned)gx, (unsigned)g.co_tiles, (unsigned)ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(C::NTHREADS), g.lds_bytes, stream, a);
    int rc = spk::check_launch("conv_kernel");
    if (rc != SPK_OK || ksplit == 1) return rc;
    a.y = d->y;
    return launch_splitk_epilogue(a, static_cast<const float*>(d->workspace), ksplit, stream);
}

// ---- tile configs ------------------------------------------------------------------------------------
//                 WM WN MT NT CI_T      CO_T  PIX_T  used by
typedef Cfg<2, 2, 2, 2, 8> Cfg0;   //  128   128   3x3 s1
typedef Cfg<1, 4, 2, 2, 8> Cfg1;   //   64   256   3x3 s1
typedef Cfg<2, 2, 1, 1, 8> Cfg2;   //   64    64   3x3 s1
typedef Cfg<1, 4, 1, 1, 8> Cfg3;   //   32   128   3x3 s1
typedef Cfg<2, 2, 2, 2, 4> Cfg4;   //  128   128   3x3 s1/s2, 7x7 s2   (half-depth chunks: 2-3 WGs per CU)
typedef Cfg<1, 4, 2, 2, 4> Cfg5;   //   64   256   3x3 s1/s2, 7x7 s2
typedef Cfg<2, 2, 1, 1, 4> Cfg6;   //   64    64   3x3 s1/s2, 7x7 s2
typedef Cfg<1, 4, 1, 1, 4> Cfg7;   //   32   128   3x3 s1/s2, 7x7 s2
typedef Cfg<2, 2, 2, 2, 16> Cfg8;  //  128   128   1x1 s1/s2   (16-channel chunks keep the three-slot ring under 64 KB)
typedef Cfg<1, 4, 2, 2, 16> Cfg9;  //   64   256   1x1 s1/s2
typedef Cfg<2, 2, 1, 1, 16> Cfg10; //   64    64   1x1 s1/s2
typedef Cfg<1, 4, 1, 1, 16> Cfg11; //   32   128   1x1 s1/s2
constexpr int kNumConfigs = 17;   // 12 = the GEMM form of a stride-1 1x1 (conv1x1_gemm.hip): 128co x 128px block, 32-channel k-tiles
constexpr int kGemmConfig = 12;
constexpr int kDgradS2Config = 13;   // the exact-tap data gradient of a 3x3 stride-2 conv (dgrad3x3s2.hip): 64co x 128px block, 8-channel chunks
constexpr int kGemm2Config = 14;     // the three-per-CU GEMM form of a stride-1 1x1 (conv1x1_gemm2.hip): 128co x 128px block, 16-channel k-tiles
constexpr int kGemm2NarrowConfig = 15;   // the same with a 64co x 128px block (four per CU)
inline bool is_gemm2(int cfg) { return cfg == kGemm2Config || cfg == kGemm2NarrowConfig; }
constexpr int kStemConfig = 16;          // the 7x7 stride-2 stem, Cin 3 -> Cout 64 per group (conv7x7_stem.hip): K = 147 exactly, 16 x 32 pixel tiles

// several weight tensors of one shape on one launch (blockIdx.y = which): their packed images one after another, as a
// grouped conv launch reads them
struct PackList { const float* w[SPK_PACK_LIST_MAX]; };

// per-family dispatchers, one translation unit each (parallel compilation)
int run_3x3s1_a(int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s);  // ids 0-3
int run_3x3s1_b(int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s);  // ids 4-7
int run_3x3s1_stats(int cfg, const spk_conv2d_desc* d, hipStream_t s);      // ids 0-7, MODE_PLAIN_STATS
int run_3x3s2_7x7s2(int kh, int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s);  // ids 4-7
int run_1x1(int stride, int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s);      // ids 8-11
int run_2x2_parity(int cfg, const spk_conv2d_desc* d, int Hd, int Wd, int pshift, hipStream_t s);     // ids 0-3
int run_1x1_gemm(const spk_conv2d_desc* d, hipStream_t s);                                // id 12
int run_dgrad_s2_fused(const spk_conv2d_desc* d, hipStream_t s);                          // id 13
bool dgrad_s2_fused_takes(int B, int K, int Cc, int Hg, int Wg);
int dgrad_s2_ksplit(int B, int K, int Cc, int Hg, int Wg, int G);         // contraction slices of the exact-tap kernel (1 = none)
long long dgrad_s2_fused_packed_floats(int K, int Cc);
bool gemm1x1_takes(int kh, int stride, int Cin, int H, int W);
long long gemm1x1_pixel_tiles(int B, int H, int W);
int run_1x1_gemm2(const spk_conv2d_desc* d, hipStream_t s);                               // ids 14, 15
bool gemm2_takes(int kh, int stride, int Cin, int Cout, int H, int W);
int gemm2_co_tile(int config);
long long gemm2_pixel_tiles(int B, int H, int W);
long long gemm2_packed_floats(int config, int Cin, int Cout);
int pack_gemm2(const PackList& list, int n_list, float* w_packed, int Cin, int Cout, int config, int tf, hipStream_t stream);
int run_stem(const spk_conv2d_desc* d, hipStream_t s);                                    // id 16
bool stem_takes(int kh, int stride, int Cin, int Cout, int H, int W);
long long stem_packed_floats();
void stem_tiles(int H, int W, int* tiles_x, int* tiles_y);
int pack_stem(const PackList& list, int n_list, float* w_packed, int Cin, int Cout, int tf, hipStream_t stream);

}  // namespace spkconv
