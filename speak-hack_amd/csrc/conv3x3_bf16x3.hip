// 3x3 stride-1 convolution on the gfx950 BF16 matrix pipe with fp32-class accuracy: every fp32 operand is split into
// two bf16 halves, x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits), and a product is formed as
//     a * b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (the dropped a_lo*b_lo term is 2^-16 relative)
// with three v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator.  bf16 x bf16 products are exact in fp32 and the
// accumulation is fp32, so the only error is the 2^-16 truncation of the operands: 3e-5 rel-L2 through the whole 13-layer
// decoder against the reference (plain bf16: 1.3e-2; fp16 inputs: 1.8e-3; the bound is 1e-3).  The bf16 MFMA runs at 16x
// the rate of the exact-f32 MFMA (32 cycles for 32x32x16 against 64 for 32x32x2), so three of them per product are 5.3x
// the f32 pipe.  OPT-IN (SPK_CONV_BF16X3): the default path stays the exact f32 kernel of conv_mfma_f32.hpp.
//
// GEMM view, per ci-chunk of 16 channels and per tap:  D[co][pix] += A[co][16 ci] * X[16 ci][pix (+) tap]
//   A: weights, split at PACK time (spk_conv2d_pack_weights_bf16x3): image [co tile][chunk][hi/lo][tap][h][64 co][8 ci] of
//      bf16, so that the MFMA A fragment of lane l (row co = l & 31, k-group h = l >> 5) is ONE 16-byte LDS read and
//      consecutive lanes read consecutive 16-byte slots (conflict-free ds_read_b128); a chunk's image is one contiguous
//      36 KB block copied to LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no conversion at run time).
//   X: the input tile with halo, split while it is staged: a thread gathers the 8 channels of one k-group at one plane
//      position (8 coalesced dword loads, or 32 with the bilinear x2 folded in), converts, and stores two 16-byte slots
//      into [hi/lo][h][position][8 ci]: the B fragment of lane l (pixel l & 31, k-group l >> 5) at a tap is again one
//      conflict-free ds_read_b128.
//   D: 32x32 tiles, col = lane & 31 = pixel, row = output channel -- the same accumulator layout as the f32 kernel, so the
//      epilogue (out_scale, demodulation, bias, noise, LeakyReLU, style) is that kernel's, element for element.
// Block = 64 co x 256 pixels over a two-stage LDS ring (58 KB per stage: ONE workgroup per CU), one barrier per chunk, and
// EIGHT waves in two roles (two per SIMD):
//   * waves 0-3, the consumers: side by side along the pixels, each 2 x 2 MFMA tiles -- nothing but fragment reads and the
//     108 MFMAs per chunk (3456 cycles);
//   * waves 4-7, the producers: everything that fills the other stage -- the LDS-DMA of the next chunk's weights, the gathers
//     (issued a chunk ahead of their use), the split into hi / lo and the LDS stores, the x2 interpolation.
// With four do-everything waves (one per SIMD) nothing overlapped: knock-out timing gave MFMA + fragments 79 us, staging 46,
// epilogue 13 for 256 -> 256 @ 64^2 and the whole kernel took their SUM, 137 (the compiler keeps the 108 MFMAs of a chunk in one
// block and the staging in another).  Two waves of different roles on a SIMD overlap by themselves.  The roles run separate
// loops (the same number of barriers), so the consumers' registers hold no staging state and the producers' no accumulators.
#include "spk_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace spkbf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CO_T = 64, PIX_T = 256, CI_T = 16, NT = 256, TAPS = 9;      // NT: threads per ROLE
constexpr int NTB = 2 * NT;                                                  // threads per workgroup (consumers + producers)
constexpr int W_BYTES = 2 * TAPS * 2 * CO_T * 16;        // one chunk's weight image: [hi/lo][tap][h][co][8 bf16]
constexpr int W_DMA = W_BYTES / (NT * 16);                // LDS-DMA instructions per thread and chunk (9)
constexpr int MAX_ROUNDS = 3;                             // gather rounds per chunk: 2 * NPOS items <= 768
static_assert(W_BYTES % (NT * 16) == 0, "whole DMA rounds");

struct Args {
    const float* x;
    const char* wp;
    const float* bias;
    const float* noise_w;
    const float* noise;
    const float* style;
    const float* in_scale;       // [B,Cin] modulation (SPK_CONV_IN_BATCH_SCALE) or null
    const float* out_scale_bc;   // [B,Cout] demodulation or null
    float* y;
    float* y_pre;                // optional: the value before the style stage (kept for the backward pass of a training forward)
    int B, Cin, Cout, H, W, Hs, Ws;
    int lgTW, lgTH, lgTB;
    int tiles_x, tiles_y;
    int n_chunks;
    int style_stride;
    unsigned flags;
    float slope, out_scale, act_gain;
    int staged;                  // epilogue through LDS with 16-byte stores (host: W % 4 == 0, TW >= 4, aligned tensors)
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {       // two RNE conversions, lo in the low half
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// 8 fp32 -> 8 bf16 "hi" + 8 bf16 "lo" (the residuals), each as four packed dwords
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned ph = pack_bf16(v[2 * q], v[2 * q + 1]);
        const float h0 = __uint_as_float(ph << 16), h1 = __uint_as_float(ph & 0xffff0000u);
        hi[q] = ph;
        lo[q] = pack_bf16(v[2 * q] - h0, v[2 * q + 1] - h1);
    }
}

template <bool UPS>
__global__ __launch_bounds__(NTB) void conv3x3_bf16x3_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid_all = threadIdx.x, tid = tid_all & (NT - 1), lane = tid & 63;     // tid: index within the role
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                       // wave within the role (0-3)
    const bool producer = __builtin_amdgcn_readfirstlane(tid_all >> 8) != 0;
    const int half = lane >> 5, l32 = lane & 31;

    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH, TB = 1 << p.lgTB;
    const int PW = TW + 2, PLANE = (TH + 2) * PW, NPOS = TB * PLANE;
    const int X_BYTES = 4 * NPOS * 16;                    // [hi/lo][h][pos][8 bf16]
    const int STAGE = W_BYTES + X_BYTES;

    // XCD-aware tile order (as the f32 kernel): every XCD walks one contiguous run of pixel tiles
    int bx;
    {
        const int n = (int)gridDim.x, q = n >> 3, r = n & 7;
        const int xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
        bx = xcd * q + min(xcd, r) + k;
    }
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y, tbi = bx / p.tiles_y;
    const int b0 = tbi << p.lgTB, y0 = ty << p.lgTH, x0 = tx << p.lgTW;
    const int co_tile = blockIdx.y, co0 = co_tile * CO_T;
    const int Hv = UPS ? 2 * p.Hs : p.Hs, Wv = UPS ? 2 * p.Ws : p.Ws;
    const size_t src_plane = (size_t)p.Hs * p.Ws;
    const float* xblk = p.x + (size_t)b0 * p.Cin * src_plane;
    const bool bsc = p.flags & SPK_CONV_IN_BATCH_SCALE;
    const bool zb = p.flags & SPK_CONV_UP_FIR1331;
    const int ci_last = p.Cin - (p.n_chunks - 1) * CI_T;      // channels of the ragged last chunk

    // ---- gather items: item = h * NPOS + pos (a k-group of 8 channels at one plane position); thread t owns items t + 256 r ----
    unsigned it_off[MAX_ROUNDS], it_dst[MAX_ROUNDS], it_ok[MAX_ROUNDS], it_sc[MAX_ROUNDS];
    int it_h[MAX_ROUNDS];
    // UPS: the x2 interpolation reads a low-resolution SOURCE tile kept in LDS as fp32 ([h][source position][8 channels], 32
    // bytes per slot): it_s00 = byte offset of the item's first tap (iy0, ix0) in a source buffer, it_dx / it_dy = byte steps
    // to the second column / row tap (0 where the tap is clamped away), the four tap weights.
    constexpr int MAX_SR = 2;                               // source-tile gather rounds: 2 * SNPOS items <= 512
    const int SPW = (TW >> 1) + 2, SPLANE = ((TH >> 1) + 2) * SPW, SNPOS = TB * SPLANE;
    const int S_BYTES = 2 * SNPOS * 32;
    unsigned it_s00[UPS ? MAX_ROUNDS : 1], it_dx[UPS ? MAX_ROUNDS : 1], it_dy[UPS ? MAX_ROUNDS : 1];
    float it_lx0[UPS ? MAX_ROUNDS : 1], it_lx1[UPS ? MAX_ROUNDS : 1], it_ly0[UPS ? MAX_ROUNDS : 1], it_ly1[UPS ? MAX_ROUNDS : 1];
#pragma unroll
    for (int r = 0; r < MAX_ROUNDS; ++r) {
        const int item = tid + NT * r;
        it_off[r] = 0; it_ok[r] = 0; it_h[r] = 0; it_sc[r] = 0;
        it_dst[r] = 0xffffffffu;                          // no item: nothing is stored
        if (UPS) { it_s00[r] = it_dx[r] = it_dy[r] = 0; it_lx0[r] = it_ly0[r] = 1.f; it_lx1[r] = it_ly1[r] = 0.f; }
        if (item < 2 * NPOS) {
            const int h = item / NPOS, pos = item - h * NPOS;
            const int tb = pos / PLANE, pidx = pos - tb * PLANE;
            const int pr = pidx / PW, pc = pidx - pr * PW;
            const int uy = y0 + pr - 1, ux = x0 + pc - 1;
            it_h[r] = h;
            // padding positions still LOAD (masked at store time): from pixel 0 of the group's first image, at the item's own
            // channels -- the offsets always carry the 8 h channel term, because a dead channel of the ragged last chunk is
            // redirected by subtracting it again (issue_loads)
            it_off[r] = (unsigned)((size_t)(8 * h) * src_plane);
            it_sc[r] = (unsigned)(8 * h);
            it_dst[r] = (unsigned)(W_BYTES + (h * NPOS + pos) * 16);           // + 2 NPOS 16 for the lo slot
            if (uy >= 0 && uy < Hv && ux >= 0 && ux < Wv && b0 + tb < p.B) {
                it_ok[r] = 1;
                int goff;
                if (!UPS) {
                    goff = uy * p.Ws + ux;
                } else {
                    // bilinear x2, align_corners=False: lambdas 0 / 0.25 / 0.75 (see conv_mfma_f32.hpp); UP_FIR1331: a
                    // neighbour outside the image counts as zero instead of being clamped
                    const int iy0 = uy == 0 ? 0 : (uy - 1) >> 1, ix0 = ux == 0 ? 0 : (ux - 1) >> 1;
                    unsigned ly = uy == 0 ? (zb ? 3u : 0u) : ((uy & 1) ? 1u : 2u);
                    unsigned lx = ux == 0 ? (zb ? 3u : 0u) : ((ux & 1) ? 1u : 2u);
                    if (iy0 + 1 < p.Hs) it_dy[r] = (unsigned)(SPW * 32); else if (zb) ly = 3u;
                    if (ix0 + 1 < p.Ws) it_dx[r] = 32u; else if (zb) lx = 3u;
                    // the source tile starts at source pixel (y0/2 - 1, x0/2 - 1)
                    it_s00[r] = (unsigned)((h * SNPOS + tb * SPLANE + (iy0 - (y0 >> 1) + 1) * SPW + (ix0 - (x0 >> 1) + 1)) * 32);
                    it_lx0[r] = lx == 0 ? 1.f : (lx == 2 ? 0.25f : 0.75f); it_lx1[r] = lx == 1 ? 0.25f : (lx == 2 ? 0.75f : 0.f);
                    it_ly0[r] = ly == 0 ? 1.f : (ly == 2 ? 0.25f : 0.75f); it_ly1[r] = ly == 1 ? 0.25f : (ly == 2 ? 0.75f : 0.f);
                    goff = iy0 * p.Ws + ix0;
                }
                it_off[r] = (unsigned)((size_t)(tb * p.Cin + 8 * h) * src_plane) + (unsigned)goff;
                it_sc[r] = (unsigned)(min(b0 + tb, p.B - 1) * p.Cin + 8 * h);
            }
        }
    }
    const int rounds = (2 * NPOS + NT - 1) / NT;            // uniform (<= MAX_ROUNDS: host-checked)
    // UPS: source-tile items (h, source position): thread t owns items t + 256 r.  Positions outside the source image load a
    // clamped (finite) neighbour: the interpolation gives them weight zero.
    unsigned si_off[UPS ? MAX_SR : 1], si_dst[UPS ? MAX_SR : 1], si_sc[UPS ? MAX_SR : 1];
    int si_h[UPS ? MAX_SR : 1];
    const int s_rounds = UPS ? (2 * SNPOS + NT - 1) / NT : 0;
    if constexpr (UPS) {
#pragma unroll
        for (int r = 0; r < MAX_SR; ++r) {
            const int item = tid + NT * r;
            si_off[r] = 0; si_dst[r] = 0xffffffffu; si_sc[r] = 0; si_h[r] = 0;
            if (item < 2 * SNPOS) {
                const int h = item / SNPOS, sp = item - h * SNPOS;
                const int tb = sp / SPLANE, sidx = sp - tb * SPLANE;
                const int sr = sidx / SPW, sc = sidx - sr * SPW;
                const int iy = min(max((y0 >> 1) - 1 + sr, 0), p.Hs - 1), ix = min(max((x0 >> 1) - 1 + sc, 0), p.Ws - 1);
                const int tbc = b0 + tb < p.B ? tb : 0;
                si_h[r] = h;
                si_dst[r] = (unsigned)((h * SNPOS + sp) * 32);
                si_off[r] = (unsigned)((size_t)(tbc * p.Cin + 8 * h) * src_plane) + (unsigned)(iy * p.Ws + ix);
                si_sc[r] = (unsigned)((b0 + tbc) * p.Cin + 8 * h);
            }
        }
    }

    // ---- fragment addresses (bytes inside a stage) ----
    // A: W image [hl][tap][h][co][16 B]; this lane: row co = m * 32 + l32, k-group `half`
    const unsigned a_base = (unsigned)((half * CO_T + l32) * 16);
    // B: X image [hl][h][pos][16 B]; this lane: pixel n * 32 + l32 of the wave's 64 pixels, k-group `half`
    unsigned b_base[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int pt = (wave * 2 + n) * 32 + l32;
        const int px = pt & (TW - 1), py = (pt >> p.lgTW) & (TH - 1);
        const int tb = min(pt >> (p.lgTW + p.lgTH), TB - 1);
        b_base[n] = (unsigned)(W_BYTES + (half * NPOS + tb * PLANE + py * PW + px) * 16);
    }
    const unsigned hl_w = TAPS * 2 * CO_T * 16, hl_x = (unsigned)(2 * NPOS * 16);

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const char* wsrc = p.wp + (size_t)co_tile * p.n_chunks * W_BYTES;

    // gather registers (producers): per round 8 channels of one plane position.  All rounds of a chunk are issued a whole chunk
    // before they are converted and stored (see the producers' loop)
    constexpr int GXN = 8;                                  // (the UPS path stages a source tile instead: s_issue / interp_store)
    float gxa[MAX_ROUNDS][GXN];
    float gsa[MAX_ROUNDS][8];
    auto issue_loads = [&](int chunk, auto r_) {
        constexpr int r = decltype(r_)::value;
        float (&gx)[GXN] = gxa[r];
        float (&gsc)[8] = gsa[r];
        // chunk's channels [16 chunk, 16 chunk + 16); the ragged last chunk clamps missing channels onto a valid one (masked at store)
        const bool lastc = chunk == p.n_chunks - 1;
        const float* xc = xblk + (size_t)chunk * CI_T * src_plane;
        const int h8 = 8 * it_h[r];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = !lastc || h8 + j < ci_last;
            const float* src = xc + it_off[r] + (ptrdiff_t)(live ? j : -h8) * (ptrdiff_t)src_plane;   // dead channel: the chunk's first plane
            gx[j] = *src;
            if (bsc) gsc[j] = p.in_scale[it_sc[r] + chunk * CI_T + (live ? j : -h8)];
        }
    };
    auto convert_store = [&](char* stage, int chunk, auto r_) {
        constexpr int r = decltype(r_)::value;
        float (&gx)[GXN] = gxa[r];
        float (&gsc)[8] = gsa[r];
        if (it_dst[r] == 0xffffffffu) return;
        const bool lastc = chunk == p.n_chunks - 1;
        const int h8 = 8 * it_h[r];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = gx[j];
            if (bsc) t *= gsc[j];
            v[j] = (it_ok[r] && (!lastc || h8 + j < ci_last)) ? t : 0.f;
        }
        u32x4 hi, lo;
        split8(v, hi, lo);
        *reinterpret_cast<u32x4*>(stage + it_dst[r]) = hi;
        *reinterpret_cast<u32x4*>(stage + it_dst[r] + hl_x) = lo;
    };
    // ---- UPS: source tile -> LDS (fp32, modulation applied here), then LDS -> LDS interpolation + split into the X image ----
    // s_issue only LOADS (the pixel and, for a modulated conv, its channel scale): any arithmetic on the loaded values here
    // gets an s_waitcnt vmcnt(0) in front of it -- 16 full memory latencies per chunk and wave before the first MFMA (ISA of
    // the first version, which multiplied in place).  The product is formed in s_store, a chunk of MFMAs later.
    float sga[UPS ? MAX_SR : 1][8], ssa[UPS ? MAX_SR : 1][8];
    auto s_issue = [&](int chunk, auto r_) {
        constexpr int r = decltype(r_)::value;
        const bool lastc = chunk == p.n_chunks - 1;
        const float* xc = xblk + (size_t)chunk * CI_T * src_plane;
        const int h8 = 8 * si_h[r];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = !lastc || h8 + j < ci_last;
            sga[r][j] = xc[si_off[r] + (ptrdiff_t)(live ? j : -h8) * (ptrdiff_t)src_plane];
        }
        if (bsc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool live = !lastc || h8 + j < ci_last;
                ssa[r][j] = p.in_scale[si_sc[r] + chunk * CI_T + (live ? j : -h8)];
            }
        }
    };
    auto s_store = [&](char* sbuf, int chunk, auto r_) {
        constexpr int r = decltype(r_)::value;
        if (si_dst[r] == 0xffffffffu) return;
        const bool lastc = chunk == p.n_chunks - 1;
        const int h8 = 8 * si_h[r];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sga[r][j];
        if (bsc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= ssa[r][j];
        }
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = (!lastc || h8 + j < ci_last) ? v[j] : 0.f;
            b[j] = (!lastc || h8 + 4 + j < ci_last) ? v[4 + j] : 0.f;
        }
        *reinterpret_cast<f32x4*>(sbuf + si_dst[r]) = a;
        *reinterpret_cast<f32x4*>(sbuf + si_dst[r] + 16) = b;
    };
    auto interp_store = [&](char* stage, const char* sbuf, auto r_) {
        constexpr int r = decltype(r_)::value;
        if (it_dst[r] == 0xffffffffu) return;
        float v[8];
        const char* s00 = sbuf + it_s00[r];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(s00 + 16 * q);
            const f32x4 b = *reinterpret_cast<const f32x4*>(s00 + it_dx[r] + 16 * q);
            const f32x4 c = *reinterpret_cast<const f32x4*>(s00 + it_dy[r] + 16 * q);
            const f32x4 d = *reinterpret_cast<const f32x4*>(s00 + it_dy[r] + it_dx[r] + 16 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = it_ly0[r] * (it_lx0[r] * a[j] + it_lx1[r] * b[j]) + it_ly1[r] * (it_lx0[r] * c[j] + it_lx1[r] * d[j]);
                v[4 * q + j] = it_ok[r] ? t : 0.f;
            }
        }
        u32x4 hi, lo;
        split8(v, hi, lo);
        *reinterpret_cast<u32x4*>(stage + it_dst[r]) = hi;
        *reinterpret_cast<u32x4*>(stage + it_dst[r] + hl_x) = lo;
    };

    auto dma_weights = [&](char* stage, int chunk) {
        const char* src = wsrc + (size_t)chunk * W_BYTES;
#pragma unroll
        for (int i = 0; i < W_DMA; ++i) {
            const int blk = i * 4 + wave;                  // 1 KB block of the chunk image
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(src + blk * 1024) + lane, stage + blk * 1024, 16, 0, 0);
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    char* const sb0 = smem + 2 * STAGE;                     // UPS: the two source-tile buffers sit behind the stages
    const int n_chunks = p.n_chunks;
    // vmcnt(N): the producers' LDS-DMA blocks are issued BEFORE the gathers of a later chunk, and loads return in order, so
    // waiting until only the N gather instructions are outstanding means the weight image has landed
    auto wait_dma = [&](int gathers) {
        switch (gathers) {                                  // simm16: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
            case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
            case 8: __builtin_amdgcn_s_waitcnt(0x0f78); break;
            case 16: __builtin_amdgcn_s_waitcnt(0x4f70); break;
            case 24: __builtin_amdgcn_s_waitcnt(0x4f78); break;
            case 32: __builtin_amdgcn_s_waitcnt(0x8f70); break;
            case 48: __builtin_amdgcn_s_waitcnt(0xcf70); break;
            default: __builtin_amdgcn_s_waitcnt(0x0f70); break;
        }
    };

    // ---- prologue (producers): chunk 0 -> stage 0; the gathers of the chunk after the next one into the registers ----
    if (producer) dma_weights(smem, 0);
    if constexpr (UPS) {
        if (producer) {
            // source tiles of chunks 0 and 1, then X(0) by interpolation out of S(0); S(2) stays in the registers
            s_issue(0, I0{});
            if (s_rounds > 1) s_issue(0, I1{});
            s_store(sb0, 0, I0{});
            if (s_rounds > 1) s_store(sb0, 0, I1{});
            if (n_chunks > 1) {
                s_issue(1, I0{});
                if (s_rounds > 1) s_issue(1, I1{});
                s_store(sb0 + S_BYTES, 1, I0{});
                if (s_rounds > 1) s_store(sb0 + S_BYTES, 1, I1{});
            }
        }
        __syncthreads();
        if (producer) {
            interp_store(smem, sb0, I0{});
            if (rounds > 1) interp_store(smem, sb0, I1{});
            if (rounds > 2) interp_store(smem, sb0, I2{});
        }
    } else {
        if (producer) {
            issue_loads(0, I0{});
            if (rounds > 1) issue_loads(0, I1{});
            if (rounds > 2) issue_loads(0, I2{});
            convert_store(smem, 0, I0{});
            if (rounds > 1) convert_store(smem, 0, I1{});
            if (rounds > 2) convert_store(smem, 0, I2{});
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): this wave's LDS-DMA blocks have landed
    __syncthreads();

    if (producer) {
        // ---- producers: during chunk i (the consumers' MFMAs out of stage i & 1) stage (i + 1) & 1 is filled ----
        if constexpr (UPS) {
            if (n_chunks > 2) {                             // S(2) -> registers
                s_issue(2, I0{});
                if (s_rounds > 1) s_issue(2, I1{});
            }
        } else {
            if (n_chunks > 1) {                             // the gathers of chunk 1 -> registers
                issue_loads(1, I0{});
                if (rounds > 1) issue_loads(1, I1{});
                if (rounds > 2) issue_loads(1, I2{});
            }
        }
        for (int i = 0; i < n_chunks; ++i) {
            char* nxt = smem + ((i + 1) & 1) * STAGE;
            const bool more = i + 1 < n_chunks;             // uniform
            int pending = 0;                                // gather instructions issued after the DMA in this iteration
            if (more) dma_weights(nxt, i + 1);
            __builtin_amdgcn_sched_barrier(0);              // the DMA stays in front of this iteration's gathers (wait_dma counts on it)
            if constexpr (UPS) {
                char* const s_nxt = sb0 + ((i + 1) & 1) * S_BYTES;       // holds S(i + 1) since the previous barrier
                char* const s_nn = sb0 + (i & 1) * S_BYTES;              // S(i) was last read while X(i) was built: free
                if (i + 2 < n_chunks) {                     // registers (S(i + 2), loaded a chunk ago) -> LDS
                    __builtin_amdgcn_sched_barrier(0);
                    s_store(s_nn, i + 2, I0{});
                    if (s_rounds > 1) s_store(s_nn, i + 2, I1{});
                }
                if (more) {                                 // X(i + 1) out of S(i + 1)
                    interp_store(nxt, s_nxt, I0{});
                    if (rounds > 1) interp_store(nxt, s_nxt, I1{});
                    if (rounds > 2) interp_store(nxt, s_nxt, I2{});
                }
                if (i + 3 < n_chunks) {
                    s_issue(i + 3, I0{});
                    if (s_rounds > 1) s_issue(i + 3, I1{});
                    pending = 8 * s_rounds;         // (the pixel gathers alone: a conservative count if scale loads get merged)
                }
            } else {
                if (more) {                                 // registers (chunk i + 1, loaded a chunk ago) -> split -> LDS
                    __builtin_amdgcn_sched_barrier(0);
                    convert_store(nxt, i + 1, I0{});
                    if (rounds > 1) convert_store(nxt, i + 1, I1{});
                    if (rounds > 2) convert_store(nxt, i + 1, I2{});
                }
                if (i + 2 < n_chunks) {
                    issue_loads(i + 2, I0{});
                    if (rounds > 1) issue_loads(i + 2, I1{});
                    if (rounds > 2) issue_loads(i + 2, I2{});
                    pending = 8 * rounds;
                }
            }
            wait_dma(pending);                              // the next stage's weight image has landed (the gathers fly on)
            __syncthreads();
        }
    } else {
        // ---- consumers: MFMAs of chunk i out of stage i & 1 ----
        bf16x8 fa[2][2][2], fb[2][2][2];                    // [slot][hi/lo][tile]
#define SPK_BF_FRAG(stage_, tap_, slot_)                                                                        \
    {                                                                                                          \
        const unsigned ta_ = a_base + (unsigned)((tap_) * 2 * CO_T * 16);                                      \
        const unsigned tb_ = (unsigned)((((tap_) / 3) * PW + (tap_) % 3) * 16);                                \
        _Pragma("unroll") for (int hl = 0; hl < 2; ++hl) {                                                     \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                      \
                fa[slot_][hl][m] = *reinterpret_cast<const bf16x8*>((stage_) + ta_ + hl * hl_w + m * 32 * 16); \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                      \
                fb[slot_][hl][n] = *reinterpret_cast<const bf16x8*>((stage_) + b_base[n] + tb_ + hl * hl_x);   \
        }                                                                                                      \
    }
#define SPK_BF_MFMA(slot_)                                                                                      \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                              \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                        \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot_][1][m], fb[slot_][0][n], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot_][0][m], fb[slot_][1][n], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot_][0][m], fb[slot_][0][n], acc[m][n], 0, 0, 0); \
        }
        for (int i = 0; i < n_chunks; ++i) {
            const char* cur = smem + (i & 1) * STAGE;
            SPK_BF_FRAG(cur, 0, 0);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 1 < TAPS) SPK_BF_FRAG(cur, tap + 1, (tap + 1) & 1);
                SPK_BF_MFMA(tap & 1);
            }
            __syncthreads();
        }
#undef SPK_BF_FRAG
#undef SPK_BF_MFMA
    }

    // ---- epilogue (element for element the f32 kernel's): out_scale, demodulation, bias, noise, LeakyReLU * gain, style ----
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_noise = p.flags & SPK_EPI_NOISE;
    const bool f_lrelu = p.flags & SPK_EPI_LRELU, f_style = p.flags & SPK_EPI_STYLE;
    const size_t HW = (size_t)p.H * p.W;
    if (p.staged) {
        // Through LDS (both stages are free now: every wave passed the last chunk's barrier): the 64 x 256 block is written in
        // accumulator order and read back as rows, so a thread finishes FOUR consecutive pixels of a channel per step --
        // 16 vector stores per thread instead of 64 dword stores, the per-channel / per-image operands loaded once per vector.
        // With the dword form the epilogue was HALF of the 256^2 layers' time (profiles/r02_f_bf16x3_layers.txt).
        constexpr int OP = PIX_T + 4, F4 = PIX_T / 4, RPI = NTB / F4;         // 8 rows per pass
        float* const ot = reinterpret_cast<float*>(smem);
        if (!producer) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ot[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * OP + (wave * 2 + n) * 32 + l32] = acc[m][n][r];
        }
        __syncthreads();
        const int f4 = tid_all % F4, row0 = tid_all / F4;      // all eight waves store
        const int pt = 4 * f4;
        const int px = pt & (TW - 1), py = (pt >> p.lgTW) & (TH - 1), tb = pt >> (p.lgTW + p.lgTH);
        const int b = b0 + tb, yy = y0 + py, xx = x0 + px;
        const bool pvv = tb < TB && b < p.B && yy < p.H && xx < p.W;      // W % 4 == 0: the vector is in or out as a whole
        if (!pvv) return;
        const size_t pix = (size_t)yy * p.W + xx;
        const size_t o0 = (size_t)b * p.Cout * HW + pix;
        f32x4 nzv = {0.f, 0.f, 0.f, 0.f};
        if (f_noise) nzv = *reinterpret_cast<const f32x4*>(p.noise + (size_t)b * HW + pix);
        const float* stp = f_style ? p.style + (size_t)b * p.style_stride : nullptr;
#pragma unroll 4
        for (int i = 0; i < CO_T / RPI; ++i) {
            const int cl = row0 + RPI * i, co = co0 + cl;
            if (co >= p.Cout) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(ot + cl * OP + pt);
            const float dm = p.out_scale_bc ? p.out_scale_bc[(size_t)b * p.Cout + co] : 1.f;
            const float bb = f_bias ? p.bias[co] : 0.f;
            const float nwc = f_noise ? p.noise_w[co] : 0.f;
            float s0 = 1.f, s1 = 0.f;
            if (f_style) { s0 = stp[co] + 1.f; s1 = stp[p.Cout + co]; }
            f32x4 pre;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = v[k] * p.out_scale;
                if (p.out_scale_bc) t *= dm;
                t += bb;
                if (f_noise) t += nwc * nzv[k];
                if (f_lrelu) t = (t > 0.f ? t : t * p.slope) * p.act_gain;
                pre[k] = t;
                if (f_style) t = t * s0 + s1;
                v[k] = t;
            }
            if (p.y_pre) *reinterpret_cast<f32x4*>(p.y_pre + o0 + (size_t)co * HW) = pre;
            *reinterpret_cast<f32x4*>(p.y + o0 + (size_t)co * HW) = v;
        }
        return;
    }
    if (producer) return;
    bool pv[2];
    size_t poff[2];
    int pb[2];
    float nz[2];
    const float* st[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int pt = (wave * 2 + n) * 32 + l32;
        const int px = pt & (TW - 1), py = (pt >> p.lgTW) & (TH - 1), tb = pt >> (p.lgTW + p.lgTH);
        const int b = b0 + tb, yy = y0 + py, xx = x0 + px;
        pv[n] = tb < TB && b < p.B && yy < p.H && xx < p.W;
        const size_t pix = (size_t)yy * p.W + xx;
        poff[n] = pv[n] ? (size_t)b * p.Cout * HW + pix : 0;
        pb[n] = pv[n] ? b : 0;
        nz[n] = (f_noise && pv[n]) ? p.noise[(size_t)b * HW + pix] : 0.f;
        st[n] = (f_style && pv[n]) ? p.style + (size_t)b * p.style_stride : nullptr;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co >= p.Cout) continue;
            const float bb = f_bias ? p.bias[co] : 0.f;
            const float nwc = f_noise ? p.noise_w[co] : 0.f;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (!pv[n]) continue;
                float v = acc[m][n][r] * p.out_scale;
                if (p.out_scale_bc) v *= p.out_scale_bc[(size_t)pb[n] * p.Cout + co];
                v += bb;
                if (f_noise) v += nwc * nz[n];
                if (f_lrelu) v = (v > 0.f ? v : v * p.slope) * p.act_gain;
                if (p.y_pre) p.y_pre[poff[n] + (size_t)co * HW] = v;
                if (f_style) v = v * (st[n][co] + 1.f) + st[n][p.Cout + co];
                p.y[poff[n] + (size_t)co * HW] = v;
            }
        }
}

// w[Cout][Cin][3][3] fp32 -> [co tile][chunk][hi/lo][tap][h][64 co][8 ci] bf16 (zero padded).  tf = 1 packs the data-gradient
// operator instead: w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx] (then opCout = Cin rows, opCin = Cout contraction channels)
__global__ __launch_bounds__(256) void pack_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cin,
                                                         int Cout, int opCin, int opCout, int tf, int n_chunks, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    long long t = idx;
    const int j = (int)(t % 8); t /= 8;
    const int col = (int)(t % CO_T); t /= CO_T;
    const int h = (int)(t % 2); t /= 2;
    const int tap = (int)(t % TAPS); t /= TAPS;
    const int hl = (int)(t % 2); t /= 2;
    const int chunk = (int)(t % n_chunks);
    const int cot = (int)(t / n_chunks);
    const int co = cot * CO_T + col, ci = chunk * CI_T + 8 * h + j;
    float v = 0.f;
    if (co < opCout && ci < opCin)
        v = tf ? w[((size_t)ci * Cin + co) * TAPS + (TAPS - 1 - tap)] : w[((size_t)co * Cin + ci) * TAPS + tap];
    const unsigned ph = pack_bf16(v, 0.f) & 0xffffu;
    const float hf = __uint_as_float(ph << 16);
    out[idx] = (unsigned short)(hl == 0 ? ph : (pack_bf16(v - hf, 0.f) & 0xffffu));
}

struct Geo { int TW, TH, TB, NPOS, tiles_x, tiles_y, tiles_b; bool ok; };
static Geo geometry(int B, int H, int W) {
    Geo g;
    g.TW = std::min(32, spk::pow2_ceil(W));
    g.TH = std::min(PIX_T / g.TW, spk::pow2_ceil(H));
    g.TB = PIX_T / (g.TW * g.TH);
    auto npos = [&]() { return g.TB * (g.TH + 2) * (g.TW + 2); };
    while (2 * npos() > MAX_ROUNDS * NT && g.TB > 1) g.TB >>= 1;       // idle pixel groups
    g.NPOS = npos();
    g.ok = 2 * g.NPOS <= MAX_ROUNDS * NT;
    g.tiles_x = spk::ceil_div(W, g.TW);
    g.tiles_y = spk::ceil_div(H, g.TH);
    g.tiles_b = spk::ceil_div(B, g.TB);
    return g;
}

}  // namespace spkbf

using namespace spkbf;

extern "C" {

int64_t spk_conv2d_packed_bytes_bf16x3(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return -1;
    return (int64_t)spk::ceil_div(Cout, CO_T) * spk::ceil_div(Cin, CI_T) * W_BYTES;
}

int spk_conv2d_pack_weights_bf16x3_tf(const float* w, void* w_packed, int Cin, int Cout, int transpose_flip, void* stream) {
    SPK_REQUIRE(w && w_packed && Cin > 0 && Cout > 0 && (transpose_flip == 0 || transpose_flip == 1), "pack_weights_bf16x3: bad arguments");
    const int opCin = transpose_flip ? Cout : Cin, opCout = transpose_flip ? Cin : Cout;
    const int n_chunks = spk::ceil_div(opCin, CI_T);
    const long long total = (long long)spk::ceil_div(opCout, CO_T) * n_chunks * (W_BYTES / 2);
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<unsigned short*>(w_packed), Cin, Cout, opCin, opCout, transpose_flip, n_chunks, total);
    return spk::check_launch("pack_bf16x3_kernel");
}

int spk_conv2d_pack_weights_bf16x3(const float* w, void* w_packed, int Cin, int Cout, void* stream) {
    return spk_conv2d_pack_weights_bf16x3_tf(w, w_packed, Cin, Cout, 0, stream);
}

int spk_conv2d_bf16x3_supported(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    return geometry(B, H, W).ok ? 1 : 0;
}

// entered from spk_conv2d_fwd when desc->flags has SPK_CONV_BF16X3
int spk_conv2d_bf16x3_fwd(const spk_conv2d_desc* d, void* stream) {
    SPK_REQUIRE(d && d->x && d->w_packed && d->y, "conv2d bf16x3: null pointer");
    SPK_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1, "conv2d bf16x3: 3x3 stride-1 kernels only");
    SPK_REQUIRE(d->groups <= 1, "conv2d bf16x3: not grouped");
    const unsigned allowed = SPK_CONV_BF16X3 | SPK_EPI_BIAS | SPK_EPI_NOISE | SPK_EPI_LRELU | SPK_EPI_STYLE | SPK_CONV_UPSAMPLE2X |
                             SPK_CONV_UP_FIR1331 | SPK_CONV_IN_BATCH_SCALE;
    SPK_REQUIRE(!(d->flags & ~allowed) && !d->stats, "conv2d bf16x3: forward-only epilogue flags (bias, noise, lrelu, style, "
                "upsample, batch scale)");
    const bool ups = d->flags & SPK_CONV_UPSAMPLE2X;
    if (ups) SPK_REQUIRE(d->H == 2 * d->Hin && d->W == 2 * d->Win, "conv2d bf16x3: upsampled output must be 2x the input");
    else SPK_REQUIRE(d->H == d->Hin && d->W == d->Win, "conv2d bf16x3: output size must equal the input size");
    SPK_REQUIRE(!(d->flags & SPK_EPI_BIAS) || d->bias, "conv2d bf16x3: SPK_EPI_BIAS without bias");
    SPK_REQUIRE(!(d->flags & SPK_EPI_NOISE) || (d->noise && d->noise_w), "conv2d bf16x3: SPK_EPI_NOISE without noise");
    SPK_REQUIRE(!(d->flags & SPK_EPI_STYLE) || d->style, "conv2d bf16x3: SPK_EPI_STYLE without style");
    SPK_REQUIRE(!(d->flags & SPK_CONV_IN_BATCH_SCALE) || d->in_scale, "conv2d bf16x3: IN_BATCH_SCALE without in_scale");
    const Geo g = geometry(d->B, d->H, d->W);
    SPK_REQUIRE(g.ok, "conv2d bf16x3: tile geometry does not fit %dx%d", d->H, d->W);
    SPK_REQUIRE((size_t)g.TB * d->Cin * d->Hin * d->Win < (1ull << 30), "conv2d bf16x3: image group too large for 32-bit offsets");
    SPK_REQUIRE((reinterpret_cast<uintptr_t>(d->w_packed) & 15) == 0, "conv2d bf16x3: packed weights must be 16-byte aligned");
    Args a;
    a.x = d->x; a.wp = reinterpret_cast<const char*>(d->w_packed); a.bias = d->bias; a.noise_w = d->noise_w; a.noise = d->noise;
    a.style = d->style; a.in_scale = (d->flags & SPK_CONV_IN_BATCH_SCALE) ? d->in_scale : nullptr; a.out_scale_bc = d->out_scale_bc; a.y = d->y; a.y_pre = d->y_pre;
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W; a.Hs = d->Hin; a.Ws = d->Win;
    a.lgTW = spk::ilog2(g.TW); a.lgTH = spk::ilog2(g.TH); a.lgTB = spk::ilog2(g.TB);
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
    a.n_chunks = spk::ceil_div(d->Cin, CI_T);
    a.style_stride = d->style_stride; a.flags = d->flags; a.slope = d->lrelu_slope; a.out_scale = d->out_scale;
    a.act_gain = d->act_gain != 0.f ? d->act_gain : 1.f;
    const auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    a.staged = (d->W % 4 == 0 && g.TW >= 4 && aligned16(d->y) && aligned16(d->y_pre) && aligned16(d->noise)) ? 1 : 0;
    const int snpos = g.TB * ((g.TH >> 1) + 2) * ((g.TW >> 1) + 2);          // UPS: source-tile positions
    SPK_REQUIRE(!ups || 2 * snpos <= 2 * NT, "conv2d bf16x3: source tile too large");
    const size_t lds = std::max(2 * (size_t)(W_BYTES + 4 * g.NPOS * 16) + (ups ? 2 * (size_t)(2 * snpos * 32) : (size_t)0),
                                a.staged ? (size_t)CO_T * (PIX_T + 4) * sizeof(float) : (size_t)0);
    SPK_REQUIRE(lds <= 160 * 1024, "conv2d bf16x3: tile does not fit LDS");
    auto kern = ups ? &conv3x3_bf16x3_kernel<true> : &conv3x3_bf16x3_kernel<false>;
    static bool raised[2] = {false, false};
    if (!raised[ups ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        raised[ups ? 1 : 0] = true;
    }
    const long long gx = (long long)g.tiles_x * g.tiles_y * g.tiles_b;
    SPK_REQUIRE(gx < (1ll << 31), "conv2d bf16x3: grid too large");
    dim3 grid((unsigned)gx, (unsigned)spk::ceil_div(d->Cout, CO_T));
    hipLaunchKernelGGL(kern, grid, dim3(NTB), lds, (hipStream_t)stream, a);
    return spk::check_launch("conv3x3_bf16x3_kernel");
}

}  // extern "C"
