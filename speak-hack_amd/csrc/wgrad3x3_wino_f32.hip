// Weight gradient of a 3x3 stride-1 pad-1 convolution as Winograd F(2x2, 3x3) on the gfx950 f32 MFMA pipe -- the companion of
// conv3x3_wino_f32.hip.  With P = A dY A^T (the 2x2 output-gradient tile spread to 4x4) and V = B^T d B (the tile's 4x4 input patch):
//
//     dU_xi[co][ci] = sum over all tiles  P_xi[co][tile] * V_xi[ci][tile]          xi = 0..15: SIXTEEN independent GEMMs whose
//     dg = G^T dU G                                                                  contraction axis is the TILE axis
//
// -- 16 multiply-adds per tile (4 output pixels) and (co, ci) pair where the direct weight gradient spends 36.  fp32 operands,
// products and accumulation; replaces the weight-gradient half of F.conv2d's backward for styleganv1.py:625,630 (SynthesisBlock conv1 /
// conv2; a x2 layer passes the materialised x2 image) and styleganv1.py:662 (DiscriminatorBlock conv1), first-order and inside the
// R1 double backward (train.py:246-255); and -- groups + folded pass pairs + a BatchNorm-folded input, wgrad_wino_kernel<AFFINE_RELU> --
// for conv2 of the torchvision Bottlenecks of the three encoders (model.py:60-62, :84-90) at 64^2 / 32^2 / 16^2.
//
// One workgroup owns a 64 co x 64 ci block of dU for a contiguous run of CHUNKS of 8 tiles (one row of 8 tiles = 2 x 16 output
// pixels); the MFMA loop is the forward kernel's -- A = P[xi][tile][co], B = V[xi][tile][ci], k = the tile pair, 64 MFMAs per chunk,
// one wave per SIMD with all 256 AGPRs as accumulators, two LDS slots, one barrier per chunk -- but BOTH operands are transformed on
// the fly:
//   * X patches: LDS-DMA (`buffer_load_dwordx4 ... lds`, issued as inline asm, see the forward kernel) of aligned 16-byte pieces,
//     six per patch row (columns -4 .. 19 of the chunk): every wave fetches the 16 channels it transforms itself into a tile of its
//     own, [row 4][piece 6][channel 16][4 floats] -- ordered by the wave's own vmcnt, no cross-wave hazard on the single raw slot;
//     a lane's patch row is then one ds_read_b128 + two ds_read_b32; a piece outside the image gets an out-of-range offset and
//     arrives as zeros (the padding);
//   * P and V rows [xi][tile][64 channels] are ROTATED by 16 channels per tile pair, so that the four tile pairs a wave's lanes
//     write land in four different bank groups (conflict-free transform writes; the fragment reads stay 32 consecutive floats);
//   * dY tiles: two 16-byte loads per lane straight into registers, two chunks ahead;
//   * per chunk and lane: 2 V transforms (32 adds each) + 2 P transforms (14 ops each), interleaved with the MFMAs.
// Epilogue: dg = G^T dU G per (co, ci) in registers, stored into the split's slab [co][tap][ci] -- the layout of the direct weight
// gradient's slabs, so the SAME fixed-order reduce (spk_wgrad_reduce_slabs) finishes: bitwise reproducible, no float atomics.
#include "conv_mfma_f32.hpp"

#include <cstdlib>

namespace spkwgw {

using spkconv::f32x16;
using spkconv::lds_f32_t;
using spkconv::static_for;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4_t;

constexpr int CO_T = 64, CI_T = 64, TC = 8, NT = 256;
constexpr int PV_FLOATS = 16 * TC * 64;                   // one slot of P (or V): [xi 16][tile 8][64]
constexpr int P_OFF = 0, V_OFF = 2 * PV_FLOATS, RAWX_OFF = 4 * PV_FLOATS;
constexpr int RAW_PIECE = 16 * 4, RAW_ROW = 6 * RAW_PIECE, RAW_WAVE = 4 * RAW_ROW;    // per wave: [row 4][piece 6][channel 16][4 floats]
constexpr int RAWX_FLOATS = 4 * RAW_WAVE;                  // 6144
constexpr int LDS_FLOATS = RAWX_OFF + RAWX_FLOATS;
constexpr int LDS_BYTES = LDS_FLOATS * 4;                  // 155 648
constexpr int X_DMA = 6;                                   // 16-byte pieces per lane and chunk

struct Args {
    const float* g;
    const float* x;
    float* slabs;            // [splits][G * Cout][9][Cin]
    const float* in_scale;   // MOD: s[B][Cin]  -- the conv's input was x * s;  AFFINE: scale[Cx] -- it was relu(x * scale + shift)
    const float* in_shift;   // AFFINE: shift[Cx]
    const float* g_scale;    // MOD: d'[B][Cout] -- the gradient that reaches the conv output is g * d'
    int G, gin, Cx, Cy;      // groups: Cin / Cout are per group; x has Cx = gin * (G - 1) + Cin channels, g has Cy = G * Cout
    int B, Cin, Cout, H, W;
    int TXB, TY;             // chunks per tile row (W / 16), tile rows (H / 2)
    int chunks_per_wg;       // even; the last workgroup's run may be shorter (the missing chunks count as zero gradients)
    int n_chunks;            // B * TY * TXB
    unsigned x_bytes;
};

__device__ __forceinline__ float fadd_(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fsub_(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

constexpr int PLAIN = 0, MODULATED = 1, AFFINE_RELU = 2;

template <int MODE>
__global__ __launch_bounds__(NT) void wgrad_wino_kernel(const Args p) {
    constexpr bool MOD = MODE == MODULATED, AFF = MODE == AFFINE_RELU;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int co_tiles = p.Cout / CO_T, grp = blockIdx.y / co_tiles;
    const int co0 = (blockIdx.y - grp * co_tiles) * CO_T, ci0 = blockIdx.z * CI_T;
    const int xc0 = grp * p.gin + ci0, gc0 = grp * p.Cout + co0;      // first channel of the block in x / in g
    const unsigned HW = (unsigned)p.H * (unsigned)p.W;
    const int n = p.chunks_per_wg;
    const int c_first = blockIdx.x * n;
    const int n_valid = min(n, p.n_chunks - c_first);         // (host: >= 1)

    // ---- X pieces: the descriptor's base sits one row and four columns BEFORE the tensor, so that every piece's offset from a
    // chunk's origin is non-negative (the range check sees the lane offset only); pieces outside the image are marked out of range
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) - (p.W + 4), 0, (int)p.x_bytes, 0x00020000);
    unsigned rel[X_DMA], voff[X_DMA];
    unsigned long long e_top[X_DMA], e_bot[X_DMA], e_left[X_DMA], e_right[X_DMA];      // lane masks (SGPR pairs): which lanes' piece k touches an edge
#pragma unroll
    for (int k = 0; k < X_DMA; ++k) {
        const int q = k * 64 + lane;                         // piece of this wave's tile: [r 4][j 6][channel 16]
        const int c = q & 15, rj = q >> 4, r = rj / 6, j = rj - 6 * r;
        rel[k] = ((unsigned)(16 * wave + c) * HW + (unsigned)(r * p.W + 4 * j)) * 4u;
        e_top[k] = __builtin_amdgcn_ballot_w64(r == 0);
        e_bot[k] = __builtin_amdgcn_ballot_w64(r == 3);
        e_left[k] = __builtin_amdgcn_ballot_w64(j == 0);
        e_right[k] = __builtin_amdgcn_ballot_w64(j == 5);
    }
    const unsigned raw_m0 = (unsigned)((RAWX_OFF + wave * RAW_WAVE) * 4);
#define WGW_DMA_X(k_, soff_)                                                                                            \
    {                                                                                                                   \
        const unsigned m0_ = raw_m0 + (k_) * 1024, vo_ = voff[k_], so_ = (soff_);                                       \
        const __amdgpu_buffer_rsrc_t rs_ = rsrc;                                                                        \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0_), "v"(vo_), "s"(rs_), "s"(so_)); \
    }
    // chunk id -> (image, tile row, chunk of the row); X origin (bytes, from the shifted base) and the edges it touches
    auto chunk_of = [&](int id, int& b_, int& ty_, int& tx_) {
        tx_ = id % p.TXB;
        const int t_ = id / p.TXB;
        ty_ = t_ % p.TY;
        b_ = t_ / p.TY;
    };
    auto advance = [&](int& b_, int& ty_, int& tx_, bool go) {          // (uniform: scalar selects)
        const int t1 = tx_ + 1, y1 = ty_ + 1;
        const bool w1 = go && t1 == p.TXB, w2 = w1 && y1 == p.TY;
        tx_ = go ? (w1 ? 0 : t1) : tx_;
        ty_ = w1 ? (w2 ? 0 : y1) : ty_;
        b_ = w2 ? b_ + 1 : b_;
    };
    auto x_origin = [&](int b_, int ty_, int tx_) -> unsigned {
        return (((unsigned)b_ * (unsigned)p.Cx + (unsigned)xc0) * HW + (unsigned)(2 * ty_ * p.W + 16 * tx_)) * 4u;
    };
    auto set_voff = [&](int ty_, int tx_) {
        const unsigned long long a_top = ty_ == 0 ? ~0ull : 0ull, a_bot = 2 * ty_ + 2 >= p.H ? ~0ull : 0ull;
        const unsigned long long a_left = tx_ == 0 ? ~0ull : 0ull, a_right = 16 * tx_ + 16 >= p.W ? ~0ull : 0ull;
        const unsigned oob = 0x80000000u;
#pragma unroll
        for (int k = 0; k < X_DMA; ++k) {                    // the mask is scalar work; one v_cndmask per piece
            const unsigned long long m = (e_top[k] & a_top) | (e_bot[k] & a_bot) | (e_left[k] & a_left) | (e_right[k] & a_right);
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(voff[k]) : "v"(rel[k]), "v"(oob), "s"(m));
        }
    };
    // ---- transform roles: wave w the block's channels 16 w .. 16 w + 15 (of X and of dY); a lane one channel and the tile pair
    // (2 tr_p, 2 tr_p + 1); its dY: the 2 x 4 gradients under the pair
    const int tr_c = 16 * wave + (lane & 15), tr_p = lane >> 4;
    const unsigned g_lane = ((unsigned)(gc0 + tr_c) * HW + 4u * tr_p) * 4u;
    auto g_origin = [&](int b_, int ty_, int tx_) -> size_t {
        return ((size_t)b_ * p.Cy * HW + (size_t)(2 * ty_ * p.W + 16 * tx_)) * 4;
    };
#define WGW_LOAD_G(dst0_, dst1_, sx_, sg_, org_, b_, valid_)                                                            \
    if (valid_) {                                                                                                       \
        const char* gp_ = reinterpret_cast<const char*>(p.g) + (org_);                                                  \
        dst0_ = *reinterpret_cast<const f32x4*>(gp_ + g_lane);                                                          \
        dst1_ = *reinterpret_cast<const f32x4*>(gp_ + g_lane + (size_t)p.W * 4);                                        \
        if constexpr (MOD) {                                                                                            \
            sx_ = p.in_scale[(size_t)(b_) * p.Cin + ci0 + tr_c];   /* (MOD is ungrouped) */                                                        \
            sg_ = p.g_scale[(size_t)(b_) * p.Cout + co0 + tr_c];                                                        \
        }                                                                                                               \
    } else {                                                                                                            \
        dst0_ = f32x4{0.f, 0.f, 0.f, 0.f};                                                                              \
        dst1_ = f32x4{0.f, 0.f, 0.f, 0.f};                                                                              \
    }

    // ---- per-lane LDS bases (floats), made opaque so that the rest folds into instruction offsets ----
    int a_base[4], b_base[4];                                // per tile pair kk (the rotation); + slot * PV + (xi * 8 + 2 kk) * 64
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a_base[kk] = P_OFF + half * 64 + ((wm * 32 + l32 + 16 * kk) & 63);
        b_base[kk] = V_OFF + half * 64 + ((wn * 32 + l32 + 16 * kk) & 63);
        asm volatile("" : "+v"(a_base[kk]), "+v"(b_base[kk]));
    }
    int xr_base = RAWX_OFF + wave * RAW_WAVE + (tr_p * 16 + (lane & 15)) * 4;     // + r * RAW_ROW: piece tr_p (last float), tr_p + 1 (all), tr_p + 2 (first)
    int v_wr = V_OFF + 2 * tr_p * 64 + ((tr_c + 16 * tr_p) & 63);                  // + slot * PV + (xi * 8 + tile of the pair) * 64
    int p_wr = P_OFF + 2 * tr_p * 64 + ((tr_c + 16 * tr_p) & 63);
    int xr_base4 = xr_base >> 2;
    asm volatile("" : "+v"(xr_base), "+v"(xr_base4), "+v"(v_wr), "+v"(p_wr));

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

    // ---- the two transforms of a lane's tile pair (tiles 2 tr_p, 2 tr_p + 1) ----
    float xr[4][6];          // the patch rows: columns 4p - 1 .. 4p + 4
    float tq[16], tv[16];
#define WGW_X_READ(r_)                                                                                                  \
    {                                                                                                                   \
        xr[r_][0] = *((const volatile lds_f32_t*)0 + (xr_base + (r_) * RAW_ROW + 3));                                   \
        const f32x4 m_ = *((const volatile lds_f32x4_t*)0 + (xr_base4 + ((r_) * RAW_ROW + RAW_PIECE) / 4));             \
        xr[r_][1] = m_.x; xr[r_][2] = m_.y; xr[r_][3] = m_.z; xr[r_][4] = m_.w;                                         \
        xr[r_][5] = *((const volatile lds_f32_t*)0 + (xr_base + (r_) * RAW_ROW + 2 * RAW_PIECE));                       \
    }
#define WGW_X_SCALE(r_)                                                                                                 \
    {                                                                                                                   \
        _Pragma("unroll") for (int e_ = 0; e_ < 6; ++e_) xr[r_][e_] *= sx_cur;                                          \
    }
    // AFFINE: the conv's input was relu(x * a + sh) (a BatchNorm folded into its consumer).  The zero padding must stay zero: a
    // position outside the image arrives as x = 0, so only its SHIFT is masked (rows by a scalar condition, the left / right column
    // by a lane mask) -- ten vector ops per chunk for the masks, fma + max per value
    float af_a = 0.f, af_sh = 0.f, sh_row[4], sh_l[4], sh_r[4];
    if constexpr (AFF) {
        af_a = p.in_scale[xc0 + tr_c];
        af_sh = p.in_shift[xc0 + tr_c];
    }
    const unsigned long long m_pair0 = __builtin_amdgcn_ballot_w64(tr_p == 0), m_pair3 = __builtin_amdgcn_ballot_w64(tr_p == 3);
    auto affine_masks = [&](int ty_, int tx_) {
        const unsigned long long ml = tx_ == 0 ? m_pair0 : 0ull, mr = 16 * tx_ + 16 >= p.W ? m_pair3 : 0ull;
        const float zero = 0.f;
        sh_row[1] = sh_row[2] = af_sh;
        sh_row[0] = ty_ == 0 ? 0.f : af_sh;
        sh_row[3] = 2 * ty_ + 2 >= p.H ? 0.f : af_sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(sh_l[r]) : "v"(sh_row[r]), "v"(zero), "s"(ml));
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(sh_r[r]) : "v"(sh_row[r]), "v"(zero), "s"(mr));
        }
    };
#define WGW_X_AFFINE(r_)                                                                                                \
    {                                                                                                                   \
        xr[r_][0] = fmaxf(fmaf(xr[r_][0], af_a, sh_l[r_]), 0.f);                                                        \
        _Pragma("unroll") for (int e_ = 1; e_ < 5; ++e_) xr[r_][e_] = fmaxf(fmaf(xr[r_][e_], af_a, sh_row[r_]), 0.f);   \
        xr[r_][5] = fmaxf(fmaf(xr[r_][5], af_a, sh_r[r_]), 0.f);                                                        \
    }
    // V = B^T d B of tile t_ (0 / 1 of the pair: patch columns 2 t_ .. 2 t_ + 3), written to V slot vs_
#define WGW_V_ROWS(t_, c_)                                                                                              \
    {                                                                                                                   \
        tq[c_] = fsub_(xr[0][2 * (t_) + (c_)], xr[2][2 * (t_) + (c_)]); tq[4 + (c_)] = fadd_(xr[1][2 * (t_) + (c_)], xr[2][2 * (t_) + (c_)]); \
        tq[8 + (c_)] = fsub_(xr[2][2 * (t_) + (c_)], xr[1][2 * (t_) + (c_)]); tq[12 + (c_)] = fsub_(xr[1][2 * (t_) + (c_)], xr[3][2 * (t_) + (c_)]); \
    }
#define WGW_V_COLS(i_)                                                                                                  \
    {                                                                                                                   \
        tv[4 * (i_)] = fsub_(tq[4 * (i_)], tq[4 * (i_) + 2]); tv[4 * (i_) + 1] = fadd_(tq[4 * (i_) + 1], tq[4 * (i_) + 2]); \
        tv[4 * (i_) + 2] = fsub_(tq[4 * (i_) + 2], tq[4 * (i_) + 1]); tv[4 * (i_) + 3] = fsub_(tq[4 * (i_) + 1], tq[4 * (i_) + 3]); \
    }
#define WGW_V_WRITE(vs_, t_, xi_)                                                                                       \
    *((volatile lds_f32_t*)0 + (v_wr + ((vs_) * PV_FLOATS + ((xi_) * TC + (t_)) * 64))) = tv[xi_];
    // P = A dY A^T of tile t_ (gradients y00 y01 / y10 y11), A = [1 0; 1 1; 1 -1; 0 -1] -- stored WITHOUT the minus signs of A's
    // last row (frequencies 3, 7, 11, 12, 13, 14 hold -P): the epilogue puts them back (P_SIGN), the loop saves ten negations a chunk
#define WGW_P_MAKE(t_, y00_, y01_, y10_, y11_)                                                                          \
    {                                                                                                                   \
        const float s0_ = fadd_(y00_, y10_), s1_ = fadd_(y01_, y11_), d0_ = fsub_(y00_, y10_), d1_ = fsub_(y01_, y11_); \
        tq[0] = y00_; tq[1] = fadd_(y00_, y01_); tq[2] = fsub_(y00_, y01_); tq[3] = y01_;                               \
        tq[4] = s0_; tq[5] = fadd_(s0_, s1_); tq[6] = fsub_(s0_, s1_); tq[7] = s1_;                                     \
        tq[8] = d0_; tq[9] = fadd_(d0_, d1_); tq[10] = fsub_(d0_, d1_); tq[11] = d1_;                                   \
        tq[12] = y10_; tq[13] = fadd_(y10_, y11_); tq[14] = fsub_(y10_, y11_); tq[15] = y11_;                           \
    }
#define WGW_P_WRITE(ps_, t_, xi_)                                                                                       \
    *((volatile lds_f32_t*)0 + (p_wr + ((ps_) * PV_FLOATS + ((xi_) * TC + (t_)) * 64))) = tq[xi_];

    // ---- prologue: chunk 0 -> slot 0; chunk 1's X requested, its dY in registers ----
    int b, ty, tx;
    f32x4 ga0, ga1, gb0, gb1;        // dY of the chunk being transformed (a) / of the one after it (b)
    float sxa = 1.f, sga = 1.f, sxb = 1.f, sgb = 1.f, sx_cur = 1.f;      // MOD: the lane's s[b][ci] / d'[b][co] of those chunks
    chunk_of(c_first, b, ty, tx);
    set_voff(ty, tx);
    {
        const unsigned so = x_origin(b, ty, tx);
        static_for<0, X_DMA>([&](auto k) { WGW_DMA_X(decltype(k)::value, so); });
        WGW_LOAD_G(ga0, ga1, sxa, sga, g_origin(b, ty, tx), b, true);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    auto transform_all = [&](auto slot_c, f32x4 g0, f32x4 g1, const float sx, const float sg) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;
        if constexpr (MOD) { sx_cur = sx; g0 *= sg; g1 *= sg; }
        static_for<0, 4>([&](auto r) { WGW_X_READ(decltype(r)::value); });
        if constexpr (MOD) static_for<0, 4>([&](auto r) { WGW_X_SCALE(decltype(r)::value); });
        if constexpr (AFF) {
            affine_masks(ty, tx);
            static_for<0, 4>([&](auto r) { WGW_X_AFFINE(decltype(r)::value); });
        }
        static_for<0, 2>([&](auto t_c) {
            constexpr int t = decltype(t_c)::value;
            static_for<0, 4>([&](auto c) { WGW_V_ROWS(t, decltype(c)::value); });
            static_for<0, 4>([&](auto i) { WGW_V_COLS(decltype(i)::value); });
            static_for<0, 16>([&](auto xi) { WGW_V_WRITE(SL, t, decltype(xi)::value); });
            if constexpr (t == 0) { WGW_P_MAKE(0, g0.x, g0.y, g1.x, g1.y); } else { WGW_P_MAKE(1, g0.z, g0.w, g1.z, g1.w); }
            static_for<0, 16>([&](auto xi) { WGW_P_WRITE(SL, t, decltype(xi)::value); });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    transform_all(std::integral_constant<int, 0>{}, ga0, ga1, sxa, sga);
    {
        advance(b, ty, tx, n_valid > 1);
        const int b1 = b, ty1 = ty, tx1 = tx;
        set_voff(ty1, tx1);
        const unsigned so = x_origin(b1, ty1, tx1);
        __syncthreads();                                     // (P_0 / V_0 published)
        static_for<0, X_DMA>([&](auto k) { WGW_DMA_X(decltype(k)::value, so); });
        WGW_LOAD_G(ga0, ga1, sxa, sga, g_origin(b1, ty1, tx1), b1, n_valid > 1);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);

    // ---- main loop: chunk i out of slot i & 1; during it chunk i + 1 (X in the raw tile, dY in ga) is transformed into the other
    // slot, then chunk i + 2's X is requested into the raw tile and its dY loaded into gb ----
    constexpr int PD = 4;
    float fa[PD + 1], fb[PD + 1];
#define WGW_FRAG(slot_, s_, reg_)                                                                                       \
    {                                                                                                                   \
        constexpr int kk_ = (s_) >> 4, xi_ = (s_) & 15;                                                                 \
        fa[reg_] = *((const volatile lds_f32_t*)0 + (a_base[kk_] + ((slot_) * PV_FLOATS + (xi_ * TC + 2 * kk_) * 64)));      \
        fb[reg_] = *((const volatile lds_f32_t*)0 + (b_base[kk_] + ((slot_) * PV_FLOATS + (xi_ * TC + 2 * kk_) * 64)));      \
    }
    auto chunk_body = [&](auto slot_c, const int i, f32x4& g0, f32x4& g1, float& gsx, float& gsg, f32x4& h0, f32x4& h1, float& hsx,
                          float& hsg) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value, O = 1 - S;
        if constexpr (MOD) { sx_cur = gsx; g0 *= gsg; g1 *= gsg; }
        const int ty1 = ty, tx1 = tx;                // (chunk i + 1: the one in the raw tile)
        // chunk i + 2 (past the run's end: the last chunk's X again, with zero gradients)
        advance(b, ty, tx, i + 2 < n_valid);
        const int b2 = b, ty2 = ty, tx2 = tx;
        const unsigned so2 = x_origin(b2, ty2, tx2);
        const size_t go2 = g_origin(b2, ty2, tx2);
        static_for<0, PD>([&](auto s) { WGW_FRAG(S, decltype(s)::value, decltype(s)::value); });
        static_for<0, 64>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            if constexpr (s + PD < 64) { WGW_FRAG(S, s + PD, (s + PD) % (PD + 1)); }
            acc[s & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s % (PD + 1)], fb[s % (PD + 1)], acc[s & 15], 0, 0, 0);
            // the raw tile holds chunk i + 1: patch rows behind MFMAs 0-3; then tile 0: V 4-11 / writes 12-19, P 20 / writes 21-28;
            // tile 1: V 29-36 / writes 37-44, P 45 / writes 46-53
#ifndef WGW_KO_XFORM
            if constexpr (s < 4) { WGW_X_READ(s); }
            if constexpr (MOD && s >= 2 && s < 6) { WGW_X_SCALE(s - 2); }
            if constexpr (AFF && s == 1) { affine_masks(ty1, tx1); }
            if constexpr (AFF && s >= 2 && s < 6) { WGW_X_AFFINE(s - 2); }
#endif
#ifndef WGW_KO_LOADS
            if constexpr (s == 6) { set_voff(ty2, tx2); }
            // (MFMA 8 waited for fragments requested after the last patch-row read: the rows are in registers, the tile is free)
            if constexpr (s >= 8 && s < 8 + X_DMA) { WGW_DMA_X(s - 8, so2); }
            if constexpr (s == 14) { WGW_LOAD_G(h0, h1, hsx, hsg, go2, b2, i + 2 < n_valid); }
#endif
#ifndef WGW_KO_XFORM
            static_for<0, 2>([&](auto t_c) {
                constexpr int t = decltype(t_c)::value, t0 = (MODE != PLAIN ? 6 : 4) + 25 * t;
                if constexpr (s >= t0 && s < t0 + 4) { WGW_V_ROWS(t, s - t0); }
                if constexpr (s >= t0 + 4 && s < t0 + 8) { WGW_V_COLS(s - t0 - 4); }
                if constexpr (s >= t0 + 8 && s < t0 + 16) { WGW_V_WRITE(O, t, 2 * (s - t0 - 8)); WGW_V_WRITE(O, t, 2 * (s - t0 - 8) + 1); }
                if constexpr (s == t0 + 16) {
                    if constexpr (t == 0) { WGW_P_MAKE(0, g0.x, g0.y, g1.x, g1.y); } else { WGW_P_MAKE(1, g0.z, g0.w, g1.z, g1.w); }
                }
                if constexpr (s >= t0 + 17 && s < t0 + 25) { WGW_P_WRITE(O, t, 2 * (s - t0 - 17)); WGW_P_WRITE(O, t, 2 * (s - t0 - 17) + 1); }
            });
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): this wave's X pieces of chunk i + 2 (and its dY) have landed
        __syncthreads();                             // P / V of chunk i + 1 complete; every wave is done with slot S
    };
    for (int i = 0; i < n; i += 2) {                 // (host: n is even -- one loop body, the accumulators stay in the AGPR file)
        chunk_body(std::integral_constant<int, 0>{}, i, ga0, ga1, sxa, sga, gb0, gb1, sxb, sgb);
        chunk_body(std::integral_constant<int, 1>{}, i + 1, gb0, gb1, sxb, sgb, ga0, ga1, sxa, sga);
    }

    // ---- epilogue: dg = G^T dU G (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]) per (co, ci), into the split's slab [co][tap][ci] ----
    float* slab = p.slabs + ((size_t)blockIdx.x * p.Cy + (size_t)grp * p.Cout) * 9 * p.Cin;
    const int ci = ci0 + wn * 32 + l32;
#define WGW_ACC(xi_, r_) ({ float v_; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v_) : "a"(acc[xi_][r_])); v_; })
#ifdef WGW_KO_EPI
    if (p.n_chunks < 0)
#endif
    static_for<0, 16>([&](auto r_c) {
        constexpr int r = decltype(r_c)::value;
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float t[3][4];                               // (G^T M)[a][j]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sj = j == 3 ? -1.f : 1.f;          // P_SIGN: -1 for frequencies (i, 3), i < 3, and (3, j), j < 3
            const float m0 = sj * WGW_ACC(j, r), m1 = sj * WGW_ACC(4 + j, r), m2 = sj * WGW_ACC(8 + j, r), m3 = -sj * WGW_ACC(12 + j, r);
            const float sm = m1 + m2, df = m1 - m2;
            t[0][j] = m0 + 0.5f * sm;
            t[1][j] = 0.5f * df;
            t[2][j] = m3 + 0.5f * sm;
        }
        float* o = slab + (size_t)co * 9 * p.Cin + ci;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float sm = t[a][1] + t[a][2], df = t[a][1] - t[a][2];
            o[(size_t)(3 * a + 0) * p.Cin] = t[a][0] + 0.5f * sm;
            o[(size_t)(3 * a + 1) * p.Cin] = 0.5f * df;
            o[(size_t)(3 * a + 2) * p.Cin] = t[a][3] + 0.5f * sm;
        }
    });
#undef WGW_ACC
#undef WGW_FRAG
#undef WGW_P_WRITE
#undef WGW_P_MAKE
#undef WGW_V_WRITE
#undef WGW_V_COLS
#undef WGW_V_ROWS
#undef WGW_X_AFFINE
#undef WGW_X_SCALE
#undef WGW_X_READ
#undef WGW_LOAD_G
#undef WGW_DMA_X
}

// chunks per workgroup (even) and the split count that follows: ONE workgroup per CU over the whole grid (one round: 2.99 ms over the
// decoder's layers at B = 8 against 3.19 with two rounds of half-length runs -- half the epilogues, half the slabs to reduce), the
// splits a multiple of 8 when there are that many (workgroups of one split then share an XCD's L2 -- they read the same X / dY chunks)
static int wgs_target() {
    static const int v = [] {
        const char* e = getenv("SPK_WGRAD_WINO_WGS");        // lab override
        return e && atoi(e) > 0 ? atoi(e) : 256;
    }();
    return v;
}

static void pick_splits(long long n_chunks, int blocks, int want, int& splits, int& per_wg) {
    long long s = want > 0 ? want : std::max(1, wgs_target() / blocks);
    if (s > 8) s -= s % 8;
    s = std::max(1ll, std::min(s, (n_chunks + 1) / 2));
    per_wg = (int)(2 * ((n_chunks + 2 * s - 1) / (2 * s)));
    splits = (int)((n_chunks + per_wg - 1) / per_wg);
}

}  // namespace spkwgw

using namespace spkwgw;

extern "C" {

int spk_conv2d_wgrad_wino_supported(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    if (Cin % CI_T || Cout % CO_T || H % 2 || W % 16) return 0;
    if ((long long)B * Cin * H * W * 4 + ((long long)W + 4) * 4 >= (1ll << 31)) return 0;     // 32-bit piece offsets, bit 31 = "out of range"
    if ((long long)Cout * H * W * 4 >= (1ll << 32)) return 0;                                  // a lane's 32-bit offset inside one image of dY
    return 1;
}

int spk_conv2d_wgrad_wino_splits(int splits, int B, int Cin, int Cout, int H, int W) {
    if (!spk_conv2d_wgrad_wino_supported(B, Cin, Cout, H, W)) return -1;
    int s, per;
    pick_splits((long long)B * (H / 2) * (W / 16), (Cin / CI_T) * (Cout / CO_T), splits, s, per);
    return s;
}

int64_t spk_conv2d_wgrad_wino_workspace_bytes(int splits, int B, int Cin, int Cout, int H, int W) {
    const int s = spk_conv2d_wgrad_wino_splits(splits, B, Cin, Cout, H, W);
    if (s < 0) return -1;
    return (int64_t)s * Cout * 9 * Cin * 4;
}

// entered from spk_conv2d_wgrad when desc->flags has SPK_CONV_WINOGRAD
int spk_conv2d_wgrad_wino(const spk_wgrad_desc* d, void* stream) {
    SPK_REQUIRE(d && d->g && d->x && d->dw, "wgrad winograd: null pointer");
    SPK_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1, "wgrad winograd: 3x3 stride-1 convs only");
    SPK_REQUIRE(!(d->flags & ~(SPK_CONV_WINOGRAD | SPK_CONV_IN_BATCH_SCALE | SPK_CONV_IN_AFFINE_RELU)), "wgrad winograd: plain, batch-scaled or "
                "affine + ReLU input (a x2 layer passes the materialised x2 image)");
    const bool mod = d->flags & SPK_CONV_IN_BATCH_SCALE, aff = d->flags & SPK_CONV_IN_AFFINE_RELU;
    const int G = d->groups > 1 ? d->groups : 1, fold = d->fold > 1 ? d->fold : 1;
    SPK_REQUIRE(!(mod && aff) && !(mod && G > 1), "wgrad winograd: IN_BATCH_SCALE is ungrouped and excludes IN_AFFINE_RELU");
    SPK_REQUIRE(!mod || (d->in_scale && d->g_scale), "wgrad winograd: IN_BATCH_SCALE needs in_scale = s[B,Cin] and g_scale = d'[B,Cout]");
    SPK_REQUIRE(!aff || (d->in_scale && d->in_shift), "wgrad winograd: IN_AFFINE_RELU without in_scale / in_shift");
    SPK_REQUIRE(G % fold == 0, "wgrad winograd: fold %d must divide groups %d", fold, G);
    SPK_REQUIRE(d->H == d->Hin && d->W == d->Win, "wgrad winograd: output size must equal the input size");
    const int gin = G > 1 ? d->group_in_stride : 0, Cx = gin * (G - 1) + d->Cin, Cy = G * d->Cout;
    SPK_REQUIRE(gin >= 0 && (G == 1 || gin == 0 || gin >= d->Cin), "wgrad winograd: group_in_stride %d", gin);
    SPK_REQUIRE(spk_conv2d_wgrad_wino_supported(d->B, d->Cin, Cy, d->H, d->W) && d->Cout % CO_T == 0 &&
                    (long long)d->B * Cx * d->H * d->W * 4 + ((long long)d->W + 4) * 4 < (1ll << 31),
                "wgrad winograd: shape not served (Cin, Cout multiples of 64, H even, W a multiple of 16, input below 2 GB)");
    int splits, per_wg;
    pick_splits((long long)d->B * (d->H / 2) * (d->W / 16), (d->Cin / CI_T) * (Cy / CO_T), d->splits, splits, per_wg);
    const int64_t need = (int64_t)splits * Cy * 9 * d->Cin * 4;
    SPK_REQUIRE(d->workspace && d->workspace_bytes >= need, "wgrad winograd: workspace too small (%lld < %lld bytes)", (long long)d->workspace_bytes, (long long)need);
    SPK_REQUIRE((reinterpret_cast<uintptr_t>(d->g) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->x) & 15) == 0, "wgrad winograd: 16-byte aligned tensors");
    Args a;
    a.g = d->g; a.x = d->x; a.slabs = static_cast<float*>(d->workspace);
    a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.g_scale = d->g_scale;
    a.G = G; a.gin = gin; a.Cx = Cx; a.Cy = Cy;
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W;
    a.TXB = d->W / 16; a.TY = d->H / 2;
    a.n_chunks = d->B * a.TY * a.TXB;
    a.chunks_per_wg = per_wg;
    a.x_bytes = (unsigned)((long long)d->B * Cx * d->H * d->W * 4 + ((long long)d->W + 4) * 4);
    static bool raised = false;
    if (!raised) {
        for (const void* f : {reinterpret_cast<const void*>(&wgrad_wino_kernel<PLAIN>), reinterpret_cast<const void*>(&wgrad_wino_kernel<MODULATED>),
                              reinterpret_cast<const void*>(&wgrad_wino_kernel<AFFINE_RELU>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        }
        raised = true;
    }
    dim3 grid((unsigned)splits, (unsigned)(Cy / CO_T), (unsigned)(d->Cin / CI_T));
    if (mod) hipLaunchKernelGGL(wgrad_wino_kernel<MODULATED>, grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
    else if (aff) hipLaunchKernelGGL(wgrad_wino_kernel<AFFINE_RELU>, grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(wgrad_wino_kernel<PLAIN>, grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
    int rc = spk::check_launch("wgrad_wino_kernel");
    if (rc != SPK_OK) return rc;
    return spk_wgrad_reduce_slabs(static_cast<const float*>(d->workspace), d->dw, splits, Cy, d->Cin, 9, d->scale, d->accumulate, fold, stream);
}

}  // extern "C"
