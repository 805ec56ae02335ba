// 1x1 stride-1 convolution as a GEMM on the gfx950 f32 MFMA pipe, second form (tile configs 14 and 15):
//     y[b, co, pix] = epi( sum_ci W[co, ci] * in(x[b, ci, pix]) ),   in = identity or max(x * scale[ci] + shift[ci], 0)
//
// The measurement this kernel is built on (tools/mfma_valu_coexec.hip, profiles/r03_*): on gfx950 the f32 MFMA runs on the
// SIMD's FP32 vector datapath -- an f32 MFMA stream and ANY vector-ALU instruction of ANOTHER wave of that SIMD do not
// overlap, their times add (4000 x 4 v_mfma_f32_32x32x2_f32 alone 437 us, 64000 v_add_u32 of a second wave alone 169 us,
// together 605 us).  So in an f32-MFMA kernel every vector instruction of every resident wave is paid for in matrix time at
// ~5 cycles each (an MFMA is 64): address arithmetic, selects, the folded BatchNorm, the whole epilogue.  Config 12
// (conv1x1_gemm.hip) and the first version of this file spent ~1 900 vector instructions per wave on a tile of 256 MFMAs --
// which is why their knock-out tables (tools/lab_gemm1x1.py, tools/lab_gemm2.py) showed phases that ADD (k-loop 50 us +
// epilogue 15 + loads 10 = 79 on 128 -> 512 @32^2 against a 46 us matrix time) no matter how many workgroups shared a CU,
// how they were staggered or prioritised.  Hence, here:
//   * NOTHING is staged through registers: the weights ([co tile][k tile][16 k][CO_T], k-major -- the A fragment of a k-step
//     is a contiguous run of rows), the activations (two channel rows of 128 pixels per instruction) and the k-tile's
//     BatchNorm scale / shift all reach LDS by LDS-DMA (global_load_lds_dwordx4) from a wave-uniform base + a per-lane
//     32-bit offset that never changes: no vector instruction per k-tile for addresses, none for zero padding (a ragged
//     last k-tile is moved back to end at Cin and its already-covered channels are zero in the packed weights);
//   * fragment reads and the epilogue's LDS traffic use immediate offsets off three per-k-tile base registers;
//   * the folded BatchNorm + ReLU is two instructions per B-fragment value, and every wave owns ALL rows of its 32 pixels
//     (1 x 4 waves), so no B value is transformed twice;
//   * three-slot ring, ONE bare s_barrier per k-tile, LDS-DMA two k-tiles ahead (s_waitcnt vmcnt(N) counts them);
//   * the epilogue runs whole tiles on a mask-free path, addresses its stores by a scalar base + one lane offset, and
//     skips scale / bias / activation / statistics work that the launch does not ask for.
// Three workgroups per CU (four for the 64-row tile).  ONE-dimensional grid, co tile fastest, each XCD walks a contiguous
// run of work items: the co tiles of a pixel tile run back to back on one L2 and x is fetched from HBM once.
// Epilogue: out_scale -> bias -> lrelu -> accumulate (y itself, or a HALF-RESOLUTION tensor added at the even pixels: the
// data gradient of the block's stride-2 downsample conv, never dilated in memory) -> store -> BatchNorm sums.
//
// replaces: F.conv2d of every stride-1 1x1 conv of the torchvision trunk (conv1 / conv3 / downsample.0, model.py:60-62)
// forward and -- on the transposed weight -- its data gradient.
// hipcc-flags: -fno-slp-vectorize
// (every vector instruction counts here, and the SLP vectoriser's v_pk_* forms come with v_mov shuffles around them)
#include "conv_mfma_f32.hpp"

namespace spkconv {

namespace {

typedef __attribute__((address_space(3))) float lds_f32;          // LDS-typed accesses: volatile ones stay ds_read / ds_write
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

constexpr int KT = 16;                 // channels per k-tile
constexpr int PX_T = 128, BP = PX_T + 4;

struct Gemm2Args {
    const float* x;
    const float* w;          // packed [G][co tile][k tile][KT][CO_T]
    const float* bias;
    const float* in_scale;
    const float* in_shift;
    const float* acc_half;   // [B][Cy][Hh][Wh] or NULL: added at the even pixels (SPK_EPI_ACCUM_HALF)
    double* stats;
    float* y;
    int Cin, Cout, HW, W;    // Cin / Cout per group
    unsigned n_px;           // B * HW
    int G, Cx, Cy, gin, co_tiles_g, px_tiles, n_kt;
    int stats_slots;
    int Wh, HWh;
    unsigned flags;
    float slope, out_scale, act_gain;
    unsigned long long* dbg; // lab builds: per-workgroup time stamps (LAB bit 128), else NULL
};

// LAB (tools/lab_gemm2.py, -DSPK_G2_LAB builds only): knock-outs -- 1 no MFMAs, 2 no epilogue, 4 no x DMA, 8 no weight DMA,
// 16 no fragment reads, 64 no k-loop barrier, 128 time stamps.  0 in the product.
template <int MT, bool AFF, int LAB = 0>
__global__ __launch_bounds__(256, (MT >= 4 ? 3 : 4)) void gemm2_kernel(const Gemm2Args p) {
    constexpr int CO_T = MT * 32;
    // ring slot: [A: KT x CO_T, k-major] [B: KT x 128 pixels] [aux, 1 KB: scale[KT] | shift[KT] of the k-tile's channels, rest unused]
    constexpr int A_FL = KT * CO_T, B_FL = KT * PX_T, AUX_FL = 256, SLOT = A_FL + B_FL + AUX_FL;
    constexpr int NBLK = A_FL / 256;                       // 1 KB LDS-DMA blocks of a k-tile's weights
    static_assert(NBLK % 4 == 0, "whole blocks per wave");
    constexpr int ND = NBLK / 4 + 2 + (AFF ? 1 : 0);       // LDS-DMA instructions per wave and k-tile
    constexpr int PASS_ROWS = 64, PASSES = CO_T / PASS_ROWS;
    static_assert(PASS_ROWS * BP + 2 * CO_T <= 3 * SLOT && 3 * SLOT * 4 < 65536, "the epilogue tile lives in the ring; 16-bit LDS offsets");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [3][A | B | aux]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = the wave's 32-pixel column block
    const int half = lane >> 5, l32 = lane & 31;

    unsigned long long lab_t[4] = {0, 0, 0, 0};
    if constexpr (LAB & 128) lab_t[0] = __builtin_amdgcn_s_memrealtime();
    // work item: every XCD (workgroup i runs on XCD i % 8) walks ONE contiguous run of items; co tile fastest
    unsigned wi;
    {
        const unsigned n = gridDim.x, q = n >> 3, r = n & 7;
        const unsigned xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        wi = xcd * q + min(xcd, r) + k;
    }
    const unsigned rest = wi / (unsigned)p.co_tiles_g;
    const int co_t = (int)(wi - rest * (unsigned)p.co_tiles_g);
    const int grp = (int)(rest / (unsigned)p.px_tiles);
    const int pxt = (int)(rest - (unsigned)grp * (unsigned)p.px_tiles);
    const int co0 = co_t * CO_T, cx0 = grp * p.gin;
    const unsigned P0 = (unsigned)pxt * PX_T;               // (host: pixels, and the byte offsets below, fit 32 bits)
    const unsigned HW = (unsigned)p.HW;

    // ---- LDS-DMA sources: wave-uniform pointer + constant per-lane element offset ----
    // x: one instruction moves two channel rows x 128 pixels (lane -> row lane / 32, pixels 4 (lane % 32) .. + 3); wave w owns rows
    // 4 w .. 4 w + 3 of the k-tile (two instructions).  Lanes past the tensor's pixels read pixel 0 of image 0 (never stored).
    const unsigned Pb = P0 + (unsigned)l32 * 4;
    unsigned x_lane = half * HW;                            // + (b * Cx * HW + pix) for lanes inside the tensor
    if (Pb < p.n_px) {
        const unsigned b = Pb / HW;
        x_lane += b * (unsigned)p.Cx * HW + (Pb - b * HW);
    }
    const float* const x_grp = p.x + (size_t)cx0 * HW;                                            // uniform
    const float* const w_tile = p.w + ((size_t)(grp * p.co_tiles_g + co_t) * p.n_kt) * A_FL;       // uniform
    const unsigned w_lane = (unsigned)lane * 4;
    // aux: lanes 0-3 copy scale[k0 .. k0+15], lanes 4-7 shift[...]; the other lanes fill the rest of the 1 KB block
    const float* aux_lane = ((lane & 4) ? p.in_shift : p.in_scale) + cx0 + (lane & 3) * 4;

    // One k-tile's LDS-DMA: ND instructions per wave, always issued (the vmcnt arithmetic below counts them); past the last
    // k-tile the last one is fetched again (L2 hits into a free slot).  k-tile kt covers channels [k0, k0 + 16) with
    // k0 = min(16 kt, Cin - 16): a ragged last tile moves back (the packed weights are zero for its already-covered part).
    auto dma_tile = [&](int kt, float* slot) {
        const int kte = min(kt, p.n_kt - 1);
        const int k0 = min(kte * KT, p.Cin - KT);
        if constexpr (!(LAB & 8)) {
            const float* src = w_tile + (size_t)kte * A_FL;
#pragma unroll
            for (int j = 0; j < NBLK / 4; ++j)
                __builtin_amdgcn_global_load_lds(src + (j * 4 + wave) * 256 + w_lane, reinterpret_cast<char*>(slot) + (j * 4 + wave) * 1024, 16, 0, 0);
        }
        if constexpr (!(LAB & 4)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* src = x_grp + (size_t)(k0 + wave * 4 + 2 * j) * HW;
                __builtin_amdgcn_global_load_lds(src + x_lane, reinterpret_cast<char*>(slot + A_FL) + (wave * 4 + 2 * j) * (PX_T * 4), 16, 0, 0);
            }
            if constexpr (AFF) __builtin_amdgcn_global_load_lds(aux_lane + k0, reinterpret_cast<char*>(slot + A_FL + B_FL), 16, 0, 0);
        }
    };

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // per-lane LDS byte addresses inside a slot; every fragment read is `base + immediate`
    const unsigned a_lane = (unsigned)(half * CO_T + l32) * 4;
    const unsigned b_lane = (unsigned)(A_FL + half * PX_T + wave * 32 + l32) * 4;
    const unsigned s_lane = (unsigned)(A_FL + B_FL + half) * 4;            // scale[2 ks + half]; shift KT floats further
    float fa[2][MT], fb[2], fs[2][2];
    if constexpr (LAB & 16) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[f][m] = 1.f + lane;
            fb[f] = 2.f + lane;
            fs[f][0] = 1.f; fs[f][1] = 0.f;
        }
    }
    // (volatile: one ds_read_b32 with a 16-bit immediate per value -- the merged ds_read2_b32 forms reach only 1 KB and cost
    // a vector add per pair)
#define SPK_LDS_F(byte_addr_) (*(const volatile lds_f32*)((lds_u8*)smem + (byte_addr_)))
#define SPK_G2_FRAG(ks_, f_)                                                                                 \
    if constexpr (!(LAB & 16)) {                                                                             \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) fa[f_][m] = SPK_LDS_F(va + (2 * (ks_) * CO_T + m * 32) * 4); \
        fb[f_] = SPK_LDS_F(vb + (2 * (ks_) * PX_T) * 4);                                                     \
        if (AFF) { fs[f_][0] = SPK_LDS_F(vs + (2 * (ks_)) * 4); fs[f_][1] = SPK_LDS_F(vs + (KT + 2 * (ks_)) * 4); } \
    }
#define SPK_G2_MFMA(f_)                                                                                      \
    if (AFF) fb[f_] = fmaxf(fb[f_] * fs[f_][0] + fs[f_][1], 0.f);                                            \
    if constexpr (!(LAB & 1)) {                                                                              \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                       \
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[f_][m], fb[f_], acc[m], 0, 0, 0);               \
    } else {                                                                                                 \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) acc[m][0] += fa[f_][m] * fb[f_];                      \
    }

    // ---- the ring.  k-tile i lives in slot i % 3.  Per k-tile ONE barrier, at its start:
    //   s_waitcnt vmcnt(ND)   this wave's LDS-DMA of k-tile i has landed (k-tile i+1's ND instructions may still fly)
    //   s_barrier             every wave's has, and every wave is done reading k-tile i-1 = slot (i+2) % 3
    //   LDS-DMA of k-tile i+2 -> slot (i+2) % 3: two whole k-tiles of flight before it is waited for
    //   8 k-steps of MFMAs out of slot i % 3, fragments one k-step ahead
    float *cur = smem, *nxt = smem + SLOT, *oth = smem + 2 * SLOT;
    const int n_kt = p.n_kt;
    dma_tile(0, cur);
    dma_tile(1, nxt);
    if constexpr (LAB & 128) lab_t[1] = __builtin_amdgcn_s_memrealtime();
    constexpr int KS = KT / 2;
    constexpr int WAIT_ND = 0x0070 | (ND & 0xf);            // s_waitcnt vmcnt(ND) lgkmcnt(0)
    unsigned slot_b = 0;                                    // byte offset of the current slot
    for (int i = 0; i < n_kt; ++i) {
        // (a bare s_barrier: __syncthreads() carries a fence, i.e. vmcnt(0) -- it would wait for the k-tile that was issued
        // one barrier ago; this wave's LDS reads of the previous k-tile are complete -- lgkmcnt(0) -- and nothing else is shared)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((LAB & 12) == 12) __builtin_amdgcn_s_waitcnt(0x0070);
        else if constexpr (LAB & 12) __builtin_amdgcn_s_waitcnt(0x0070 | (LAB & 4 ? NBLK / 4 : ND - NBLK / 4));
        else __builtin_amdgcn_s_waitcnt(WAIT_ND);
        if constexpr (!(LAB & 64)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        dma_tile(i + 2, oth);
        const unsigned va = a_lane + slot_b, vb = b_lane + slot_b, vs = s_lane + slot_b;
        SPK_G2_FRAG(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, KS>([&](auto s_) {
            constexpr int ks = decltype(s_)::value;
            if constexpr (ks + 1 < KS) { SPK_G2_FRAG(ks + 1, (ks + 1) & 1); }
            SPK_G2_MFMA(ks & 1);
            __builtin_amdgcn_sched_barrier(0);              // k-steps stay in order: MT MFMAs on MT accumulators each
        });
        float* t = cur; cur = nxt; nxt = oth; oth = t;
        slot_b = slot_b == 2 * SLOT * 4 ? 0u : slot_b + SLOT * 4;
    }
#undef SPK_G2_FRAG
#undef SPK_G2_MFMA
#undef SPK_LDS_F

    __builtin_amdgcn_s_waitcnt(0x0070);                     // vmcnt(0) lgkmcnt(0): no LDS-DMA may land in the ring from here on
    if constexpr (LAB & 128) {
        float t = 0.f;                                      // (a lab stamp only: the accumulators are forced to be complete first)
        for (int m = 0; m < MT; ++m) t += acc[m][0];
        if (t == 123.456f) p.y[tid] = t;
        lab_t[2] = __builtin_amdgcn_s_memrealtime();
    }
    auto lab_finish = [&]() {
        if constexpr (LAB & 128) {
            if (p.dbg && tid == 0) {
                unsigned long long* o = p.dbg + 6 * (size_t)blockIdx.x;
                o[0] = lab_t[0]; o[1] = lab_t[1]; o[2] = lab_t[2]; o[3] = __builtin_amdgcn_s_memrealtime();
                o[4] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
                o[5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
            }
        }
    };
    if constexpr (LAB & 2) {
        float t = 0.f;
        for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) t += acc[m][r];
        if (t == 123.456f) p.y[tid] = t;
        lab_finish();
        return;
    }

    // ---- epilogue: 64 rows at a time through the ring (every wave brings its 32 pixels of them); a channel row leaves as 512
    // contiguous bytes of 16-byte stores and is handled by one half-wave, so its BatchNorm sums are one 32-lane DPP reduction.
    // Row r of a pass: thread tid / 32 + 8 i, pixels 4 (tid % 32) .. + 3.  Rows come in groups of 8 (host: Cout % 8 == 0), so
    // the ragged last co tile is a shorter loop, not a mask. ----
    float* const red = smem + PASS_ROWS * BP;               // [2][CO_T] row sums / sums of squares
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_lrelu = p.flags & SPK_EPI_LRELU;
    const bool f_accum = p.flags & SPK_EPI_ACCUM, f_stats = p.flags & SPK_EPI_STATS;
    const bool f_half = p.acc_half != nullptr;
    const bool f_affine = f_bias || p.out_scale != 1.f;
    const int rsub = tid >> 5, ocol = (tid & 31) * 4;
    const unsigned Po = P0 + (unsigned)ocol;
    const bool o_ok = Po < p.n_px;
    const bool full = P0 + PX_T <= p.n_px;                  // uniform: no pixel of this tile is outside the tensor
    unsigned o_lane = 0, h_lane = 0;                        // element offsets of (image, pixel) + row rsub in y / in acc_half
    bool h_even = false;
    if (o_ok) {
        const unsigned b = Po / HW, pix = Po - b * HW;
        o_lane = b * (unsigned)p.Cy * HW + pix + (unsigned)rsub * HW;
        if (f_half) {
            const unsigned hh = pix / (unsigned)p.W, ww = pix - hh * (unsigned)p.W;   // W % 4 == 0: the vector's four pixels share the row
            h_even = (hh & 1) == 0;
            h_lane = b * (unsigned)p.Cy * (unsigned)p.HWh + (hh >> 1) * (unsigned)p.Wh + (ww >> 1) + (unsigned)rsub * (unsigned)p.HWh;
        }
    }
    const unsigned w_st = (unsigned)((4 * half) * BP + wave * 32 + l32) * 4;       // staging store: + (row, register) immediates
    const unsigned r_ld = (unsigned)(rsub * BP + ocol) * 4;                         // row-loop load: + 8 i rows
    const unsigned red_b = (unsigned)(PASS_ROWS * BP + rsub) * 4;                   // this half-wave's row sums
    const int slot = pxt % p.stats_slots;
    const bool own_slot = p.stats_slots >= p.px_tiles;
    static_for<0, PASSES>([&](auto h_) {
        constexpr int h = decltype(h_)::value;
        __syncthreads();                                    // every wave is done with the ring / the previous pass
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *(volatile lds_f32*)((lds_u8*)smem + w_st + ((m * 32 + (r & 3) + 8 * (r >> 2)) * BP) * 4) = acc[2 * h + m][r];
        __syncthreads();
        const int row0 = co0 + h * PASS_ROWS;               // first channel of the pass (within the group)
        const int n_it = min(PASS_ROWS / 8, max(0, (p.Cout - row0) / 8));
        float* const y_pass = p.y + (size_t)(grp * p.Cout + row0) * HW;                                 // uniform
        const float* const half_pass = f_half ? p.acc_half + (size_t)(grp * p.Cout + row0) * p.HWh : nullptr;
        const float* const bias_pass = f_bias ? p.bias + grp * p.Cout + row0 : nullptr;
        auto rows = [&](auto masked_) {
            constexpr bool MASKED = decltype(masked_)::value;
            // fully unrolled: iteration i's LDS offsets are immediates, its global addresses a scalar base + the lane's constant offset
            static_for<0, PASS_ROWS / 8>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                if (i >= n_it) return;                       // uniform
                float ssum = 0.f, ssq = 0.f;
                if (!MASKED || o_ok) {
                    const f32x4 t = *(const volatile lds_f32x4*)((lds_u8*)smem + r_ld + i * (8 * BP * 4));
                    float v0 = t[0], v1 = t[1], v2 = t[2], v3 = t[3];
                    if (f_affine) {
                        const float bb = f_bias ? bias_pass[8 * i + rsub] : 0.f;
                        v0 = v0 * p.out_scale + bb; v1 = v1 * p.out_scale + bb; v2 = v2 * p.out_scale + bb; v3 = v3 * p.out_scale + bb;
                    }
                    if (f_lrelu) {
                        v0 = (v0 > 0.f ? v0 : v0 * p.slope) * p.act_gain; v1 = (v1 > 0.f ? v1 : v1 * p.slope) * p.act_gain;
                        v2 = (v2 > 0.f ? v2 : v2 * p.slope) * p.act_gain; v3 = (v3 > 0.f ? v3 : v3 * p.slope) * p.act_gain;
                    }
                    float* const dst = y_pass + (size_t)(i * 8) * HW + o_lane;
                    if (f_accum) {
                        const float4 old = *reinterpret_cast<const float4*>(dst);
                        v0 += old.x; v1 += old.y; v2 += old.z; v3 += old.w;
                    }
                    if (f_half && h_even) {
                        const float2 q = *reinterpret_cast<const float2*>(half_pass + (size_t)(i * 8) * p.HWh + h_lane);
                        v0 += q.x; v2 += q.y;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
                    if (f_stats) {
                        ssum = (v0 + v1) + (v2 + v3);
                        ssq = fmaf(v3, v3, fmaf(v2, v2, fmaf(v1, v1, v0 * v0)));
                    }
                }
                if (f_stats) {                               // uniform flag: every lane takes part
                    ssum = spk::half_wave_sum_hi(ssum);
                    ssq = spk::half_wave_sum_hi(ssq);
                    if (l32 == 31) {
                        *(volatile lds_f32*)((lds_u8*)smem + red_b + (h * PASS_ROWS + 8 * i) * 4) = ssum;
                        *(volatile lds_f32*)((lds_u8*)smem + red_b + (CO_T + h * PASS_ROWS + 8 * i) * 4) = ssq;
                    }
                }
            });
        };
        if (full) rows(std::false_type{});
        else rows(std::true_type{});
    });
    if (f_stats) {
        // the block's row sums leave as two contiguous runs of doubles
        __syncthreads();
        if (tid < 2 * CO_T) {
            const int cl = tid % CO_T, co = co0 + cl;
            if (co < p.Cout) {
                const int cg = grp * p.Cout + co;
                double* dst = p.stats + (size_t)slot * 2 * p.Cy + (tid >= CO_T ? p.Cy : 0) + cg;
                if (own_slot) *dst = (double)red[tid];
                else atomicAdd(dst, (double)red[tid]);
            }
        }
    }
    lab_finish();
}

template <int MT>
int launch(const Gemm2Args& a, bool aff, unsigned grid, hipStream_t stream) {
    constexpr int CO_T = MT * 32;
    constexpr size_t lds = 3 * (size_t)(KT * CO_T + KT * PX_T + 256) * sizeof(float);
    auto kern = aff ? &gemm2_kernel<MT, true> : &gemm2_kernel<MT, false>;
#ifdef SPK_G2_LAB
    if (const char* e = getenv("SPK_G2_LAB")) {
        switch (atoi(e)) {
#define SPK_G2_CASE(n_) case n_: kern = &gemm2_kernel<MT, true, n_>; break;
            SPK_G2_CASE(1) SPK_G2_CASE(2) SPK_G2_CASE(3) SPK_G2_CASE(4) SPK_G2_CASE(8) SPK_G2_CASE(12) SPK_G2_CASE(14) SPK_G2_CASE(16)
            SPK_G2_CASE(128) SPK_G2_CASE(222) SPK_G2_CASE(142) SPK_G2_CASE(30) SPK_G2_CASE(94) SPK_G2_CASE(13) SPK_G2_CASE(64)
#undef SPK_G2_CASE
            default: break;
        }
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return spk::check_launch("gemm2_kernel");
}

}  // namespace

int gemm2_co_tile(int config) { return config == kGemm2Config ? 128 : 64; }

bool gemm2_takes(int kh, int stride, int Cin, int Cout, int H, int W) {
    return kh == 1 && stride == 1 && ((long long)H * W) % 4 == 0 && Cin >= KT && Cin % 4 == 0 && Cout % 8 == 0;
}

long long gemm2_pixel_tiles(int B, int H, int W) { return ((long long)B * H * W + PX_T - 1) / PX_T; }

long long gemm2_packed_floats(int config, int Cin, int Cout) {
    const int co_t = gemm2_co_tile(config);
    return (long long)spk::ceil_div(Cout, co_t) * spk::ceil_div(Cin, KT) * KT * co_t;
}

// w[Cout][Cin] (tf = 0) or its transpose (tf = 1: the operator has Cin output rows and Cout contraction channels) ->
// [co tile][k tile][KT][CO_T], rows past the operator's Cout zero.  k-tile kt holds channels k0 .. k0 + 15 with
// k0 = min(16 kt, Cin - 16): a ragged last tile moves back to end at Cin, and its channels below 16 kt -- the previous
// tile has them -- are zero.
__global__ void pack_gemm2_kernel(const PackList list, float* __restrict__ wp, int Cin, int Cout, int tf, int co_t, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float* __restrict__ w = list.w[blockIdx.y];
    wp += (size_t)blockIdx.y * n;
    const int opCin = tf ? Cout : Cin, opCout = tf ? Cin : Cout;
    const int n_kt = (opCin + KT - 1) / KT;
    long long t = idx;
    const int row = (int)(t % co_t); t /= co_t;
    const int k = (int)(t % KT); t /= KT;
    const int kt = (int)(t % n_kt);
    const int cot = (int)(t / n_kt);
    const int k0 = min(kt * KT, opCin - KT);
    const int co = cot * co_t + row, ci = k0 + k;
    float v = 0.f;
    if (co < opCout && ci >= kt * KT) v = tf ? w[(size_t)ci * Cin + co] : w[(size_t)co * Cin + ci];
    wp[idx] = v;
}

int pack_gemm2(const PackList& list, int n_list, float* w_packed, int Cin, int Cout, int config, int tf, hipStream_t stream) {
    SPK_REQUIRE((tf ? Cout : Cin) >= KT, "pack_weights: configs 14 / 15 need at least %d contraction channels", KT);
    const long long nf = gemm2_packed_floats(config, tf ? Cout : Cin, tf ? Cin : Cout);
    hipLaunchKernelGGL(pack_gemm2_kernel, dim3((unsigned)((nf + 255) / 256), (unsigned)n_list), dim3(256), 0, stream, list, w_packed,
                       Cin, Cout, tf, gemm2_co_tile(config), nf);
    return spk::check_launch("pack_gemm2_kernel");
}

int run_1x1_gemm2(const spk_conv2d_desc* d, hipStream_t stream) {
    SPK_REQUIRE(gemm2_takes(d->kh, d->stride, d->Cin, d->Cout, d->H, d->W),
                "conv2d: configs 14 / 15 (GEMM form) take stride-1 1x1 convs with H*W %% 4 == 0, Cin >= 16, Cin %% 4 == 0, Cout %% 8 == 0");
    const int G = d->groups > 1 ? d->groups : 1;
    const int gin = G > 1 ? d->group_in_stride : d->Cin;
    const long long Cx = (long long)gin * (G - 1) + d->Cin, Cy = (long long)G * d->Cout, HWl = (long long)d->H * d->W;
    SPK_REQUIRE(d->B * Cx * HWl < (1ll << 30) && d->B * Cy * HWl < (1ll << 30), "conv2d: configs 14 / 15 address x and y with 32-bit byte offsets");
    SPK_REQUIRE(!(d->flags & SPK_CONV_IN_AFFINE_RELU) || ((reinterpret_cast<uintptr_t>(d->in_scale) | reinterpret_cast<uintptr_t>(d->in_shift)) & 15) == 0,
                "conv2d: configs 14 / 15 with IN_AFFINE_RELU need 16-byte aligned in_scale / in_shift");
    SPK_REQUIRE(!(d->flags & ~(SPK_EPI_BIAS | SPK_EPI_LRELU | SPK_EPI_ACCUM | SPK_EPI_STATS | SPK_CONV_IN_AFFINE_RELU | SPK_EPI_ACCUM_HALF)) &&
                    !d->out_scale_bc && !d->y_pre,
                "conv2d: configs 14 / 15 (GEMM form) take bias / lrelu / accum / accum-half / stats / in-affine only");
    SPK_REQUIRE(((reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->w_packed) | reinterpret_cast<uintptr_t>(d->y)) & 15) == 0,
                "conv2d: configs 14 / 15 need 16-byte aligned x, y and weights");
    Gemm2Args a;
    a.x = d->x; a.w = d->w_packed; a.bias = d->bias; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.stats = d->stats; a.y = d->y;
    a.Cin = d->Cin; a.Cout = d->Cout; a.HW = d->H * d->W; a.W = d->W; a.n_px = (unsigned)(d->B * HWl);
    a.G = G; a.gin = gin; a.Cx = (int)Cx; a.Cy = (int)Cy;
    const int co_t = gemm2_co_tile(d->config);
    a.co_tiles_g = spk::ceil_div(d->Cout, co_t);
    a.px_tiles = (int)gemm2_pixel_tiles(d->B, d->H, d->W);
    a.n_kt = spk::ceil_div(d->Cin, KT);
    a.stats_slots = d->stats_slots > 1 ? d->stats_slots : 1;
    a.flags = d->flags; a.slope = d->lrelu_slope; a.out_scale = d->out_scale; a.act_gain = d->act_gain != 0.f ? d->act_gain : 1.f;
    a.acc_half = nullptr; a.Wh = a.HWh = 0;
    a.dbg = nullptr;
#ifdef SPK_G2_LAB
    if (const char* e = getenv("SPK_G2_DBG")) a.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 10));
#endif
    if (d->flags & SPK_EPI_ACCUM_HALF) {
        SPK_REQUIRE(d->accum_half && d->W % 4 == 0 && (reinterpret_cast<uintptr_t>(d->accum_half) & 7) == 0,
                    "conv2d: SPK_EPI_ACCUM_HALF needs accum_half (8-byte aligned) and W %% 4 == 0");
        a.acc_half = d->accum_half;
        a.Wh = (d->W - 1) / 2 + 1;
        a.HWh = ((d->H - 1) / 2 + 1) * a.Wh;
    }
    const long long grid = (long long)a.px_tiles * a.co_tiles_g * a.G;
    SPK_REQUIRE(grid < (1ll << 31), "conv2d: grid too large");
    const bool aff = d->flags & SPK_CONV_IN_AFFINE_RELU;
    if (d->config == kGemm2Config) return launch<4>(a, aff, (unsigned)grid, stream);
    return launch<2>(a, aff, (unsigned)grid, stream);
}

}  // namespace spkconv
