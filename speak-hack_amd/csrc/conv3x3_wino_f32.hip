// 3x3 stride-1 pad-1 convolution as Winograd F(2x2, 3x3) on the gfx950 f32 MFMA pipe -- fp32 operands, fp32 products, fp32
// accumulation, 16 multiplies per 2x2 output tile and (ci, co) pair where the direct form spends 36: 2.25x fewer matrix
// instructions for the same convolution.  This is what the reference's own backend does for these layers (PyTorch hands an
// fp32 3x3 nn.Conv2d -- styleganv1.py:615-616,625,630; the discriminator's conv1 of every block, styleganv1.py:662-672 -- to
// MIOpen, whose fp32 3x3 solvers are Winograd kernels); the arithmetic differs from the direct form only in summation order
// and in the +-1, 1/2 transforms: 1e-6-class rel-L2 per layer, asserted in tests/test_wino_gpu.py (bound of the path: 1e-3).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, per (co, ci);  d = the 4x4 input patch of the tile
//     U[xi][co][ci] = G g G^T     -- weights, transformed at PACK time (spk_conv2d_pack_weights_wino), xi = 4 i + j in 0..15
//     V[xi][ci][t]  = B^T d B     -- input patches, transformed while staging (32 adds per patch)
//     M[xi][co][t]  = sum_ci U[xi][co][ci] V[xi][ci][t]   -- SIXTEEN independent GEMMs: v_mfma_f32_32x32x2_f32, A = U, B = V
//     Y = A^T M A                 -- 24 adds per (co, tile), lane-local: the 16 M[xi] of a (co, tile) sit in the same
//                                    accumulator position of 16 accumulator tiles
//
// Block = 64 co x 64 tiles (16 x 4 tiles = 32 x 8 output pixels), 4 waves (2 x 2: 32 co x 32 tiles each), ci chunks of 8.
// A wave holds 16 accumulator tiles = 256 registers: ONE wave per SIMD, one workgroup per CU, so the wave hides its own
// latencies (the weight-gradient kernels' situation): everything that feeds chunk i + 1 is interleaved, one piece behind one
// MFMA, with the 64 MFMAs of chunk i.
//   * U: LDS-DMA (global_load_lds_dwordx4) of the chunk's contiguous 32 KB packed block, one chunk ahead, 2 slots;
//   * raw input: `buffer_load_dword ... lds` gathers -- a lane fetches any address, the wave's 64 dwords land consecutively in
//     LDS; addresses outside the image are out of range of the buffer descriptor and return 0, which IS the zero padding: no
//     select, no mask, no vector instruction.  Wave w stages the planes of channels w and w + 4 for itself (10 x 34 floats
//     each, two chunks ahead, 2 slots) and transforms them: no other wave reads its raw tile;
//   * V: wave w transforms its two planes for all 64 tiles (a lane = a tile: 8 ds_read_b64 + 32 adds + 16 ds_write_b32 per
//     plane), 2 slots;  one barrier per chunk.
// LDS: 2 x 32 KB (U) + 2 x 32 KB (V) + 2 x 11 KB (raw) = 150 KB.  Vector instructions per chunk of 64 MFMAs: 64 adds.
// Epilogue: output transform and the f32 kernel's epilogue element for element (out_scale, bias, noise, LeakyReLU, [y_pre], style,
// [accumulate]) in registers; a lane stores its tile's 2 x 2 pixels as two 8-byte stores, 16 lanes a whole 128-byte line.
#include "conv_mfma_f32.hpp"

namespace spkwino {

using spkconv::f32x16;
using spkconv::lds_f32_t;
using spkconv::static_for;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x2 lds_f32x2_t;

constexpr int CO_T = 64, CI_T = 8, NTILE = 64, NT = 256;
// region shapes (64 tiles each): WIDE = 32 x 8 output pixels (16 x 4 tiles) -- whole 128-byte output lines; SQUARE = 16 x 16 (8 x 8 tiles)
// for images narrower than 32 pixels.  Raw plane with its halo: 34 x 10 = 340 / 18 x 18 = 324 floats, both within the 352-float pitch.
constexpr int WIDE = 0, SQUARE = 1;
constexpr int region_w(int shape) { return shape == SQUARE ? 16 : 32; }
constexpr int region_h(int shape) { return shape == SQUARE ? 16 : 8; }
constexpr int RAW_PLANE = 352, RAW_WAVE = 2 * RAW_PLANE;                    // 704 = 11 x 64: a wave's two planes = 11 gathers
constexpr int RAW_GATHERS = RAW_WAVE / 64;
constexpr int U_FLOATS = 16 * CI_T * CO_T, V_FLOATS = 16 * CI_T * NTILE, RAW_FLOATS = 4 * RAW_WAVE;
constexpr int U_OFF = 0, V_OFF = 2 * U_FLOATS, RAW_OFF = V_OFF + 2 * V_FLOATS;
constexpr int PRM_OFF = RAW_OFF + 2 * RAW_FLOATS;                            // [64 co][bias, noise weight, style s0 + 1, style s1] of the region
constexpr int DEM_OFF = PRM_OFF + 4 * CO_T;                                  // [64 co] demodulation d[b, co] of the region's image (modulated convs)
constexpr int RGBW_OFF = DEM_OFF + CO_T;                                     // [3][64 co] weights of a fused toRGB (SPK_EPI_TORGB)
constexpr int LDS_FLOATS = RGBW_OFF + 3 * CO_T;
constexpr int LDS_BYTES = LDS_FLOATS * 4;                                    // 155 648
constexpr int RGB_PART_OFF = V_OFF + V_FLOATS;                               // toRGB partial sums [4 channel groups][64 tiles][3][4 px]: V slot 1, idle during an epilogue
static_assert(RAW_GATHERS == 11 && 34 * 10 <= RAW_PLANE && 18 * 18 <= RAW_PLANE, "geometry");
constexpr int U_DMA = U_FLOATS * 4 / 1024 / 4;                               // 1 KB blocks per wave and chunk (8)

struct Args {
    const float* x;
    const float* wp;
    const float* bias;
    const float* noise_w;
    const float* noise;
    const float* style;
    float* y;
    float* y_pre;
    const float* out_scale_dev;
    const float* in_scale;       // MOD: [B, Cin] modulation s[b, ci] (SPK_CONV_IN_BATCH_SCALE): the conv runs on x * s
    const float* out_scale_bc;   // MOD: [B, Cout] demodulation d[b, co] applied right after the contraction, or null
    const float* rgb_w;          // RGB: [3][Cout] weights / [3] bias of the 1x1 conv fused behind the epilogue, its output [B,3,H,W]
    const float* rgb_bias;
    float* rgb_y;
    int B, Cin, Cout, H, W;
    int G, gin, Cx, Cy, co_tiles_g;   // groups (the encoders' passes side by side): Cin / Cout per group; x has Cx = gin * (G - 1) + Cin channels,
                                 // y has Cy = G * Cout; blockIdx.y = group * co_tiles_g + channel tile; the groups' images one after another
    int regions_x, regions_y;
    int n_chunks;                // of the whole contraction (the packed image's chunk count)
    int ksplit, cps;             // the contraction in `ksplit` slices of `cps` chunks (even), a work item = (region, slice)
    size_t slice_floats;         // ksplit > 1: slice z writes its partial sums at y + z * slice_floats (the caller's split-K workspace)
    int style_stride;
    unsigned flags;
    unsigned x_bytes;
    float slope, out_scale, act_gain;
};

// ---- weights: U = G g G^T, image [co tile 64][chunk 8][xi 16][ci 8][co 64] (zero padded), one 32 KB block per (co tile, chunk) ----
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int opCin, int opCout,
                                 int transpose_flip, int n_chunks, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co_l = (int)(idx % CO_T);
    long long r = idx / CO_T;
    const int ci_l = (int)(r % CI_T);
    r /= CI_T;
    const int xi = (int)(r % 16);
    r /= 16;
    const int chunk = (int)(r % n_chunks), co_tile = (int)(r / n_chunks);
    const int co = co_tile * CO_T + co_l, ci = chunk * CI_T + ci_l;
    float v = 0.f;
    if (co < opCout && ci < opCin) {
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                // transpose_flip: the data-gradient operator, w'[co][ci][a][b] = w[ci][co][2-a][2-b] (co indexes the forward conv's Cin)
                g[a][b] = transpose_flip ? w[(((size_t)ci * Cin + co) * 3 + (2 - a)) * 3 + (2 - b)] : w[(((size_t)co * Cin + ci) * 3 + a) * 3 + b];
        const int i = xi >> 2, j = xi & 3;
        // rows of G: [1,0,0], [.5,.5,.5], [.5,-.5,.5], [0,0,1]
        float t[3];   // (G g)[i][b]
#pragma unroll
        for (int b = 0; b < 3; ++b)
            t[b] = i == 0 ? g[0][b] : (i == 3 ? g[2][b] : 0.5f * ((g[0][b] + g[2][b]) + (i == 1 ? g[1][b] : -g[1][b])));
        v = j == 0 ? t[0] : (j == 3 ? t[2] : 0.5f * ((t[0] + t[2]) + (j == 1 ? t[1] : -t[1])));
    }
    out[idx] = v;
}

// the same for up to 32 weights in ONE launch (blockIdx.y = item): a training step re-packs both images of every decoder layer after
// each optimizer step -- 20 launches of 26-35 us beside a busy second stream
struct PackWinoList {
    const float* w[SPK_WINO_PACK_MAX];
    float* out[SPK_WINO_PACK_MAX];
    int Cin[SPK_WINO_PACK_MAX], Cout[SPK_WINO_PACK_MAX], tf[SPK_WINO_PACK_MAX];
};

__global__ void pack_wino_list_kernel(const PackWinoList a) {
    const int it = blockIdx.y;
    const float* __restrict__ w = a.w[it];
    float* __restrict__ out = a.out[it];
    const int Cin = a.Cin[it], Cout = a.Cout[it], transpose_flip = a.tf[it];
    const int opCin = transpose_flip ? Cout : Cin, opCout = transpose_flip ? Cin : Cout;
    const int n_chunks = (opCin + CI_T - 1) / CI_T;
    const long long total = (long long)((opCout + CO_T - 1) / CO_T) * n_chunks * U_FLOATS;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int co_l = (int)(idx % CO_T);
        long long r = idx / CO_T;
        const int ci_l = (int)(r % CI_T);
        r /= CI_T;
        const int xi = (int)(r % 16);
        r /= 16;
        const int chunk = (int)(r % n_chunks), co_tile = (int)(r / n_chunks);
        const int co = co_tile * CO_T + co_l, ci = chunk * CI_T + ci_l;
        float v = 0.f;
        if (co < opCout && ci < opCin) {            // (the arithmetic of pack_wino_kernel, expression for expression)
            float g[3][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    g[p][q] = transpose_flip ? w[(((size_t)ci * Cin + co) * 3 + (2 - p)) * 3 + (2 - q)] : w[(((size_t)co * Cin + ci) * 3 + p) * 3 + q];
            const int i = xi >> 2, j = xi & 3;
            float t[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                t[q] = i == 0 ? g[0][q] : (i == 3 ? g[2][q] : 0.5f * ((g[0][q] + g[2][q]) + (i == 1 ? g[1][q] : -g[1][q])));
            v = j == 0 ? t[0] : (j == 3 ? t[2] : 0.5f * ((t[0] + t[2]) + (j == 1 ? t[1] : -t[1])));
        }
        out[idx] = v;
    }
}

// One vector add / subtract, spelled out: given the whole transform, the compiler pairs the operands for v_pk_add_f32 and pays for
// it with a v_mov_b32 per pair (24 moves per chunk of 64 MFMAs; a packed op costs its two scalar ones on this part anyway).
__device__ __forceinline__ float fadd_(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fsub_(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence: it waits for vmcnt(0) too, i.e. for every global
// STORE the wave has issued -- behind the epilogue's stores that is a full HBM write latency (~5 000 cycles, measured with
// tools/lab_wino_phases.py) with the matrix pipe idle, twice per region.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// MOD: the modulated convolution of the StyleGAN2 variant.  B^T (d s) B = s B^T d B: the modulation of the
// plane's channel -- one scalar per plane, the region lies in one image -- multiplies the 16 transformed values (16 more vector ops
// per plane); the demodulation rides on the epilogue's out_scale through a second LDS table.
// RGB (SPK_EPI_TORGB): the layer's 64 output channels sit in ONE workgroup (Cout <= 64), so the 1x1 conv to 3 channels that follows the
// last block (styleganv1.py:607) is a reduction over the epilogue's own registers: 12 fused multiply-adds per channel row, the four
// channel groups (2 waves x 2 half-waves) meet through LDS.  The 134 MB activation is then neither re-read nor -- y == NULL -- written.
template <bool MOD, int SHAPE, bool RGB = false>
__global__ __launch_bounds__(NT) void wino_kernel(const Args p) {
    constexpr int RW = region_w(SHAPE), RH = region_h(SHAPE), TXN = RW / 2, RAW_W = RW + 2, RAW_H = RH + 2, RAW_USED = RAW_W * RAW_H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    // PERSISTENT workgroups (one per CU: the kernel holds 512 registers per lane): workgroup w walks regions of its XCD's
    // contiguous share of the region list (the hardware deals workgroup w to XCD w % 8), neighbours in the list running
    // side by side on the same L2 -- and, the point of it, the first loads of region r + 1 are requested BEFORE the epilogue
    // of region r, so that a region's global-load latency hides behind its predecessor's output transform and stores.
    const int n_regions = p.regions_x * p.regions_y * p.B * p.ksplit;      // work items: (region, contraction slice), slice fastest
    const int xcd = (int)blockIdx.x & 7, wk = (int)blockIdx.x >> 3, wpx = (int)gridDim.x >> 3;      // (host: gridDim.x % 8 == 0)
    const int share_q = n_regions >> 3, share_r = n_regions & 7;
    const int reg_begin = xcd * share_q + min(xcd, share_r), reg_end = reg_begin + share_q + (xcd < share_r ? 1 : 0);
    const int grp = (int)blockIdx.y / p.co_tiles_g, co_tile = (int)blockIdx.y - grp * p.co_tiles_g, co0 = co_tile * CO_T;     // co0: within the group
    const size_t HW = (size_t)p.H * p.W;

    // ---- the raw gathers: byte offsets of this lane's 11 elements (plane j of channel wave + 4 j), or out of range ----
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    unsigned voff[RAW_GATHERS], voff_next[RAW_GATHERS];   // the gathers of the region being requested / of the region after it
    auto region_coords = [&](int item, int& b_, int& y0_, int& x0_, int& kb_) {
        const int reg = item / p.ksplit;
        kb_ = (item - reg * p.ksplit) * p.cps;               // first chunk of the item's slice
        const int rx_ = reg % p.regions_x;
        const int t_ = reg / p.regions_x;
        b_ = t_ / p.regions_y;
        y0_ = (t_ - b_ * p.regions_y) * RH;
        x0_ = rx_ * RW;
    };
    // per lane and gather, once: the element's byte offset from the region's origin pixel in plane 0 of the image (wraps for the
    // halo's row -1 / column -1: the sum with the origin is in range whenever the element exists), and which image edges would
    // put it outside (bit 0: region at the top edge, 1: bottom, 2: left, 3: right, 4: the pitch's padding -- never loaded)
    unsigned rel[RAW_GATHERS], edge[RAW_GATHERS];
#pragma unroll
    for (int k = 0; k < RAW_GATHERS; ++k) {
        const int e = k * 64 + lane;
        const int j = e >= RAW_PLANE ? 1 : 0, q = e - j * RAW_PLANE;
        const int r = q / RAW_W, c = q - r * RAW_W;
        rel[k] = ((unsigned)(wave + 4 * j) * (unsigned)HW + (unsigned)((r - 1) * p.W + (c - 1))) * 4u;
        edge[k] = (r == 0 ? 1u : 0u) | (r == RAW_H - 1 ? 2u : 0u) | (c == 0 ? 4u : 0u) | (c == RAW_W - 1 ? 8u : 0u) | (q >= RAW_USED ? 16u : 0u);
    }
    auto set_voff = [&](unsigned (&vo)[RAW_GATHERS], int b_, int y0_, int x0_) {
        // (host: the whole tensor is below 2^29 floats)
        const unsigned origin = (((unsigned)b_ * (unsigned)p.Cx + (unsigned)(grp * p.gin)) * (unsigned)HW + (unsigned)(y0_ * p.W + x0_)) * 4u;
        const unsigned at = (y0_ == 0 ? 1u : 0u) | (y0_ + RH >= p.H ? 2u : 0u) | (x0_ == 0 ? 4u : 0u) | (x0_ + RW >= p.W ? 8u : 0u) | 16u;
#pragma unroll
        for (int k = 0; k < RAW_GATHERS; ++k) vo[k] = (edge[k] & at) ? 0x80000000u : origin + rel[k];
    };
    const unsigned chunk_bytes = (unsigned)(CI_T * HW * 4);
    const float* wsrc = p.wp + (size_t)blockIdx.y * p.n_chunks * U_FLOATS;

    // LDS-DMA as inline assembly: the compiler treats a pending `... lds` load as a flat access that may complete out of order
    // with the ds_reads and answers with s_waitcnt lgkmcnt(0) in front of EVERY fragment use until the next vmcnt(0) -- with one
    // wave per SIMD that is an LDS round trip per four MFMAs.  Issued behind its back, the fragment reads keep their counted
    // waits (lgkmcnt(6)); completion is this kernel's business: s_waitcnt vmcnt(0) before the chunk barrier.  (LDS base = 0:
    // the kernel has no static LDS.)
#define WINO_DMA_RAW(vo_in_, chunk_, slot_, k_)                                                                         \
    {   /* (operands through locals: an asm operand inside a lambda does not capture) */                                \
        const unsigned m0_ = (unsigned)((RAW_OFF + (slot_) * RAW_FLOATS + (k_) * 64) * 4) + raw_m0, vo_ = (vo_in_);     \
        const unsigned so_ = (unsigned)(chunk_) * chunk_bytes;                                                          \
        const __amdgpu_buffer_rsrc_t rs_ = rsrc;                                                                        \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(m0_), "v"(vo_), "s"(rs_), "s"(so_)); \
    }
#define WINO_DMA_U(chunk_, slot_, k_)                                                                                   \
    {                                                                                                                   \
        const unsigned m0_ = (unsigned)((U_OFF + (slot_) * U_FLOATS) * 4 + (k_) * 1024) + u_m0, vo_ = lane16;           \
        const float* sb_ = wsrc + (size_t)(chunk_) * U_FLOATS + (wave * U_DMA + (k_)) * 256;                            \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0_), "v"(vo_), "s"(sb_));   \
    }
    const unsigned raw_m0 = (unsigned)(wave * RAW_WAVE * 4), u_m0 = (unsigned)(wave * U_DMA * 1024), lane16 = (unsigned)lane * 16u;

    // ---- per-lane LDS bases (floats); everything else is an instruction immediate ----
    // (made opaque: the compiler then folds only the small compile-time parts into the 16-bit instruction offsets instead of
    // materialising one address register per constant that does not fit)
    int a_base = U_OFF + half * CO_T + wm * 32 + l32;               // + slot * U_FLOATS + (xi * 8 + 2 kk) * 64
    int b_base = V_OFF + half * NTILE + wn * 32 + l32;              // + slot * V_FLOATS + (xi * 8 + 2 kk) * 64
    const int ty_l = lane / TXN, tx_l = lane % TXN;
    int raw_rd = (RAW_OFF + wave * RAW_WAVE + 2 * ty_l * RAW_W + 2 * tx_l) >> 1;   // in float pairs; + slot * RAW_FLOATS + j * RAW_PLANE + r * 34 + {0, 2}
    int v_wr = V_OFF + wave * NTILE + lane;                         // + slot * V_FLOATS + (xi * 8 + 4 j) * 64
    asm volatile("" : "+v"(a_base), "+v"(b_base), "+v"(raw_rd), "+v"(v_wr));

    f32x16 acc[16];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    // the input transform of plane j of raw slot rs into V slot vs, in three kinds of pieces (reads, adds, writes)
    float d[2][16], tq[2][16], tv[2][16];
#define WINO_T_READ(rs_, j_, r_)                                                                                        \
    {                                                                                                                   \
        const f32x2 lo_ = *((const volatile lds_f32x2_t*)0 + (raw_rd + ((rs_) * RAW_FLOATS + (j_) * RAW_PLANE + (r_) * RAW_W) / 2)); \
        const f32x2 hi_ = *((const volatile lds_f32x2_t*)0 + (raw_rd + ((rs_) * RAW_FLOATS + (j_) * RAW_PLANE + (r_) * RAW_W + 2) / 2)); \
        d[j_][4 * (r_) + 0] = lo_.x; d[j_][4 * (r_) + 1] = lo_.y; d[j_][4 * (r_) + 2] = hi_.x; d[j_][4 * (r_) + 3] = hi_.y; \
    }
    // B^T d over the rows (column c_): t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3
#define WINO_T_ROWS(j_, c_)                                                                                             \
    {                                                                                                                   \
        tq[j_][c_] = fsub_(d[j_][c_], d[j_][8 + (c_)]); tq[j_][4 + (c_)] = fadd_(d[j_][4 + (c_)], d[j_][8 + (c_)]);     \
        tq[j_][8 + (c_)] = fsub_(d[j_][8 + (c_)], d[j_][4 + (c_)]); tq[j_][12 + (c_)] = fsub_(d[j_][4 + (c_)], d[j_][12 + (c_)]); \
    }
    // (.) B over the columns (row i_); MOD: times the plane's modulation
#define WINO_T_SCALE(j_, i_, sc_)                                                                                       \
    { tv[j_][4 * (i_)] *= (sc_); tv[j_][4 * (i_) + 1] *= (sc_); tv[j_][4 * (i_) + 2] *= (sc_); tv[j_][4 * (i_) + 3] *= (sc_); }
#define WINO_T_COLS(j_, i_)                                                                                             \
    {                                                                                                                   \
        tv[j_][4 * (i_)] = fsub_(tq[j_][4 * (i_)], tq[j_][4 * (i_) + 2]); tv[j_][4 * (i_) + 1] = fadd_(tq[j_][4 * (i_) + 1], tq[j_][4 * (i_) + 2]); \
        tv[j_][4 * (i_) + 2] = fsub_(tq[j_][4 * (i_) + 2], tq[j_][4 * (i_) + 1]); tv[j_][4 * (i_) + 3] = fsub_(tq[j_][4 * (i_) + 1], tq[j_][4 * (i_) + 3]); \
    }
#define WINO_T_WRITE(vs_, j_, xi_)                                                                                      \
    *((volatile lds_f32_t*)0 + (v_wr + ((vs_) * V_FLOATS + ((xi_) * CI_T + 4 * (j_)) * NTILE))) = tv[j_][xi_];

    if constexpr (RGB) {             // (published by the first chunk barrier, long before the first epilogue)
        if (tid < 3 * CO_T) smem[RGBW_OFF + tid] = (tid & (CO_T - 1)) < p.Cout ? p.rgb_w[(tid >> 6) * p.Cout + (tid & (CO_T - 1))] : 0.f;
    }
    const int n = p.cps;
    int reg = reg_begin + wk;
    if (reg >= reg_end) return;
    int b, y0, x0, kb, b_cur, kb_cur;   // b, kb: image / first chunk of the item whose operands are being requested; *_cur: of the item being computed
    region_coords(reg, b, y0, x0, kb);
    b_cur = b;
    kb_cur = kb;
    set_voff(voff, b, y0, x0);
    // ---- first region: U_0, raw_0, raw_1 requested; raw_0 -> V_0 ----
    static_for<0, U_DMA>([&](auto k) { WINO_DMA_U(kb, 0, decltype(k)::value); });
    static_for<0, RAW_GATHERS>([&](auto k) { WINO_DMA_RAW(voff[decltype(k)::value], kb, 0, decltype(k)::value); });
    static_for<0, RAW_GATHERS>([&](auto k) { WINO_DMA_RAW(voff[decltype(k)::value], kb + 1, 1, decltype(k)::value); });
    __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): this wave's DMA has landed (its raw planes are its own)
    static_for<0, 2>([&](auto j) {
        static_for<0, 4>([&](auto r) { WINO_T_READ(0, decltype(j)::value, decltype(r)::value); });
        static_for<0, 4>([&](auto c) { WINO_T_ROWS(decltype(j)::value, decltype(c)::value); });
        static_for<0, 4>([&](auto i) { WINO_T_COLS(decltype(j)::value, decltype(i)::value); });
        if constexpr (MOD) {
            const float sc0_ = p.in_scale[(size_t)b * p.Cin + kb * CI_T + wave + 4 * decltype(j)::value];
            static_for<0, 4>([&](auto i) { WINO_T_SCALE(decltype(j)::value, decltype(i)::value, sc0_); });
        }
        static_for<0, 16>([&](auto xi) { WINO_T_WRITE(0, decltype(j)::value, decltype(xi)::value); });
        __builtin_amdgcn_sched_barrier(0);           // one plane at a time: the register peak of this block decides what is spilled kernel-wide
    });
    __syncthreads();

    // ---- main loop.  The chunks of ALL the workgroup's regions form one stream: chunk i of a region runs out of U / V slot
    // i & 1 (n is even: every region starts in slot 0); during it the weights of the stream's next chunk and the raw planes of
    // the one after that are requested, and the raw planes of the next chunk become its V -- across a region boundary these
    // belong to the NEXT region, so that a region's first operands are in LDS when its predecessor's epilogue ends ----
#ifndef SPK_WINO_PD
#define SPK_WINO_PD 4
#endif
    constexpr int PD = SPK_WINO_PD;                  // fragment prefetch distance in MFMAs (lab: -DSPK_WINO_PD=n)
    float fa[PD + 1], fb[PD + 1];
#define WINO_FRAG(slot_, s_, reg_)                                                                                      \
    {                                                                                                                   \
        constexpr int kk_ = (s_) >> 4, xi_ = (s_) & 15;                                                                 \
        fa[reg_] = *((const volatile lds_f32_t*)0 + (a_base + ((slot_) * U_FLOATS + (xi_ * CI_T + 2 * kk_) * CO_T)));   \
        fb[reg_] = *((const volatile lds_f32_t*)0 + (b_base + ((slot_) * V_FLOATS + (xi_ * CI_T + 2 * kk_) * NTILE)));  \
    }
    auto chunk_body = [&](auto slot_c, auto first_c, const int i) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value, O = 1 - S;     // this chunk's slot; the other one receives the next chunk
        constexpr bool FIRST = decltype(first_c)::value;          // a region's first chunk: its first 16 MFMAs START the accumulators (C = 0)
        const int c1 = i + 1 < n ? kb_cur + i + 1 : kb;           // the stream's next chunk (weights: the same channel tile; past n: the next item's first)
        const int c2 = i + 2 >= n ? kb + i + 2 - n : kb_cur + i + 2;      // the chunk after it (past n: in the next item -- `voff` is that item's by then)
        float msc[2] = {1.f, 1.f};                                // MOD: the modulation of the two planes being transformed (chunk c1)
        if constexpr (MOD) {
            const float* ms_ = p.in_scale + (size_t)(i + 1 < n ? b_cur : b) * p.Cin + c1 * CI_T + wave;
            msc[0] = ms_[0];
            msc[1] = ms_[4];
        }
        static_for<0, PD>([&](auto s) { WINO_FRAG(S, decltype(s)::value, decltype(s)::value); });
        static_for<0, 64>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            if constexpr (s + PD < 64) { WINO_FRAG(S, s + PD, (s + PD) % (PD + 1)); }
            if constexpr (FIRST && s < 16) acc[s & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s % (PD + 1)], fb[s % (PD + 1)], zero16, 0, 0, 0);
            else acc[s & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s % (PD + 1)], fb[s % (PD + 1)], acc[s & 15], 0, 0, 0);
            // requests: the next chunk's weights into the other U slot, the raw planes of the chunk after it into THIS raw
            // slot (its planes became this chunk's V one chunk ago)
            if constexpr (s < U_DMA) { WINO_DMA_U(c1, O, s); }
            else if constexpr (s < U_DMA + RAW_GATHERS) { WINO_DMA_RAW(voff[s - U_DMA], c2, S, s - U_DMA); }
            // raw slot O -> V slot O.  plane 0: reads behind MFMAs 2-5, adds 8-15, writes 16-23; plane 1: 12-15, 24-31, 32-39
            static_for<0, 2>([&](auto j_c) {
                constexpr int j = decltype(j_c)::value, tr = 2 + 10 * j, ta = 8 + 16 * j;
                if constexpr (s >= tr && s < tr + 4) { WINO_T_READ(O, j, s - tr); }
                if constexpr (s >= ta && s < ta + 4) { WINO_T_ROWS(j, s - ta); }
                if constexpr (s >= ta + 4 && s < ta + 8) { WINO_T_COLS(j, s - ta - 4); if constexpr (MOD) { WINO_T_SCALE(j, s - ta - 4, msc[j]); } }
                if constexpr (s >= ta + 8 && s < ta + 16) { WINO_T_WRITE(O, j, 2 * (s - ta - 8)); WINO_T_WRITE(O, j, 2 * (s - ta - 8) + 1); }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): the requested weights and this wave's raw planes have landed
        __syncthreads();                             // the next chunk's V is complete, every wave is done with slot S
    };
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_noise = p.flags & SPK_EPI_NOISE;
    const bool f_lrelu = p.flags & SPK_EPI_LRELU, f_style = p.flags & SPK_EPI_STYLE, f_accum = p.flags & SPK_EPI_ACCUM;
    const float osc = p.out_scale_dev ? p.out_scale * *p.out_scale_dev : p.out_scale;
    const float slope_ = f_lrelu ? p.slope : 1.f, gain_ = f_lrelu ? p.act_gain : 1.f;    // (no LeakyReLU: slope 1, gain 1)
    // epilogue geometry: lane = tile of the region (l32 + 32 wn) -> its 2 x 2 output pixels; accumulator register r of the lane's
    // half = channel row (r & 3) + 8 (r >> 2) + 4 half of the wave's 32
    const int tile = wn * 32 + l32, oy = 2 * (tile / TXN), ox = 2 * (tile % TXN);
    const int row_lane = wm * 32 + 4 * half;                              // + (r & 3) + 8 (r >> 2): row of the 64-channel tile
    // the operand table of the epilogue: thread t fetches operand t & 3 (bias, noise weight, style scale, style shift) of channel t >> 2
    const int prm_which = tid & 3, prm_c = co0 + (tid >> 2);
    const bool prm_on = prm_c < p.Cout && (prm_which == 0 ? f_bias : (prm_which == 1 ? f_noise : f_style));
    const float* const prm_src = (prm_which == 0 ? p.bias : (prm_which == 1 ? p.noise_w : p.style + (prm_which == 3 ? p.Cout : 0))) + prm_c;
    const float prm_neutral = prm_which == 2 ? 1.f : 0.f;                 // + 0, + 0 * noise, * 1, + 0

    using F_ = std::false_type;
#ifdef SPK_WINO_LAB      // tools/lab_wino_phases.py: cycle stamps of one workgroup's second region, written over the output's first floats
    unsigned long long lab_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int lab_region = 0;
#define LAB_STAMP(k_) { if (lab_region == 1) lab_t[k_] = __builtin_readcyclecounter(); }
#else
#define LAB_STAMP(k_)
#endif
    for (;;) {
        LAB_STAMP(0)
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
        // the region after this one (the last region repeats itself: loads nobody uses)
        const int cur_b = b, cur_y0 = y0, cur_x0 = x0;
        b_cur = b;
        kb_cur = kb;
        // everything the epilogue reads from global memory is requested NOW and lands behind the region's MFMAs: the noise of
        // this thread's four pixels, and (wave 0: one channel per lane) the channel's bias / noise weight / style pair, which
        // reach the other waves through a 1 KB LDS table -- no load latency between the last MFMA and the stores
        const size_t pix = (size_t)(cur_y0 + oy) * p.W + (cur_x0 + ox);          // this lane's tile: pixels (pix, pix + 1), (pix + W, pix + W + 1)
        f32x2 nz0, nz1;
        nz0.x = nz0.y = nz1.x = nz1.y = 0.f;
        if (f_noise) {
            nz0 = *reinterpret_cast<const f32x2*>(p.noise + (size_t)cur_b * HW + pix);
            nz1 = *reinterpret_cast<const f32x2*>(p.noise + (size_t)cur_b * HW + pix + p.W);
        }
        // thread t fetches operand t & 3 of channel t >> 2: bias / noise weight / style scale + 1 / style shift, or its neutral value
        float prm = prm_neutral;
        if (prm_on) {                                    // ONE load per thread, no branch per operand kind
            const float t_ = prm_src[prm_which >= 2 ? (size_t)cur_b * p.style_stride : 0];
            prm = prm_which == 2 ? t_ + 1.f : t_;
        }
        float dem = 1.f;
        if constexpr (MOD) {
            if (p.out_scale_bc && tid < CO_T && co0 + tid < p.Cout) dem = p.out_scale_bc[(size_t)cur_b * p.Cout + co0 + tid];
        }
        const bool more = reg + wpx < reg_end;
        if (more) {
            reg += wpx;
            region_coords(reg, b, y0, x0, kb);
        }
        set_voff(voff_next, b, y0, x0);

        // (host: n is even.)  From the last pair of chunks on, every raw request belongs to the next region: its offsets move in,
        // once, by selects (no branch in the body).
        // ONE loop body, entered once per region: the 256 accumulator registers never cross a control-flow merge other than the
        // loop header and stay in the AGPR file.  (Peeling the first pair of chunks so that its MFMAs start from C = 0 -- no
        // zero fill -- was tried: the compiler then copies all accumulators to VGPRs at the loop exit and spills.)
        LAB_STAMP(1)
        for (int i = 0; i < n; i += 2) {
            const bool last_pair = i + 2 >= n;
#pragma unroll
            for (int k = 0; k < RAW_GATHERS; ++k) voff[k] = last_pair ? voff_next[k] : voff[k];
            chunk_body(std::integral_constant<int, 0>{}, F_{}, i);
            chunk_body(std::integral_constant<int, 1>{}, F_{}, i + 1);
        }

        // ---- epilogue, in registers: output transform Y = A^T M A (A^T = [1,1,1,0; 0,1,-1,-1]) of one channel row at a time, then the
        // f32 kernel's epilogue element for element (out_scale, bias, noise, LeakyReLU, [y_pre], style, [accumulate]) on the lane's
        // 2 x 2 pixels, and two 8-byte stores: lanes 0-15 / 16-31 of a half-wave cover 32 consecutive pixels of two image rows --
        // whole 128-byte lines -- so the block needs no trip through LDS (which would also collide with the next region's operands).
        // Every stage runs unconditionally; an absent one has its neutral operand in the table.
        LAB_STAMP(2)
        smem[PRM_OFF + tid] = prm;
        if constexpr (MOD) { if (tid < CO_T) smem[DEM_OFF + tid] = dem; }
        lds_barrier();
        unsigned hw_ = (unsigned)HW;                 // (opaque per region: the 16 rows' channel offsets must not be hoisted out of the region
        asm volatile("" : "+s"(hw_));                //  loop as 32 live registers)
        const size_t o0 = ((size_t)cur_b * p.Cy + (size_t)grp * p.Cout) * hw_ + pix + (size_t)(kb_cur / p.cps) * p.slice_floats;
        LAB_STAMP(3)
        // The accumulators are read out of the AGPR file one element at a time, by hand (v_accvgpr_read_b32 with an "a" operand):
        // left to the compiler, the first use of element r of a tile copies the whole 16-register tile to VGPRs, a row touches all
        // 16 tiles, and 256 VGPRs of copies push everything else that is alive into scratch -- whose reloads then queue, in
        // vmcnt order, behind the epilogue's own stores (25 000 cycles per region, measured).
#define WINO_ACC(xi_, r_) ({ float v_; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v_) : "a"(acc[xi_][r_])); v_; })
        float rgb[3][4];                                 // RGB: this lane's share (its 16 channel rows) of its tile's 2 x 2 pixels x 3 channels
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) rgb[o][q] = 0.f;
        static_for<0, 16>([&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            const int row = row_lane + (r & 3) + 8 * (r >> 2);
            const float4 q_ = *reinterpret_cast<const float4*>(smem + PRM_OFF + 4 * row);
            float s0[4], s1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m0 = WINO_ACC(4 * i, r), m1 = WINO_ACC(4 * i + 1, r), m2 = WINO_ACC(4 * i + 2, r), m3 = WINO_ACC(4 * i + 3, r);
                s0[i] = (m0 + m1) + m2;
                s1[i] = (m1 - m2) - m3;
            }
            f32x2 top, bot;
            top.x = (s0[0] + s0[1]) + s0[2]; top.y = (s1[0] + s1[1]) + s1[2];
            bot.x = (s0[1] - s0[2]) - s0[3]; bot.y = (s1[1] - s1[2]) - s1[3];
            float osc_r = osc;
            if constexpr (MOD) osc_r = osc * smem[DEM_OFF + row];
            top.x = top.x * osc_r + q_.x; top.y = top.y * osc_r + q_.x; bot.x = bot.x * osc_r + q_.x; bot.y = bot.y * osc_r + q_.x;
            top.x += q_.y * nz0.x; top.y += q_.y * nz0.y; bot.x += q_.y * nz1.x; bot.y += q_.y * nz1.y;
            top.x = (top.x > 0.f ? top.x : top.x * slope_) * gain_; top.y = (top.y > 0.f ? top.y : top.y * slope_) * gain_;
            bot.x = (bot.x > 0.f ? bot.x : bot.x * slope_) * gain_; bot.y = (bot.y > 0.f ? bot.y : bot.y * slope_) * gain_;
            const int co = co0 + row;
            if constexpr (RGB) {                     // (no y_pre / accumulate with a fused toRGB: host)
                top.x = top.x * q_.z + q_.w; top.y = top.y * q_.z + q_.w; bot.x = bot.x * q_.z + q_.w; bot.y = bot.y * q_.z + q_.w;
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float w_ = smem[RGBW_OFF + o * CO_T + row];      // (rows past Cout: 0)
                    rgb[o][0] += w_ * top.x; rgb[o][1] += w_ * top.y; rgb[o][2] += w_ * bot.x; rgb[o][3] += w_ * bot.y;
                }
                if (p.y && co < p.Cout) {
                    const size_t off = o0 + (size_t)co * hw_;
                    *reinterpret_cast<f32x2*>(p.y + off) = top;
                    *reinterpret_cast<f32x2*>(p.y + off + p.W) = bot;
                }
                __builtin_amdgcn_sched_barrier(0);       // one channel row at a time: hoisted table reads of 16 rows would not fit the registers
            } else if (co < p.Cout) {
                const size_t off = o0 + (size_t)co * hw_;
                if (p.y_pre) {
                    *reinterpret_cast<f32x2*>(p.y_pre + off) = top;
                    *reinterpret_cast<f32x2*>(p.y_pre + off + p.W) = bot;
                }
                top.x = top.x * q_.z + q_.w; top.y = top.y * q_.z + q_.w; bot.x = bot.x * q_.z + q_.w; bot.y = bot.y * q_.z + q_.w;
                if (f_accum) {
                    const f32x2 o_t = *reinterpret_cast<const f32x2*>(p.y + off), o_b = *reinterpret_cast<const f32x2*>(p.y + off + p.W);
                    top.x += o_t.x; top.y += o_t.y; bot.x += o_b.x; bot.y += o_b.y;
                }
#ifdef SPK_WINO_NOSTORE   // LAB knock-out: everything but the stores themselves
                if (top.x == 12345.678f)
#endif
                {
                    *reinterpret_cast<f32x2*>(p.y + off) = top;
                    *reinterpret_cast<f32x2*>(p.y + off + p.W) = bot;
                }
            }
        });
#undef WINO_ACC
        if constexpr (RGB) {
            // the four channel groups of a tile: (wm, half) -> LDS [group][tile][3][4], summed in group order by thread (tile, channel)
            float* part = smem + RGB_PART_OFF + ((wm * 2 + half) * NTILE + tile) * 12;
#pragma unroll
            for (int o = 0; o < 3; ++o) *reinterpret_cast<float4*>(part + 4 * o) = make_float4(rgb[o][0], rgb[o][1], rgb[o][2], rgb[o][3]);
            lds_barrier();
            if (tid < 3 * NTILE) {
                const int o = tid >> 6, t = tid & (NTILE - 1);
                const float* ps = smem + RGB_PART_OFF + t * 12 + 4 * o;
                float4 v = *reinterpret_cast<const float4*>(ps);
#pragma unroll
                for (int g = 1; g < 4; ++g) {
                    const float4 u = *reinterpret_cast<const float4*>(ps + g * NTILE * 12);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                const float bo = p.rgb_bias ? p.rgb_bias[o] : 0.f;
                const size_t off = ((size_t)cur_b * 3 + o) * hw_ + (size_t)(cur_y0 + 2 * (t / TXN)) * p.W + (cur_x0 + 2 * (t % TXN));
                f32x2 a_, b_;
                a_.x = v.x + bo; a_.y = v.y + bo; b_.x = v.z + bo; b_.y = v.w + bo;
                *reinterpret_cast<f32x2*>(p.rgb_y + off) = a_;
                *reinterpret_cast<f32x2*>(p.rgb_y + off + p.W) = b_;
            }
        }
        LAB_STAMP(4)
        lds_barrier();                               // (the table has been read: the next region may rewrite it)
#ifdef SPK_WINO_LAB
        if (lab_region == 1 && blockIdx.x == 8 && blockIdx.y == 0 && tid == 0)
            for (int k = 0; k < 5; ++k) p.y[k] = (float)(long long)(lab_t[k] - lab_t[0]);
        ++lab_region;
#endif
        if (!more) break;
    }
#undef WINO_FRAG
#undef WINO_T_WRITE
#undef WINO_T_COLS
#undef WINO_T_SCALE
#undef WINO_T_ROWS
#undef WINO_T_READ
#undef WINO_DMA_U
#undef WINO_DMA_RAW
}

}  // namespace spkwino

using namespace spkwino;

extern "C" {

int64_t spk_conv2d_packed_bytes_wino(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return -1;
    return (int64_t)spk::ceil_div(Cout, CO_T) * spk::ceil_div(Cin, CI_T) * U_FLOATS * 4;
}

int spk_conv2d_pack_weights_wino(const float* w, float* w_packed, int Cin, int Cout, int transpose_flip, void* stream) {
    SPK_REQUIRE(w && w_packed && Cin > 0 && Cout > 0 && (transpose_flip == 0 || transpose_flip == 1), "pack_weights_wino: bad arguments");
    const int opCin = transpose_flip ? Cout : Cin, opCout = transpose_flip ? Cin : Cout;
    const int n_chunks = spk::ceil_div(opCin, CI_T);
    const long long total = (long long)spk::ceil_div(opCout, CO_T) * n_chunks * U_FLOATS;
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, w_packed, Cin, Cout,
                       opCin, opCout, transpose_flip, n_chunks, total);
    return spk::check_launch("pack_wino_kernel");
}

static int wino_shape(int H, int W) {         // -1: no region shape tiles the image
    if (H % region_h(WIDE) == 0 && W % region_w(WIDE) == 0) return WIDE;
    if (H % region_h(SQUARE) == 0 && W % region_w(SQUARE) == 0) return SQUARE;
    return -1;
}

int spk_conv2d_pack_weights_wino_list(const float* const* w, float* const* w_packed, const int* Cin, const int* Cout, const int* transpose_flip,
                                      int n, void* stream) {
    SPK_REQUIRE(w && w_packed && Cin && Cout && transpose_flip && n > 0 && n <= SPK_WINO_PACK_MAX, "pack_weights_wino_list: 1..%d items", SPK_WINO_PACK_MAX);
    PackWinoList a;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        SPK_REQUIRE(w[i] && w_packed[i] && Cin[i] > 0 && Cout[i] > 0 && (transpose_flip[i] == 0 || transpose_flip[i] == 1), "pack_weights_wino_list: item %d", i);
        a.w[i] = w[i]; a.out[i] = w_packed[i]; a.Cin[i] = Cin[i]; a.Cout[i] = Cout[i]; a.tf[i] = transpose_flip[i];
        most = std::max<long long>(most, (transpose_flip[i] ? spk_conv2d_packed_bytes_wino(Cout[i], Cin[i]) : spk_conv2d_packed_bytes_wino(Cin[i], Cout[i])) / 4);
    }
    dim3 grid((unsigned)std::min<long long>((most + 255) / 256, 2048), (unsigned)n);
    hipLaunchKernelGGL(pack_wino_list_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return spk::check_launch("pack_wino_list_kernel");
}

int spk_conv2d_wino_supported(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    if (wino_shape(H, W) < 0) return 0;     // whole 32 x 8 or 16 x 16 pixel regions
    if (Cin % (2 * CI_T)) return 0; // whole chunks (the gathers of a ragged last chunk would reach past the tensor), an even number of them
    if ((long long)B * Cin * H * W * 4 >= (1ll << 31)) return 0;          // 32-bit buffer offsets, bit 31 = "out of range"
    return 1;
}

// Slices of the channel contraction (split-K) for a problem with too few (region, channel tile) pairs to fill the CUs: the smallest
// power of two that brings the grid to >= 3/4 of a workgroup per CU while a slice keeps >= 4 chunks (an even number); `want` > 0: that
// many if the chunk count allows.  1 = no split.  A split launch writes partial sums [ksplit][B][Cout][H][W] into the workspace and
// the direct kernels' split-K finisher applies the epilogue.
int spk_conv2d_wino_ksplit(int want, int B, int Cin, int Cout, int H, int W) {
    if (!spk_conv2d_wino_supported(B, Cin, Cout, H, W)) return -1;
    const int shape = wino_shape(H, W), n_chunks = Cin / CI_T;
    const long long pairs = (long long)B * (H / region_h(shape)) * (W / region_w(shape)) * spk::ceil_div(Cout, CO_T);
    auto ok = [&](int ks) { return ks >= 1 && n_chunks % (2 * ks) == 0 && n_chunks / ks >= 4; };
    if (want > 0) {
        int ks = want;
        while (ks > 1 && !ok(ks)) --ks;
        return ks;
    }
    int ks = 1;
    while (pairs * ks < 192 && ok(2 * ks)) ks *= 2;
    return ks;
}

int64_t spk_conv2d_wino_workspace_bytes(int ksplit, int B, int Cin, int Cout, int H, int W) {
    const int ks = spk_conv2d_wino_ksplit(ksplit, B, Cin, Cout, H, W);
    if (ks < 0) return -1;
    return ks > 1 ? (int64_t)ks * B * Cout * H * W * 4 : 0;
}

// entered from spk_conv2d_fwd when desc->flags has SPK_CONV_WINOGRAD
int spk_conv2d_wino_fwd(const spk_conv2d_desc* d, void* stream) {
    SPK_REQUIRE(d && d->x && d->w_packed, "conv2d winograd: null pointer");
    SPK_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1, "conv2d winograd: 3x3 stride-1 kernels only");
    const int G = d->groups > 1 ? d->groups : 1;
    SPK_REQUIRE(G == 1 || !(d->flags & ~(SPK_CONV_WINOGRAD | SPK_EPI_ACCUM)), "conv2d winograd: a grouped launch takes SPK_EPI_ACCUM only (the encoders' "
                "data gradients)");
    const int gin = G > 1 ? d->group_in_stride : d->Cin, Cx = gin * (G - 1) + d->Cin, Cy = G * d->Cout;
    SPK_REQUIRE(G == 1 || gin >= d->Cin, "conv2d winograd: groups read disjoint channels");
    const unsigned epi = SPK_EPI_BIAS | SPK_EPI_NOISE | SPK_EPI_LRELU | SPK_EPI_STYLE | SPK_EPI_ACCUM;
    const unsigned allowed = SPK_CONV_WINOGRAD | epi | SPK_CONV_IN_BATCH_SCALE | SPK_EPI_TORGB;
    SPK_REQUIRE(!(d->flags & ~allowed) && !d->stats && !d->accum_half,
                "conv2d winograd: plain or batch-scaled input; epilogue flags bias, noise, lrelu, style, accum, torgb");
    const bool mod = d->flags & SPK_CONV_IN_BATCH_SCALE, rgb = d->flags & SPK_EPI_TORGB;
    SPK_REQUIRE(d->y || rgb, "conv2d winograd: null pointer");
    if (rgb)
        SPK_REQUIRE(d->rgb_w && d->rgb_y && d->rgb_channels == 3 && d->Cout <= CO_T && !mod && !d->y_pre && !(d->flags & SPK_EPI_ACCUM) &&
                        (reinterpret_cast<uintptr_t>(d->rgb_y) & 7) == 0,
                    "conv2d winograd: SPK_EPI_TORGB needs rgb_w [3][Cout], rgb_y [B,3,H,W], Cout <= 64 (one channel tile), no y_pre / accumulate / modulation");
    SPK_REQUIRE(!mod || d->in_scale, "conv2d winograd: IN_BATCH_SCALE without in_scale[B,Cin]");
    SPK_REQUIRE(!d->out_scale_bc || mod, "conv2d winograd: out_scale_bc (demodulation) goes with SPK_CONV_IN_BATCH_SCALE");
    SPK_REQUIRE(d->H == d->Hin && d->W == d->Win, "conv2d winograd: output size must equal the input size");
    SPK_REQUIRE(spk_conv2d_wino_supported(d->B, d->Cin, d->Cout, d->H, d->W) && (long long)d->B * Cx * d->H * d->W * 4 < (1ll << 31),
                "conv2d winograd: %dx%d is not a whole number of 32 x 8 or 16 x 16 regions (or the input exceeds 2 GB)", d->H, d->W);
    SPK_REQUIRE(!(d->flags & SPK_EPI_BIAS) || d->bias, "conv2d winograd: SPK_EPI_BIAS without bias");
    SPK_REQUIRE(!(d->flags & SPK_EPI_NOISE) || (d->noise && d->noise_w), "conv2d winograd: SPK_EPI_NOISE without noise");
    SPK_REQUIRE(!(d->flags & SPK_EPI_STYLE) || d->style, "conv2d winograd: SPK_EPI_STYLE without style");
    const auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    SPK_REQUIRE(aligned16(d->w_packed) && aligned16(d->y) && aligned16(d->y_pre) && aligned16(d->noise) && (reinterpret_cast<uintptr_t>(d->x) & 3) == 0,
                "conv2d winograd: tensors must be 16-byte aligned");
    const int shape = wino_shape(d->H, d->W);
    const int ks = spk_conv2d_wino_ksplit(d->ksplit, d->B, d->Cin, Cy, d->H, d->W);       // (the groups' channel tiles count towards the grid)
    const size_t out_floats = (size_t)d->B * Cy * d->H * d->W;
    if (ks > 1)
        SPK_REQUIRE(d->workspace && aligned16(d->workspace) && d->workspace_bytes >= (int64_t)ks * (int64_t)out_floats * 4,
                    "conv2d winograd: %d contraction slices need a workspace of %lld bytes (spk_conv2d_wino_workspace_bytes)", ks,
                    (long long)ks * (long long)out_floats * 4);
    Args a;
    a.x = d->x; a.wp = d->w_packed; a.bias = d->bias; a.noise_w = d->noise_w; a.noise = d->noise; a.style = d->style;
    a.y = d->y; a.y_pre = d->y_pre; a.out_scale_dev = d->out_scale_dev;
    a.in_scale = mod ? d->in_scale : nullptr; a.out_scale_bc = d->out_scale_bc;
    a.rgb_w = d->rgb_w; a.rgb_bias = d->rgb_bias; a.rgb_y = d->rgb_y;
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W;
    a.G = G; a.gin = gin; a.Cx = Cx; a.Cy = Cy; a.co_tiles_g = spk::ceil_div(d->Cout, CO_T);
    a.regions_x = d->W / region_w(shape); a.regions_y = d->H / region_h(shape);
    a.n_chunks = spk::ceil_div(d->Cin, CI_T);
    a.ksplit = ks; a.cps = a.n_chunks / ks; a.slice_floats = 0;
    a.style_stride = d->style_stride; a.flags = d->flags;
    a.x_bytes = (unsigned)((long long)d->B * Cx * d->H * d->W * 4);
    a.slope = d->lrelu_slope; a.out_scale = d->out_scale; a.act_gain = d->act_gain != 0.f ? d->act_gain : 1.f;
    SPK_REQUIRE(!rgb || ks == 1, "conv2d winograd: SPK_EPI_TORGB with a sliced contraction (%d slices): pass ksplit = 1 or run the 1x1 separately", ks);
    if (ks > 1) {       // partial sums, raw: the whole epilogue (and the demodulation) belongs to the finisher
        a.y = static_cast<float*>(d->workspace); a.y_pre = nullptr; a.out_scale_dev = nullptr; a.out_scale_bc = nullptr;
        a.flags = d->flags & ~epi; a.out_scale = 1.f; a.slice_floats = out_floats;
    }
    void (*kern)(const Args) = mod ? (shape == SQUARE ? &wino_kernel<true, SQUARE> : &wino_kernel<true, WIDE>)
                               : rgb ? (shape == SQUARE ? &wino_kernel<false, SQUARE, true> : &wino_kernel<false, WIDE, true>)
                                     : (shape == SQUARE ? &wino_kernel<false, SQUARE> : &wino_kernel<false, WIDE>);
    static bool raised[6] = {false, false, false, false, false, false};
    const int which = (mod ? 2 : rgb ? 4 : 0) + (shape == SQUARE ? 1 : 0);
    if (!raised[which]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        raised[which] = true;
    }
    const long long n_items = (long long)a.regions_x * a.regions_y * d->B * ks;
    SPK_REQUIRE(n_items < (1ll << 31), "conv2d winograd: grid too large");
    // persistent: one workgroup per CU over all channel tiles, a multiple of 8 per channel tile (the XCD shares), at most one per item
    const int co_tiles = G * spk::ceil_div(d->Cout, CO_T);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return spk::fail(SPK_ELAUNCH, "conv2d winograd: no device properties");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    long long gx = std::max(8ll, ((long long)n_cu / co_tiles) / 8 * 8);
    static int lab_mult = -1;       // LAB: SPK_WINO_WGS = workgroups per CU slot (0: one workgroup per item)
    if (lab_mult < 0) { const char* e = getenv("SPK_WINO_WGS"); lab_mult = e ? atoi(e) : 1; }
    gx = lab_mult == 0 ? (n_items + 7) / 8 * 8 : gx * lab_mult;
    gx = std::min(gx, (n_items + 7) / 8 * 8);
    dim3 grid((unsigned)gx, (unsigned)co_tiles);
    hipLaunchKernelGGL(kern, grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
    int rc = spk::check_launch("wino_kernel");
    if (rc != SPK_OK || ks == 1) return rc;
    spkconv::ConvArgs f = {};
    f.bias = d->bias; f.noise_w = d->noise_w; f.noise = d->noise; f.style = d->style; f.out_scale_bc = d->out_scale_bc;
    f.y = d->y; f.y_pre = d->y_pre; f.B = d->B; f.Cin = d->Cin; f.Cout = d->Cout; f.Cy = Cy; f.Cx = Cx; f.G = G; f.H = d->H; f.W = d->W;
    f.style_stride = d->style_stride; f.flags = d->flags & epi; f.slope = d->lrelu_slope; f.out_scale = d->out_scale;
    f.act_gain = a.act_gain; f.out_scale_dev = d->out_scale_dev;
    return spkconv::launch_splitk_epilogue(f, static_cast<const float*>(d->workspace), ks, (hipStream_t)stream);
}

}  // extern "C"
