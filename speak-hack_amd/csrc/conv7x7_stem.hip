// The ResNet-50 stem: 7x7 stride-2 conv, Cin = 3, Cout = 64 per group (torchvision resnet.py conv1; model.py:60 wraps it), config 16.
// hipcc-flags: -fno-slp-vectorize
//
// On the generic tap kernel the stem pads Cin 3 -> 4 and walks 49 taps of 4-channel chunks: 49.5 TFLOP/s.  Here it is one
// implicit GEMM with K = 147 exactly:
//   * v_mfma_f32_32x32x1_2b_f32 with the PIXELS as the A operand: lanes 0-31 / 32-63 feed two blocks = two output rows of 32
//     pixels; both half-waves use the same k, so an A fragment is `lane base + immediate(ci, ky, kx)` -- with the 32x32x2 form the
//     halves would need different (ky, kx) and the offset would not be an immediate.  B = the weights, k-major [147][64] in LDS;
//   * the 16 x 32 output tile's input patch (37 rows x 69 columns x 3 channels) sits in LDS with its columns split by parity
//     (even columns, then odd columns of a row): pixel ox reads column 2 ox + kx = parity kx & 1, index ox + kx / 2 -- stride 1
//     across the lanes, no bank conflict; zero padding is stored as zeros;
//   * staging: wave w owns patch rows w, w + 4, ... (a row task = channel x row: scalar arithmetic), lane l column l (lanes 0-4 also
//     columns 64-68): a global load is `scalar row base + constant lane offset`, an LDS store `constant lane address + immediate`;
//     the packed weights arrive by LDS-DMA (37 KB, 37 instructions per workgroup);
//   * no vector instruction in the k loop (on gfx950 the f32 MFMA and the vector ALU do not overlap: DESIGN.md 4.7); two
//     workgroups per CU (69 KB of LDS each) cover each other's staging and epilogue.
// Epilogue: plain stores (16-byte: four consecutive pixels of a channel per accumulator group) + SPK_EPI_STATS.
#include "conv_mfma_f32.hpp"

namespace spkconv {

namespace {

constexpr int ST_K = 147, ST_KP = 148;               // contraction (ci, ky, kx); packed rows (row 147 = zeros: whole KB for the DMA)
constexpr int ST_TH = 16, ST_TW = 32;                // output tile
constexpr int ST_PR = 2 * ST_TH + 5;                 // 37 patch rows
constexpr int ST_PC = 2 * ST_TW + 5;                 // 69 patch columns
constexpr int ST_PH = ST_TW + 3;                     // 35 entries per column parity
constexpr int ST_PROW = 2 * ST_PH;                   // floats per patch row (even columns, then odd)
constexpr int ST_W_FL = ST_KP * 64;                  // 9472 floats = 37 KB
constexpr int ST_P_FL = 3 * ST_PR * ST_PROW;         // 7770 floats
constexpr int ST_RED_FL = 4 * 64 * 2;                // per-wave channel sums
constexpr int ST_LDS_BYTES = (ST_W_FL + ST_P_FL + ST_RED_FL) * 4;
constexpr int ST_TASKS = 3 * ST_PR;                  // 111 (channel, row) staging tasks
constexpr int ST_TPW = (ST_TASKS + 3) / 4;           // 28 per wave

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef __attribute__((address_space(3))) float st_lds_f32;
typedef __attribute__((address_space(3))) unsigned char st_lds_u8;

struct StemArgs {
    const float* x; const float* w; float* y; double* stats;
    int B, Cx, gin, Cy, H, W, Hin, Win, tiles_x, tiles_y, stats_slots;
};

template <int I, int N, class F>
__device__ __forceinline__ void st_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        st_static_for<I + 1, N>(f);
    }
}

template <bool STATS>
__global__ __launch_bounds__(256, 2) void stem7x7s2_kernel(const StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int grp = blockIdx.y;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;

    // ---- weights: [148][64] floats of this group by LDS-DMA, 1 KB per instruction ----
    {
        const char* wsrc = reinterpret_cast<const char*>(p.w + (size_t)grp * ST_W_FL) + lane * 16;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int idx = j * 4 + wave;
            if (idx < ST_W_FL / 256)
                __builtin_amdgcn_global_load_lds(wsrc + idx * 1024, reinterpret_cast<char*>(smem) + idx * 1024, 16, 0, 0);
        }
    }
    // ---- input patch: task q = wave + 4 j = (channel, patch row); lane l -> column l, lanes 0-4 also column 64 + l ----
    {
        const float* xb = p.x + ((size_t)b * p.Cx + (size_t)grp * p.gin) * p.Hin * p.Win;
        const int ix_a = 2 * ox0 - 3 + lane, ix_b = ix_a + 64;
        const bool ok_a = (unsigned)ix_a < (unsigned)p.Win, ok_b = lane < ST_PC - 64 && (unsigned)ix_b < (unsigned)p.Win;
        const unsigned off_a = (unsigned)min(max(ix_a, 0), p.Win - 1) * 4u, off_b = (unsigned)min(max(ix_b, 0), p.Win - 1) * 4u;
        // LDS byte address of (task 0, this lane's column); a task adds ST_PROW floats
        const unsigned dst_a = (unsigned)(ST_W_FL + wave * ST_PROW + (lane & 1) * ST_PH + (lane >> 1)) * 4u;
        const unsigned dst_b = (unsigned)(ST_W_FL + wave * ST_PROW + (lane & 1) * ST_PH + 32 + (lane >> 1)) * 4u;
        float va[ST_TPW], vb[ST_TPW];
#pragma unroll
        for (int j = 0; j < ST_TPW; ++j) {
            const int q = min(wave + 4 * j, ST_TASKS - 1);                   // (uniform)
            const int ci = q / ST_PR, r = q - ci * ST_PR;
            const int iy = 2 * oy0 - 3 + r;
            const bool row_ok = (unsigned)iy < (unsigned)p.Hin;
            const char* row = reinterpret_cast<const char*>(xb + ((size_t)ci * p.Hin + min(max(iy, 0), p.Hin - 1)) * p.Win);
            const float a = *reinterpret_cast<const float*>(row + off_a);
            const float c = *reinterpret_cast<const float*>(row + off_b);
            va[j] = (row_ok && ok_a) ? a : 0.f;
            vb[j] = (row_ok && ok_b) ? c : 0.f;
        }
#pragma unroll
        for (int j = 0; j < ST_TPW; ++j) {
            if (wave + 4 * j < ST_TASKS) {                                   // (uniform)
                *(st_lds_f32*)((st_lds_u8*)smem + (dst_a + (unsigned)(j * 4 * ST_PROW * 4))) = va[j];
                if (lane < ST_PC - 64) *(st_lds_f32*)((st_lds_u8*)smem + (dst_b + (unsigned)(j * 4 * ST_PROW * 4))) = vb[j];
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0): this wave's DMA and stores have landed
    __syncthreads();

    // ---- K = 147 steps of 4 MFMAs: two row pairs (rows 4 w + 2 p + half) x two channel tiles ----
    f32x32 acc[2][2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[pr][n][i] = 0.f;
    unsigned a_addr = (unsigned)(ST_W_FL + 2 * (4 * wave + half) * ST_PROW + l32) * 4u;
    unsigned b_addr = (unsigned)l32 * 4u;
    asm volatile("" : "+v"(a_addr), "+v"(b_addr));
    float fa[2][2], fb[2][2];
#define SPK_ST_FRAG(k_, f_)                                                                                                   \
    {                                                                                                                         \
        constexpr int ci_ = (k_) / 49, ky_ = ((k_) % 49) / 7, kx_ = (k_) % 7;                                                 \
        constexpr int ia_ = ((ci_ * ST_PR + ky_) * 2 + (kx_ & 1)) * ST_PH + (kx_ >> 1);                                       \
        fa[f_][0] = *(const volatile st_lds_f32*)((st_lds_u8*)smem + (a_addr + (unsigned)(ia_ * 4)));                         \
        fa[f_][1] = *(const volatile st_lds_f32*)((st_lds_u8*)smem + (a_addr + (unsigned)((ia_ + 4 * ST_PROW) * 4)));         \
        fb[f_][0] = *(const volatile st_lds_f32*)((st_lds_u8*)smem + (b_addr + (unsigned)((k_) * 256)));                      \
        fb[f_][1] = *(const volatile st_lds_f32*)((st_lds_u8*)smem + (b_addr + (unsigned)((k_) * 256 + 128)));               \
    }
    SPK_ST_FRAG(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    st_static_for<0, ST_K>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        if constexpr (k + 1 < ST_K) {
            SPK_ST_FRAG(k + 1, (k + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F | (4 << 8));          // lgkmcnt(4): this step's fragments, not the next one's
        } else {
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[pr][n] = __builtin_amdgcn_mfma_f32_32x32x1f32(fa[k & 1][pr], fb[k & 1][n], acc[pr][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    });
#undef SPK_ST_FRAG

    // ---- epilogue: accumulator i of block blk (= output row): pixel ox = (i & 3) + 8 (i >> 2) + 4 half, channel n 32 + l32 ----
    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int oy = oy0 + 4 * wave + 2 * pr + blk;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float* yrow = p.y + (((size_t)b * p.Cy + (size_t)grp * 64 + n * 32 + l32) * p.H + oy) * p.W;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int ox = ox0 + 8 * q4 + 4 * half;
                    const bool in = oy < p.H && ox < p.W;                  // (W % 4 == 0: a group of four is in or out)
                    const float v0 = acc[pr][n][16 * blk + 4 * q4], v1 = acc[pr][n][16 * blk + 4 * q4 + 1];
                    const float v2 = acc[pr][n][16 * blk + 4 * q4 + 2], v3 = acc[pr][n][16 * blk + 4 * q4 + 3];
                    if (in) {
                        *reinterpret_cast<float4*>(yrow + ox) = make_float4(v0, v1, v2, v3);
                        if (STATS) {
                            ssum[n] += (v0 + v1) + (v2 + v3);
                            ssq[n] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                        }
                    }
                }
            }
        }
    if (STATS) {
        float* red = smem + ST_W_FL + ST_P_FL;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float s = ssum[n] + __shfl_xor(ssum[n], 32), q = ssq[n] + __shfl_xor(ssq[n], 32);
            if (half == 0) {
                red[(wave * 64 + n * 32 + l32) * 2] = s;
                red[(wave * 64 + n * 32 + l32) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < 64) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s += (double)red[(w * 64 + tid) * 2];
                q += (double)red[(w * 64 + tid) * 2 + 1];
            }
            const int cg = grp * 64 + tid;
            double* sp = p.stats + (size_t)((int)blockIdx.x % p.stats_slots) * 2 * p.Cy;
            if (p.stats_slots >= (int)gridDim.x) {            // this pixel tile owns its copy (the caller zeroed it)
                sp[cg] = s;
                sp[p.Cy + cg] = q;
            } else {
                atomicAdd(sp + cg, s);
                atomicAdd(sp + p.Cy + cg, q);
            }
        }
    }
}

// wp[g][k][co] = w_g[co][k], k = (ci, ky, kx) flattened; row 147 zero
__global__ __launch_bounds__(256) void pack_stem_kernel(PackList list, float* __restrict__ wp) {
    const int g = blockIdx.y;
    const float* w = list.w[g];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < ST_W_FL; e += gridDim.x * 256) {
        const int k = e >> 6, co = e & 63;
        wp[(size_t)g * ST_W_FL + e] = k < ST_K ? w[co * ST_K + k] : 0.f;
    }
}

}  // namespace

bool stem_takes(int kh, int stride, int Cin, int Cout, int H, int W) {
    static const bool allow = [] { const char* e = getenv("SPK_CONV_STEM"); return !e || atoi(e) != 0; }();
    return allow && kh == 7 && stride == 2 && Cin == 3 && Cout == 64 && W % 4 == 0 && H >= 1;
}
long long stem_packed_floats() { return ST_W_FL; }
void stem_tiles(int H, int W, int* tiles_x, int* tiles_y) {
    *tiles_x = spk::ceil_div(W, ST_TW);
    *tiles_y = spk::ceil_div(H, ST_TH);
}

int pack_stem(const PackList& list, int n, float* w_packed, int Cin, int Cout, int transpose_flip, hipStream_t stream) {
    SPK_REQUIRE(Cin == 3 && Cout == 64 && transpose_flip == 0, "pack_weights: config %d packs the 7x7 stem weight [64][3][7][7] (no transpose)", kStemConfig);
    hipLaunchKernelGGL(pack_stem_kernel, dim3(8, (unsigned)n), dim3(256), 0, stream, list, w_packed);
    return spk::check_launch("pack_stem_kernel");
}

int run_stem(const spk_conv2d_desc* d, hipStream_t stream) {
    const int G = d->groups > 1 ? d->groups : 1;
    SPK_REQUIRE(stem_takes(d->kh, d->stride, d->Cin, d->Cout, d->H, d->W) && d->kw == 7, "conv2d: config %d is the 7x7 stride-2 stem (Cin 3, Cout 64, W %% 4 == 0)", kStemConfig);
    SPK_REQUIRE(!(d->flags & ~SPK_EPI_STATS) && d->out_scale == 1.f && !d->out_scale_dev && !d->out_scale_bc && !d->y_pre,
                "conv2d: the stem form takes SPK_EPI_STATS only");
    SPK_REQUIRE(((reinterpret_cast<uintptr_t>(d->y) | reinterpret_cast<uintptr_t>(d->w_packed)) & 15) == 0, "conv2d: stem form: y and w_packed must be 16-byte aligned");
    SPK_REQUIRE((long long)d->Hin * d->Win * 3 < (1ll << 29), "conv2d: stem form: image too large for 32-bit offsets");
    StemArgs a;
    a.x = d->x; a.w = d->w_packed; a.y = d->y; a.stats = (d->flags & SPK_EPI_STATS) ? d->stats : nullptr;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Hin = d->Hin; a.Win = d->Win;
    a.gin = G > 1 ? d->group_in_stride : 0;
    a.Cx = a.gin * (G - 1) + 3;
    a.Cy = G * 64;
    stem_tiles(d->H, d->W, &a.tiles_x, &a.tiles_y);
    a.stats_slots = d->stats_slots > 1 ? d->stats_slots : 1;
    const long long tiles = (long long)a.tiles_x * a.tiles_y * d->B;
    SPK_REQUIRE(tiles < (1ll << 31), "conv2d: stem form: too many tiles");
    auto kern = a.stats ? &stem7x7s2_kernel<true> : &stem7x7s2_kernel<false>;
    static bool raised[2] = {false, false};
    if (!raised[a.stats ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        raised[a.stats ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)G), dim3(256), ST_LDS_BYTES, stream, a);
    return spk::check_launch("stem7x7s2_kernel");
}

}  // namespace spkconv
