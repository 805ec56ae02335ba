// 1x1 stride-1 convolution as a plain GEMM on the gfx950 f32 MFMA pipe (tile config 12):
//     y[b, co, pix] = epi( sum_ci W[co, ci] * in(x[b, ci, pix]) ),   in = identity or max(x * scale[ci] + shift[ci], 0)
// The tap kernel (conv_mfma_f32.hpp) is built around 3x3 chunks of 36 k-steps; for a 1x1 a chunk is 8 k-steps, so its
// staging dominates (55-80 TFLOP/s on the ResNet-50 trunk's bottleneck convs, half of that trunk's launches).  A 1x1 needs
// no halo and no gather: rows of x are contiguous pixels, rows of W contiguous input channels.  A workgroup owns a
// 128co x 128px block (2x2 waves, each 2x2 MFMA tiles of 32x32), walks 32-channel k-tiles with 16-byte loads (4 + 4 per
// thread per k-tile against 64 MFMAs per wave), LDS double-buffered with one barrier per k-tile, fragments read one k-step
// ahead.  Pixels are the flattened (b, pix) axis; HW % 32 == 0 keeps every 32-pixel fragment row inside one image.
// Epilogue as in the tap kernel: out_scale -> bias -> lrelu -> accumulate -> store -> BatchNorm sums (reduced across the
// waves in LDS, one copy per pixel tile).  Grouped launches as there (blockIdx.y = group * co tiles + co tile).
//
// Epilogue: the 128 x 128 block goes through LDS so that a channel row leaves as 512 contiguous bytes of 16-byte stores;
// with the accumulators stored straight from the MFMA layout (128-byte pieces of a row at four different times) this
// kernel -- like the tap kernel -- wrote the trunk's layer1 outputs at ~1.2 TB/s and lost to the tap kernel everywhere
// (forward 8.2 ms over the trunk's 1x1 layers against 7.1); with the staged epilogue 64->256 @64^2 x 6 groups went from
// 188 to 123 us (tap kernel: 168).
// STATUS (round 1): spk_conv2d_pick_config returns it where it measured faster (Cout >= 128, contraction no deeper than
// ~the output width, >= 2048 pixels); it has no split-K, so the deep-K / few-pixel layers (2048->512 @8^2: 200 us
// against 87) stay on the tap kernel.  tools/bench_encoder_layers.py with SPK_CONV1X1_GEMM=0 / 1 compares the two.
//
// replaces: F.conv2d of every stride-1 1x1 conv of the torchvision trunk (conv1 / conv3 / downsample.0 of layer1,
// model.py:60-62) forward, and -- on the transposed weight -- its data gradient.
#include "conv_mfma_f32.hpp"

namespace spkconv {

namespace {

constexpr int GM = 128, GN = 128, GK = 32, APITCH = GK + 1, BPITCH = GN + 4;
constexpr int A_FLOATS = GM * APITCH, B_FLOATS = GK * BPITCH, GBUF = A_FLOATS + B_FLOATS;

struct GemmArgs {
    const float* x;
    const float* w;          // [G][Cout][Cin] row-major (the "packed" image of config 12)
    const float* bias;
    const float* in_scale;
    const float* in_shift;
    double* stats;
    float* y;
    int B, Cin, Cout, HW;    // Cin / Cout per group
    long long n_px;          // B * HW
    int G, Cx, Cy, gin, co_tiles_g;
    int stats_slots;
    unsigned flags;
    float slope, out_scale, act_gain;
#ifdef SPK_GEMM_LAB
    int lab;                 // knock-outs (lab build only): 1 no MFMAs, 2 no x loads, 4 no y stores
#endif
};

#ifdef SPK_GEMM_LAB
#define SPK_GLAB(bit_) (p.lab & (bit_))
#else
#define SPK_GLAB(bit_) false
#endif

template <bool AFF>
__global__ __launch_bounds__(256, 2) void conv1x1_gemm_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;

    const int grp = (int)blockIdx.y / p.co_tiles_g;
    const int co0 = ((int)blockIdx.y - grp * p.co_tiles_g) * GM;          // within the group
    const int cx0 = grp * p.gin;
    const long long P0 = (long long)blockIdx.x * GN;
    const size_t HW = (size_t)p.HW;

    // staging roles.  A: rows tid/8 + 32 i (i < 4), k = 4 (tid % 8) .. +3.  B: k rows tid/32 + 8 i, pixels 4 (tid % 32) .. +3.
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const int brow = tid >> 5, bcol = (tid & 31) * 4;
    const long long Pb = P0 + bcol;
    const bool b_ok = Pb < p.n_px;                                         // (n_px % 4 == 0: a vector is in or out as a whole)
    const float* xb = p.x;
    if (b_ok) {
        const long long b = Pb / p.HW;
        xb += ((size_t)b * p.Cx + cx0) * HW + (size_t)(Pb - b * p.HW);
    }
    const float* wb = p.w + ((size_t)grp * p.Cout + co0) * p.Cin;

    // Two register sets: a k-tile's loads are issued TWO compute phases (8K MFMA cycles) before its LDS store, which
    // covers an HBM miss; the folded affine is applied at the store, so nothing waits on a load before the MFMAs.
    struct Stage { float4 a[4], b[4]; float sc[4], sh[4]; };
    Stage sA, sB;
    // load() only LOADS (out-of-range rows / channels read a valid address); every select and the folded affine happen in
    // store(), two compute phases later, behind a scheduling fence: with the zero-select right behind its load the compiler
    // put an s_waitcnt in front of the MFMAs of the SAME phase (ISA: "Lx15 vmcnt(7) L vmcnt(1) MFMAx64") and the prefetch
    // distance was nil.
    auto load = [&](Stage& st, int kt) {
        const int k0 = kt * GK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = arow + 32 * i, k = k0 + acol;
            const bool a_ok = co0 + r < p.Cout && k < p.Cin;
            st.a[i] = *reinterpret_cast<const float4*>(wb + (a_ok ? (size_t)r * p.Cin + k : 0));
            const int ci = k0 + brow + 8 * i;
            const bool x_ok = b_ok && ci < p.Cin;
            if (!SPK_GLAB(2)) st.b[i] = *reinterpret_cast<const float4*>(xb + (x_ok ? (size_t)ci * HW : 0));
            else st.b[i] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (AFF) {
                // channels past Cin (or pixels past the tensor): scale 0, shift 0 -> max(0, 0) = 0
                st.sc[i] = p.in_scale[cx0 + (x_ok ? ci : 0)];
                st.sh[i] = p.in_shift[cx0 + (x_ok ? ci : 0)];
            }
        }
    };
    auto store = [&](const Stage& st, int buf, int kt) {
        __builtin_amdgcn_sched_barrier(0);
        float* as_ = smem + buf * GBUF;
        float* bs_ = as_ + A_FLOATS;
        const int k0 = kt * GK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool a_ok = co0 + arow + 32 * i < p.Cout && k0 + acol < p.Cin;
            const bool x_ok = b_ok && k0 + brow + 8 * i < p.Cin;
            const int o = (arow + 32 * i) * APITCH + acol;
            as_[o] = a_ok ? st.a[i].x : 0.f; as_[o + 1] = a_ok ? st.a[i].y : 0.f;
            as_[o + 2] = a_ok ? st.a[i].z : 0.f; as_[o + 3] = a_ok ? st.a[i].w : 0.f;
            float4 v = st.b[i];
            if (AFF) {
                v.x = fmaxf(v.x * st.sc[i] + st.sh[i], 0.f); v.y = fmaxf(v.y * st.sc[i] + st.sh[i], 0.f);
                v.z = fmaxf(v.z * st.sc[i] + st.sh[i], 0.f); v.w = fmaxf(v.w * st.sc[i] + st.sh[i], 0.f);
            }
            if (!x_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(bs_ + (brow + 8 * i) * BPITCH + bcol) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    auto compute = [&](int buf) {
        const float* as_ = smem + buf * GBUF + (wm * 64 + l32) * APITCH + half;
        const float* bs_ = smem + buf * GBUF + A_FLOATS + half * BPITCH + wn * 64 + l32;
        float fa[2][2], fb[2][2];
#define SPK_G_FRAG(ks_, slot_)                                                                               \
    {                                                                                                        \
        fa[slot_][0] = as_[2 * (ks_)]; fa[slot_][1] = as_[32 * APITCH + 2 * (ks_)];                          \
        fb[slot_][0] = bs_[2 * (ks_) * BPITCH]; fb[slot_][1] = bs_[2 * (ks_) * BPITCH + 32];                 \
    }
        SPK_G_FRAG(0, 0);
        static_for<0, GK / 2>([&](auto s_) {
            constexpr int ks = decltype(s_)::value;
            if constexpr (ks + 1 < GK / 2) {
                SPK_G_FRAG(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks & 1][m], fb[ks & 1][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
        });
#undef SPK_G_FRAG
    };

    const int n_kt = (p.Cin + GK - 1) / GK;
    load(sA, 0);
    store(sA, 0, 0);
    __syncthreads();
    // Every phase issues its loads UNCONDITIONALLY (past the last k-tile: that tile again, never stored): with the loads under
    // `if (kt + 2 < n_kt)` the compiler has to assume the short path and waits for vmcnt(0) before the stores -- i.e. for the
    // loads issued a moment ago, not only for the ones issued a phase earlier.
    load(sA, min(1, n_kt - 1));
    for (int kt = 0; kt < n_kt; kt += 2) {
        // tile kt out of buffer 0; sA = tile kt+1 (in flight since the previous phase); sB <- tile kt+2
        load(sB, min(kt + 2, n_kt - 1));
        if (!SPK_GLAB(1)) compute(0);
        store(sA, 1, min(kt + 1, n_kt - 1));
        __syncthreads();
        if (kt + 1 >= n_kt) break;
        // tile kt+1 out of buffer 1; sB = tile kt+2; sA <- tile kt+3
        load(sA, min(kt + 3, n_kt - 1));
        if (!SPK_GLAB(1)) compute(1);
        store(sB, 0, min(kt + 2, n_kt - 1));
        __syncthreads();
    }

    // ---- epilogue: the 128co x 128px block goes through LDS (it is exactly as large as the two staging buffers), so
    // that a channel row leaves as 512 contiguous bytes of 16-byte stores instead of four 128-byte pieces at different
    // times; a row is handled by one half-wave, so its BatchNorm sums need one 32-lane butterfly and no second stage ----
    constexpr int OPITCH = GN + 4;
    static_assert(GM * OPITCH <= 2 * GBUF, "the output tile reuses the staging buffers");
    float* const ot = smem;                                   // the last k-tile's barrier has passed
    float* const red = smem + 2 * GBUF;                       // [2][GM] row sums / sums of squares
#ifdef SPK_GEMM_LAB
    if (SPK_GLAB(8)) {
        float t = 0.f;
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) t += acc[m][n][r];
        if (t == 123.456f) p.y[tid] = t;
        return;
    }
#endif
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ot[(wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * OPITCH + wn * 64 + n * 32 + l32] = acc[m][n][r];
    __syncthreads();
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_lrelu = p.flags & SPK_EPI_LRELU;
    const bool f_accum = p.flags & SPK_EPI_ACCUM, f_stats = (p.flags & SPK_EPI_STATS) && !SPK_GLAB(16);
    const int ocol = (tid & 31) * 4;
    const long long Po = P0 + ocol;
    const bool o_ok = Po < p.n_px;
    size_t ooff = 0;
    if (o_ok) {
        const long long b = Po / p.HW;
        ooff = (size_t)b * p.Cy * HW + (size_t)(Po - b * p.HW);
    }
    double* sp = f_stats ? p.stats + (size_t)((int)blockIdx.x % p.stats_slots) * 2 * p.Cy : nullptr;
    const bool own_slot = p.stats_slots >= (int)gridDim.x;
#pragma unroll 4
    for (int i = 0; i < GM / 8; ++i) {
        const int cl = (tid >> 5) + 8 * i;
        const int co = co0 + cl;
        const bool cv = co < p.Cout;                           // uniform over the half-wave
        const int cg = grp * p.Cout + (cv ? co : 0);
        float ssum = 0.f, ssq = 0.f;
        if (cv && o_ok) {
            float4 v = *reinterpret_cast<const float4*>(ot + cl * OPITCH + ocol);
            const float bb = f_bias ? p.bias[cg] : 0.f;
            v.x = v.x * p.out_scale + bb; v.y = v.y * p.out_scale + bb; v.z = v.z * p.out_scale + bb; v.w = v.w * p.out_scale + bb;
            if (f_lrelu) {
                v.x = (v.x > 0.f ? v.x : v.x * p.slope) * p.act_gain; v.y = (v.y > 0.f ? v.y : v.y * p.slope) * p.act_gain;
                v.z = (v.z > 0.f ? v.z : v.z * p.slope) * p.act_gain; v.w = (v.w > 0.f ? v.w : v.w * p.slope) * p.act_gain;
            }
            float4* dst = reinterpret_cast<float4*>(p.y + ooff + (size_t)cg * HW);
            if (f_accum) {
                const float4 old = *dst;
                v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
            }
            if (!SPK_GLAB(4) || v.x == 123.456f) *dst = v;
            ssum = (v.x + v.y) + (v.z + v.w);
            ssq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        if (f_stats) {                                         // uniform flag: every lane takes part
            ssum = spk::half_wave_sum_hi(ssum);
            ssq = spk::half_wave_sum_hi(ssq);
            if (l32 == 31) {
                red[cl] = ssum;
                red[GM + cl] = ssq;
            }
        }
    }
    if (f_stats) {
        // the block's 128 + 128 row sums leave as two contiguous runs of doubles (one store instruction each)
        __syncthreads();
        const int cl = tid & (GM - 1), co = co0 + cl;
        if (co < p.Cout) {
            const int cg = grp * p.Cout + co;
            double* dst = sp + (tid >= GM ? p.Cy : 0) + cg;
            if (own_slot) *dst = (double)red[tid];
            else atomicAdd(dst, (double)red[tid]);
        }
    }
}

}  // namespace

bool gemm1x1_takes(int kh, int stride, int Cin, int H, int W) {
    return kh == 1 && stride == 1 && Cin % 4 == 0 && ((long long)H * W) % 32 == 0;
}

long long gemm1x1_pixel_tiles(int B, int H, int W) { return ((long long)B * H * W + GN - 1) / GN; }

int run_1x1_gemm(const spk_conv2d_desc* d, hipStream_t stream) {
    SPK_REQUIRE(gemm1x1_takes(d->kh, d->stride, d->Cin, d->H, d->W), "conv2d: config 12 (GEMM form) takes stride-1 1x1 convs with Cin %% 4 == 0 and H*W %% 32 == 0");
    SPK_REQUIRE(!(d->flags & ~(SPK_EPI_BIAS | SPK_EPI_LRELU | SPK_EPI_ACCUM | SPK_EPI_STATS | SPK_CONV_IN_AFFINE_RELU)) && !d->out_scale_bc && !d->y_pre,
                "conv2d: config 12 (GEMM form) takes bias / lrelu / accum / stats / in-affine only");
    SPK_REQUIRE(((reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->w_packed) | reinterpret_cast<uintptr_t>(d->y)) & 15) == 0,
                "conv2d: config 12 needs 16-byte aligned x, y and weights");
    GemmArgs a;
    a.x = d->x; a.w = d->w_packed; a.bias = d->bias; a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.stats = d->stats; a.y = d->y;
    a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.HW = d->H * d->W; a.n_px = (long long)d->B * a.HW;
    a.G = d->groups > 1 ? d->groups : 1;
    a.gin = a.G > 1 ? d->group_in_stride : d->Cin;
    a.Cx = a.gin * (a.G - 1) + d->Cin;
    a.Cy = a.G * d->Cout;
    a.co_tiles_g = spk::ceil_div(d->Cout, GM);
    a.stats_slots = d->stats_slots > 1 ? d->stats_slots : 1;
    a.flags = d->flags; a.slope = d->lrelu_slope; a.out_scale = d->out_scale; a.act_gain = d->act_gain != 0.f ? d->act_gain : 1.f;
#ifdef SPK_GEMM_LAB
    { const char* e = getenv("SPK_GEMM_LAB"); a.lab = e ? atoi(e) : 0; }
#endif
    const long long gx = gemm1x1_pixel_tiles(d->B, d->H, d->W);
    SPK_REQUIRE(gx < (1ll << 31) && (long long)a.G * a.co_tiles_g < 65536, "conv2d: grid too large");
    const size_t lds = (2 * (size_t)GBUF + 2 * GM) * sizeof(float);
    const bool aff = d->flags & SPK_CONV_IN_AFFINE_RELU;
    auto kern = aff ? &conv1x1_gemm_kernel<true> : &conv1x1_gemm_kernel<false>;
    static bool raised[2] = {false, false};
    if (!raised[aff]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return spk::fail(SPK_ELAUNCH, "hipFuncSetAttribute(LDS): %s", hipGetErrorString(e));
        raised[aff] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)(a.G * a.co_tiles_g)), dim3(256), lds, stream, a);
    return spk::check_launch("conv1x1_gemm_kernel");
}

}  // namespace spkconv
