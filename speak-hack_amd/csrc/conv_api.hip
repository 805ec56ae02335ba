// C-ABI entry points of the conv family (include/spk.h), config tables / heuristics, the split-K
// finishing kernel and the weight packer.  The MFMA kernel template lives in conv_mfma_f32.hpp and
// is instantiated in conv_inst_*.hip.
#include "conv_mfma_f32.hpp"

namespace spkconv {

// y = epi( sum_z ws[z] ): fixed summation order (bitwise reproducible), one element per thread.
// BN statistics: a wave whose 64 elements share one output channel reduces with a butterfly and issues
// one fp64 atomic; ragged waves fall back to per-lane atomics.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvArgs p, const float* __restrict__ ws, int ksplit) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * p.Cout * HW;
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_noise = p.flags & SPK_EPI_NOISE;
    const bool f_lrelu = p.flags & SPK_EPI_LRELU, f_style = p.flags & SPK_EPI_STYLE;
    const bool f_accum = p.flags & SPK_EPI_ACCUM, f_stats = p.flags & SPK_EPI_STATS;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t rounds = (total + nthreads - 1) / nthreads;
    for (size_t it = 0; it < rounds; ++it) {   // every lane runs every round (wave-wide shuffles below)
        const size_t idx = it * nthreads + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool in = idx < total;
        float v = 0.f;
        int co = -1;
        if (in) {
            for (int z = 0; z < ksplit; ++z) v += ws[(size_t)z * total + idx];
            const size_t pix = idx % HW;
            co = (int)((idx / HW) % p.Cout);
            const int b = (int)(idx / (HW * p.Cout));
            v *= p.out_scale_dev ? p.out_scale * *p.out_scale_dev : p.out_scale;
            if (p.out_scale_bc) v *= p.out_scale_bc[(size_t)b * p.Cout + co];
            if (f_bias) v += p.bias[co];
            if (f_noise) v += p.noise_w[co] * p.noise[(size_t)b * HW + pix];
            if (f_lrelu) v = (v > 0.f ? v : v * p.slope) * p.act_gain;
            if (p.y_pre) p.y_pre[idx] = v;
            if (f_style) {
                const float* st = p.style + (size_t)b * p.style_stride;
                v = v * (st[co] + 1.f) + st[p.Cout + co];
            }
            if (f_accum) v += p.y[idx];
            p.y[idx] = v;
        }
        if (f_stats) {
            const int co0 = __shfl(co, 0);
            if (__all(co == co0) && co0 >= 0) {
                float s = v, ss = v * v;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    s += __shfl_xor(s, off);
                    ss += __shfl_xor(ss, off);
                }
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(p.stats + co0, (double)s);
                    atomicAdd(p.stats + p.Cout + co0, (double)ss);
                }
            } else if (in) {
                atomicAdd(p.stats + co, (double)v);
                atomicAdd(p.stats + p.Cout + co, (double)v * v);
            }
        }
    }
}

// Vector form for planes that are a multiple of 4 floats: one thread finishes 4 consecutive outputs of one (b, co)
// plane with 16-byte loads of every slice; BatchNorm sums are reduced over the lanes that share the plane (the whole
// wave, or a power-of-two lane segment for tiny planes) before the fp64 atomics.
__global__ __launch_bounds__(256) void splitk_epilogue_vec_kernel(const ConvArgs p, const float* __restrict__ ws, int ksplit) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t total4 = (size_t)p.B * p.Cout * HW / 4, total = total4 * 4;
    const bool f_bias = p.flags & SPK_EPI_BIAS, f_noise = p.flags & SPK_EPI_NOISE;
    const bool f_lrelu = p.flags & SPK_EPI_LRELU, f_style = p.flags & SPK_EPI_STYLE;
    const bool f_accum = p.flags & SPK_EPI_ACCUM, f_stats = p.flags & SPK_EPI_STATS;
    const size_t q4 = HW / 4;                        // float4s per plane
    const int seg = q4 >= 64 ? 64 : (int)q4;         // lanes that share a plane (q4 is a power of two when < 64: checked on the host)
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t rounds = (total4 + nthreads - 1) / nthreads;
    for (size_t it = 0; it < rounds; ++it) {         // every lane runs every round (wave-wide shuffles below)
        const size_t i4 = it * nthreads + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool in = i4 < total4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int co = 0;
        if (in) {
            const size_t idx = i4 * 4;
            float4 acc = *reinterpret_cast<const float4*>(ws + idx);
            for (int z = 1; z < ksplit; ++z) {
                const float4 t = *reinterpret_cast<const float4*>(ws + (size_t)z * total + idx);
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
            v[0] = acc.x; v[1] = acc.y; v[2] = acc.z; v[3] = acc.w;
            const size_t pix = idx % HW;
            co = (int)((idx / HW) % p.Cout);
            const int b = (int)(idx / (HW * p.Cout));
            float scale = p.out_scale_dev ? p.out_scale * *p.out_scale_dev : p.out_scale;
            if (p.out_scale_bc) scale *= p.out_scale_bc[(size_t)b * p.Cout + co];
            const float bb = f_bias ? p.bias[co] : 0.f;
            const float nw = f_noise ? p.noise_w[co] : 0.f;
            float s0 = 1.f, s1 = 0.f;
            if (f_style) {
                const float* st = p.style + (size_t)b * p.style_stride;
                s0 = st[co] + 1.f; s1 = st[p.Cout + co];
            }
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f), old = nz;
            if (f_noise) nz = *reinterpret_cast<const float4*>(p.noise + (size_t)b * HW + pix);
            if (f_accum) old = *reinterpret_cast<const float4*>(p.y + idx);
            const float nzv[4] = {nz.x, nz.y, nz.z, nz.w}, oldv[4] = {old.x, old.y, old.z, old.w};
            float pre[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = v[k] * scale + bb + nw * nzv[k];
                if (f_lrelu) t = (t > 0.f ? t : t * p.slope) * p.act_gain;
                pre[k] = t;
                if (f_style) t = t * s0 + s1;
                v[k] = t + oldv[k];
            }
            if (p.y_pre) *reinterpret_cast<float4*>(p.y_pre + idx) = make_float4(pre[0], pre[1], pre[2], pre[3]);
            *reinterpret_cast<float4*>(p.y + idx) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (f_stats) {
            float s = (v[0] + v[1]) + (v[2] + v[3]);
            float ss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            for (int off = seg >> 1; off >= 1; off >>= 1) {
                s += __shfl_xor(s, off);
                ss += __shfl_xor(ss, off);
            }
            if (in && (threadIdx.x & (seg - 1)) == 0) {
                atomicAdd(p.stats + co, (double)s);
                atomicAdd(p.stats + p.Cout + co, (double)ss);
            }
        }
    }
}

int launch_splitk_epilogue(const ConvArgs& a_in, const float* ws, int ksplit, hipStream_t stream) {
    ConvArgs a = a_in;
    a.Cout = a_in.Cy;            // the finisher walks the output tensor: every group's channels
    const size_t out_floats = (size_t)a.B * a.Cout * a.H * a.W;
    const size_t HW = (size_t)a.H * a.W, q4 = HW / 4;
    // lanes of a wave must split evenly into planes for the segmented sums: q4 a multiple of 64, or a power of two below
    const bool seg_ok = HW % 4 == 0 && (q4 % 64 == 0 || (q4 < 64 && (q4 & (q4 - 1)) == 0));
    const bool aligned = (uintptr_t)ws % 16 == 0 && (uintptr_t)a.y % 16 == 0 && (!a.y_pre || (uintptr_t)a.y_pre % 16 == 0) &&
                         (!a.noise || (uintptr_t)a.noise % 16 == 0);
    if (seg_ok && aligned) {
        const unsigned blocks = (unsigned)std::min<size_t>((out_floats / 4 + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(splitk_epilogue_vec_kernel, dim3(blocks), dim3(256), 0, stream, a, ws, ksplit);
        return spk::check_launch("splitk_epilogue_vec_kernel");
    }
    const unsigned blocks = (unsigned)std::min<size_t>((out_floats + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, stream, a, ws, ksplit);
    return spk::check_launch("splitk_epilogue_kernel");
}

__global__ void pack_weights_kernel(const PackList list, float* __restrict__ wp, int taps, int Cin_orig,
                                    int opCin, int opCout, int CO_T, int CI_T, int n_chunks, int tf, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float* __restrict__ w = list.w[blockIdx.y];
    wp += (size_t)blockIdx.y * total;
    long long t = idx;
    const int co_in = (int)(t % CO_T); t /= CO_T;
    const int ci = (int)(t % CI_T); t /= CI_T;
    const int tap = (int)(t % taps); t /= taps;
    const int chunk = (int)(t % n_chunks);
    const int cot = (int)(t / n_chunks);
    const int co = cot * CO_T + co_in, cig = chunk * CI_T + ci;
    float v = 0.f;
    if (co < opCout && cig < opCin) {
        if (tf == 0) v = w[((size_t)co * Cin_orig + cig) * taps + tap];
        else if (tf == 1) v = w[((size_t)cig * Cin_orig + co) * taps + (taps - 1 - tap)];
        else if (tf == 3) {
            // parity kernels of ConvTranspose2d(4, stride 2, pad 1) (taps = 4, w is [Cin][Cout][4][4], Cin_orig = Cout here):
            // output channel co = q*Cout + co_o, q = (qy, qx); window position (a, b) = input pixel (y-1+a, x-1+b) for the
            // output pixel (2y-1+qy, 2x-1+qx): oy = 2*iy - 1 + ky gives ky = 2, 0 for qy = 0 and ky = 3, 1 for qy = 1.
            const int q = co / Cin_orig, co_o = co - q * Cin_orig, a = tap >> 1, b = tap & 1;
            const int ky = (q >> 1) ? (a ? 1 : 3) : (a ? 0 : 2), kx = (q & 1) ? (b ? 1 : 3) : (b ? 0 : 2);
            v = w[(((size_t)cig * Cin_orig + co_o) * 4 + ky) * 4 + kx];
        } else {
            // parity kernels of the 3x3 stride-2 pad-1 data gradient (taps = 4 here, w is [Cout][Cin][3][3]): output
            // channel co = q*Cin + ci, q = (py, px); window position (a, b) = gradient pixel (m+a, n+b) for the input
            // pixel (2m+py, 2n+px).  Even coordinate: only a = 0 reaches it, through tap 1; odd: a = 0 through tap 2,
            // a = 1 through tap 0.
            const int q = co / Cin_orig, ci_o = co - q * Cin_orig, a = tap >> 1, b = tap & 1;
            const int ky = (q >> 1) ? (a ? 0 : 2) : (a ? -1 : 1), kx = (q & 1) ? (b ? 0 : 2) : (b ? -1 : 1);
            if (ky >= 0 && kx >= 0) v = w[(((size_t)cig * Cin_orig + ci_o) * 3 + ky) * 3 + kx];
        }
    }
    wp[idx] = v;
}

// The same image for tf = 0 / 1, one workgroup per (co tile, chunk) block of CO_T x CI_T x taps floats, through LDS: the kernel above
// reads one float per thread at a stride of Cin * taps floats (64 cache lines per wave-load; 271 us for the 57 MB of the trunk's
// layer-4 3x3 weights, a tenth of the HBM rate -- 0.54 ms of every generator step, since every weight is re-packed after the
// optimizer step).  Here the SOURCE is walked in its own order -- runs of CI_T * taps floats per output channel (tf = 0) or of
// CO_T * taps floats per input channel (tf = 1: the transposed, flipped operator) -- and the packed order is formed in LDS.
__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(const PackList list, float* __restrict__ wp, int taps, int Cin_orig,
                                                                int opCin, int opCout, int CO_T, int CI_T, int n_chunks, int tf,
                                                                long long total) {
    extern __shared__ float tile[];                       // [CO_T][CI_T * taps] (tf = 0) or [CI_T][CO_T * taps] (tf = 1)
    const int chunk = blockIdx.x % n_chunks, cot = blockIdx.x / n_chunks;
    const float* __restrict__ w = list.w[blockIdx.y];
    const int n = CO_T * CI_T * taps, run = tf ? CO_T * taps : CI_T * taps, pitch = run | 1;      // (odd pitch: bank-conflict-free columns)
    constexpr int U = 20;                                 // loads per thread, ALL in flight before the first LDS store (n <= 256 U: host)
    // element e = (row, rem) of the block in SOURCE order: row = co_in (tf = 0) / ci (tf = 1), rem runs over the row's contiguous
    // floats; stepping e by 256 is an add and a carry, the source offset is row * (Cin_orig * taps) + rem from the block's origin
    const size_t origin = tf ? ((size_t)chunk * CI_T * Cin_orig + (size_t)cot * CO_T) * taps
                             : ((size_t)cot * CO_T * Cin_orig + (size_t)chunk * CI_T) * taps;
    const int rowstride = Cin_orig * taps;
    const bool edge = (cot + 1) * CO_T > opCout || (chunk + 1) * CI_T > opCin;      // (uniform) ragged channels: per-element checks
    const int q256 = 256 / run, r256 = 256 - q256 * run;
    int row = threadIdx.x / run, rem = threadIdx.x - row * run;
    float v[U];
    int dst[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        bool ok = threadIdx.x + 256 * k < n;
        if (edge && ok) {
            const int sub = rem / taps;
            const int co = cot * CO_T + (tf ? sub : row), cig = chunk * CI_T + (tf ? row : sub);
            ok = co < opCout && cig < opCin;
        }
        const float x = w[ok ? origin + (size_t)row * rowstride + rem : origin];
        v[k] = ok ? x : 0.f;
        dst[k] = threadIdx.x + 256 * k < n ? row * pitch + rem : -1;
        row += q256; rem += r256;
        if (rem >= run) { rem -= run; ++row; }
    }
#pragma unroll
    for (int k = 0; k < U; ++k)
        if (dst[k] >= 0) tile[dst[k]] = v[k];
    __syncthreads();
    float* out = wp + (size_t)blockIdx.y * total + (size_t)blockIdx.x * n;
    for (int e = threadIdx.x; e < n; e += 256) {          // packed order: [tap][ci][co_in]
        const int co_in = e % CO_T, t = e / CO_T;
        const int ci = t % CI_T, tap = t / CI_T;
        out[e] = tf ? tile[ci * pitch + co_in * taps + (taps - 1 - tap)] : tile[co_in * pitch + ci * taps + tap];
    }
}

// config 12: w[Cout][Cin] -> itself (tf = 0) or its transpose [Cin][Cout] (tf = 1)
__global__ void pack_rowmajor_kernel(const PackList list, float* __restrict__ wp, int Cin, int Cout, int tf, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float* __restrict__ w = list.w[blockIdx.y];
    wp += (size_t)blockIdx.y * n;
    if (!tf) { wp[idx] = w[idx]; return; }
    const int co = (int)(idx % Cout), ci = (int)(idx / Cout);       // wp[ci][co]
    wp[idx] = w[(size_t)co * Cin + ci];
}

struct CfgDims { int co_t, ci_t, pix_t; };
static const CfgDims kDims[kNumConfigs] = {
    {Cfg0::CO_T, Cfg0::CI_T, Cfg0::PIX_T},   {Cfg1::CO_T, Cfg1::CI_T, Cfg1::PIX_T},   {Cfg2::CO_T, Cfg2::CI_T, Cfg2::PIX_T},
    {Cfg3::CO_T, Cfg3::CI_T, Cfg3::PIX_T},   {Cfg4::CO_T, Cfg4::CI_T, Cfg4::PIX_T},   {Cfg5::CO_T, Cfg5::CI_T, Cfg5::PIX_T},
    {Cfg6::CO_T, Cfg6::CI_T, Cfg6::PIX_T},   {Cfg7::CO_T, Cfg7::CI_T, Cfg7::PIX_T},   {Cfg8::CO_T, Cfg8::CI_T, Cfg8::PIX_T},
    {Cfg9::CO_T, Cfg9::CI_T, Cfg9::PIX_T},   {Cfg10::CO_T, Cfg10::CI_T, Cfg10::PIX_T}, {Cfg11::CO_T, Cfg11::CI_T, Cfg11::PIX_T},
    {128, 32, 128}, {64, 8, 128}, {128, 16, 128}, {64, 16, 128}, {64, 3, 512}};

static bool supported_kernel(int kh, int kw, int stride) {
    if (kh == 2 && kw == 2) return stride == 1;   // the parity form of the 3x3 stride-2 data gradient (SPK_CONV_DGRAD_S2)
    if (kh == 4 && kw == 4) return stride == 2;   // the data gradient of ConvTranspose2d(4, stride 2, pad 1) (styleganv1.py:231)
    return kh == kw && (kh == 1 || kh == 3 || kh == 7) && (stride == 1 || stride == 2) && !(kh == 7 && stride == 1);
}

static bool config_valid(int cfg, int kh, int kw, int stride) {
    if (!supported_kernel(kh, kw, stride) || cfg < 0 || cfg >= kNumConfigs) return false;
    if (cfg == kGemmConfig || is_gemm2(cfg)) return kh == 1 && stride == 1;
    if (cfg == kDgradS2Config) return kh == 2;
    if (cfg == kStemConfig) return kh == 7 && stride == 2;
    if (kh == 1) return cfg >= 8 && cfg <= 11;
    if (kh == 2) return cfg <= 3;
    if (kh == 3 && stride == 1) return cfg <= 7;
    return cfg >= 4 && cfg <= 7;  // 3x3 s2, 4x4 s2, 7x7 s2
}

template <int KH, int S>
static Geometry geom_k(int cfg, int B, int Cin, int Cout, int H, int W) {
    switch (cfg) {
        case 0: return geometry<Cfg0, KH, KH, S>(B, Cin, Cout, H, W);
        case 1: return geometry<Cfg1, KH, KH, S>(B, Cin, Cout, H, W);
        case 2: return geometry<Cfg2, KH, KH, S>(B, Cin, Cout, H, W);
        case 3: return geometry<Cfg3, KH, KH, S>(B, Cin, Cout, H, W);
        case 4: return geometry<Cfg4, KH, KH, S>(B, Cin, Cout, H, W);
        case 5: return geometry<Cfg5, KH, KH, S>(B, Cin, Cout, H, W);
        case 6: return geometry<Cfg6, KH, KH, S>(B, Cin, Cout, H, W);
        case 7: return geometry<Cfg7, KH, KH, S>(B, Cin, Cout, H, W);
        case 8: return geometry<Cfg8, KH, KH, S>(B, Cin, Cout, H, W);
        case 9: return geometry<Cfg9, KH, KH, S>(B, Cin, Cout, H, W);
        case 10: return geometry<Cfg10, KH, KH, S>(B, Cin, Cout, H, W);
        default: return geometry<Cfg11, KH, KH, S>(B, Cin, Cout, H, W);
    }
}

static Geometry geometry_any(int kh, int stride, int cfg, int B, int Cin, int Cout, int H, int W) {
    Geometry g;
    if (!config_valid(cfg, kh, kh, stride)) { g.ok = false; return g; }
    if (cfg == kGemmConfig) {            // no tile geometry to fit: pixel tiles of 128 over the flattened (b, pix) axis
        g.ok = gemm1x1_takes(kh, stride, Cin, H, W);
        g.TW = g.TH = g.TB = 1; g.PLANE = 1;
        g.tiles_x = (int)gemm1x1_pixel_tiles(B, H, W); g.tiles_y = g.tiles_b = 1;
        g.n_chunks = 1;                  // never splits K
        g.co_tiles = spk::ceil_div(Cout, 128);
        g.lds_bytes = 0;
        return g;
    }
    if (is_gemm2(cfg)) {                 // pixel tiles of 128 over the flattened (b, pix) axis; a one-dimensional grid
        g.ok = gemm2_takes(kh, stride, Cin, Cout, H, W);
        g.TW = g.TH = g.TB = 1; g.PLANE = 1;
        g.tiles_x = (int)gemm2_pixel_tiles(B, H, W); g.tiles_y = g.tiles_b = 1;
        g.n_chunks = 1;                  // never splits K
        g.co_tiles = spk::ceil_div(Cout, gemm2_co_tile(cfg));
        g.lds_bytes = 0;
        return g;
    }
    if (cfg == kStemConfig) {            // 16 x 32 pixel tiles of one image; all 64 channels of a group per workgroup
        g.ok = stem_takes(kh, stride, Cin, Cout, H, W);
        g.TW = 32; g.TH = 16; g.TB = 1; g.PLANE = 37 * 69;
        stem_tiles(H, W, &g.tiles_x, &g.tiles_y); g.tiles_b = B;
        g.n_chunks = 1;                  // never splits K
        g.co_tiles = 1;
        g.lds_bytes = 0;
        return g;
    }
    if (cfg == kDgradS2Config) {         // Cout = 4 x the input-side channels (the four classes); everything ragged is masked
        g.ok = Cout % 4 == 0 && dgrad_s2_fused_takes(B, Cin, Cout / 4, H, W);
        g.TW = 16; g.TH = 8; g.TB = 1; g.PLANE = 9 * 17;
        g.tiles_x = spk::ceil_div(W, 16); g.tiles_y = spk::ceil_div(H, 8); g.tiles_b = B;
        g.n_chunks = 1;                  // never splits K
        g.co_tiles = spk::ceil_div(Cout / 4, 64);
        g.lds_bytes = 0;
        return g;
    }
    if (kh == 1) return stride == 1 ? geom_k<1, 1>(cfg, B, Cin, Cout, H, W) : geom_k<1, 2>(cfg, B, Cin, Cout, H, W);
    if (kh == 2) return geom_k<2, 1>(cfg, B, Cin, Cout, H, W);
    if (kh == 3) return stride == 1 ? geom_k<3, 1>(cfg, B, Cin, Cout, H, W) : geom_k<3, 2>(cfg, B, Cin, Cout, H, W);
    if (kh == 4) return geom_k<4, 2>(cfg, B, Cin, Cout, H, W);
    return geom_k<7, 2>(cfg, B, Cin, Cout, H, W);
}

// Heuristic from measurements on MI355X (tools/bench_conv.py, profiles/): the shallow-chunk configs
// win on every decoder layer because 2-3 workgroups fit a CU and cover each other's barriers and
// epilogues; the wide-pixel tile for Cout <= 64; the small tile when the whole problem is small.
static int pick_config(int kh, int stride, int B, int Cin, int Cout, int H, int W) {
    const long long pixels = (long long)B * H * W;
    // Cout = 128, 3x3 stride 1, the fixed-geometry builds (tools/bench_conv.py, profiles/r03_u_conv_config_ab.txt): the
    // wide-pixel tile (64 co x 256 px, three per CU) wins with Cin = 256 (256 -> 128 @128^2 x2: 552 us against 568 at B=8, 1085 /
    // 1107 at B=16) and with Cin <= 128 once there are >= 256K pixels (128 -> 128 @128^2, B=16: 584 / 596); at B=8 that layer is
    // 1024 workgroups = 1.33 rounds of 768 slots and the square tile's two full rounds of 512 win (305 against 323).
    const bool wide = Cout <= 64 ? pixels >= 64 * 1024
                                 : (Cout <= 128 && Cin <= 256 && kh == 3 && stride == 1 && pixels >= 128 * 1024 &&
                                    !(Cin <= 128 && pixels < 256 * 1024));
    // 1x1 on the tap kernel: the 64x64 tile beats the 128x128 one up to 16^2 x 8 pixels, and for Cout = 128 up to 32^2 x 8
    // (sweep of configs 8-12 over the trunk's shapes, 6 groups: e.g. 256->1024 @16^2 data gradient 76 us against 85)
    const bool small1x1 = kh == 1 && (pixels <= 2048 || (Cout <= 128 && pixels <= 8192));
    // 3x3 stride-1 with a short contraction and few pixels (the trunk's 128 -> 128 @32^2 and 256 -> 256 @16^2 layers, per group):
    // the 64x64 tile fills the chip where the 128x128 one leaves CUs idle (config sweep of round 1, -0.5 ms per G step; it
    // changes the summation order of the discriminator's small layers, which the round-1 gradient criterion could not absorb)
    static const bool small3_on = [] { const char* e = getenv("SPK_CONV_SMALL3X3"); return !e || atoi(e) != 0; }();
    const bool small3x3 = small3_on && kh == 3 && stride == 1 && Cin <= 256 && Cout > 64 && pixels <= 8192;
    const int shape = Cout <= 32 ? 3 : (wide ? 1 : (Cout <= 64 || small1x1 || small3x3 ? 2 : (pixels >= 2048 ? 0 : 2)));
    const int base = kh == 1 ? 8 : (kh == 2 ? 0 : 4);   // 1x1 -> ids 8-11, 2x2 -> 0-3; everything else prefers ids 4-7
    const int lo = kh == 1 ? 8 : ((kh == 3 && stride == 1) || kh == 2 ? 0 : 4), hi = kh == 1 ? 11 : (kh == 2 ? 3 : 7);
    // The GEMM form of a stride-1 1x1 (conv1x1_gemm.hip, config 12), where it measured faster than the tap kernel on the
    // trunk's shapes (tools/bench_encoder_layers.py, SPK_CONV1X1_GEMM=1 forces it everywhere it applies, =0 nowhere): a
    // full 128-row block of output channels, a contraction no deeper than the output is wide (it has no split-K), and
    // enough pixels -- e.g. 64->256 @64^2: 123 us against 168, 128->512 @32^2: 96 against 119; but 2048->512 @8^2: 200
    // against 87.
    // The three-per-CU GEMM form (conv1x1_gemm2.hip, configs 14 / 15); SPK_CONV1X1_GEMM2 = 1 forces it wherever it
    // applies, 0 nowhere, 2 = 64-row tiles everywhere, 3 = 128-row tiles everywhere (tools/bench_encoder_layers.py).
    static const int gemm2_mode = [] { const char* e = getenv("SPK_CONV1X1_GEMM2"); return e ? atoi(e) : -1; }();
    if (kh == 1 && stride == 1 && gemm2_mode != 0 && gemm2_takes(kh, stride, Cin, Cout, H, W)) {
        if (gemm2_mode > 0)
            return gemm2_mode == 2 ? kGemm2NarrowConfig : (gemm2_mode == 3 ? kGemm2Config : (Cout <= 64 ? kGemm2NarrowConfig : kGemm2Config));
        // Measured on the trunk's shapes, 6 groups x batch 8 (profiles/r03_*: tools/lab_gemm2.py, tools/bench_encoder_layers.py): the lean
        // form wins wherever the launch fills the chip -- >= 768 workgroups of 128 rows (three per CU), else >= 768 of 64 rows --
        // and the contraction is not so deep that a few long workgroups are all there is (Cin <= 512; 1024 -> 256 @16^2 ties with
        // the tap kernel's split-K, 2048 -> 512 @8^2 loses).  The heuristic does not see the group count: it assumes the trunk's six.
        const long long px_tiles = gemm2_pixel_tiles(B, H, W);
        if (Cin <= 512) {
            if (Cout > 64 && px_tiles * spk::ceil_div(Cout, 128) >= 128) return kGemm2Config;
            if (px_tiles * spk::ceil_div(Cout, 64) >= 128) return kGemm2NarrowConfig;
        }
    }
    static const int gemm_mode = [] { const char* e = getenv("SPK_CONV1X1_GEMM"); return e ? atoi(e) : -1; }();
    if (kh == 1 && stride == 1 && gemm_mode != 0 && gemm1x1_takes(kh, stride, Cin, H, W)) {
        const bool wins = Cout >= 128 && ((Cin <= Cout && pixels >= 2048) || (Cin <= 2 * Cout && pixels >= 8192));
        if (gemm_mode > 0 ? (Cout >= 64 && pixels >= 128) : wins) return kGemmConfig;
    }
    if (stem_takes(kh, stride, Cin, Cout, H, W)) return kStemConfig;     // (SPK_CONV_STEM=0: the tap kernel)
    const int want = base + shape;
    if (geometry_any(kh, stride, want, B, Cin, Cout, H, W).ok) return want;
    static const int alt[4][3] = {{2, 3, 1}, {2, 3, 0}, {3, 0, 1}, {2, 0, 1}};
    for (int i = 0; i < 3; ++i) {
        const int c = base + alt[shape][i];
        if (geometry_any(kh, stride, c, B, Cin, Cout, H, W).ok) return c;
    }
    for (int c = lo; c <= hi; ++c)
        if (geometry_any(kh, stride, c, B, Cin, Cout, H, W).ok) return c;
    return want;
}

}  // namespace spkconv

using namespace spkconv;

extern "C" {

int spk_conv2d_num_configs(void) { return kNumConfigs; }

int spk_conv2d_config_valid(int config, int kh, int kw, int stride) { return config_valid(config, kh, kw, stride) ? 1 : 0; }

int spk_conv2d_pick_config(int kh, int kw, int stride, int B, int Cin, int Cout, int H, int W) {
    SPK_REQUIRE(supported_kernel(kh, kw, stride), "conv2d: unsupported kernel %dx%d stride %d", kh, kw, stride);
    SPK_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv2d: bad shape");
    return pick_config(kh, stride, B, Cin, Cout, H, W);
}

int spk_conv2d_dgrad_s2_config(int B, int Cin, int Cout, int Hin, int Win) {
    SPK_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "conv2d: bad shape");
    // the exact-tap kernel everywhere it applies (SPK_DGRAD_S2_FUSED=0: the four zero-padded 2x2 kernels of the general kernel)
    static const bool fused_on = [] { const char* e = getenv("SPK_DGRAD_S2_FUSED"); return !e || atoi(e) != 0; }();
    // (its pixel tile is 16 x 8: on the trunk's 8 x 8 gradients, 512 -> 512 x 6 groups, half of every tile is empty and it
    // measured 308 us against 228 for the 2x2 form; 138 against 277 at 32 x 32, 165 against 246 at 16 x 16)
    if (fused_on && Win > 8 && dgrad_s2_fused_takes(B, Cin, Cout, Hin, Win)) return kDgradS2Config;
    return pick_config(2, 1, B, Cin, 4 * Cout, Hin, Win);
}

int spk_conv2d_config_info(int config, int* co_tile, int* ci_tile, int* pix_tile) {
    SPK_REQUIRE(config >= 0 && config < kNumConfigs, "conv2d: bad config %d", config);
    if (co_tile) *co_tile = kDims[config].co_t;
    if (ci_tile) *ci_tile = kDims[config].ci_t;
    if (pix_tile) *pix_tile = kDims[config].pix_t;
    return SPK_OK;
}

int64_t spk_conv2d_packed_floats(int config, int kh, int kw, int Cin, int Cout) {
    if (config < 0 || config >= kNumConfigs || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0) return -1;
    if (config == kGemmConfig) return kh == 1 && kw == 1 ? (int64_t)Cout * Cin : -1;      // plain [Cout][Cin]
    if (is_gemm2(config)) return kh == 1 && kw == 1 ? (int64_t)gemm2_packed_floats(config, Cin, Cout) : -1;
    if (config == kStemConfig) return kh == 7 && kw == 7 && Cin == 3 && Cout == 64 ? (int64_t)stem_packed_floats() : -1;
    if (config == kDgradS2Config) return (kh == 2 && kw == 2 && Cout % 4 == 0) ? (int64_t)dgrad_s2_fused_packed_floats(Cin, Cout / 4) : -1;
    const CfgDims& c = kDims[config];
    return (int64_t)spk::ceil_div(Cout, c.co_t) * spk::ceil_div(Cin, c.ci_t) * kh * kw * c.ci_t * c.co_t;
}

int64_t spk_conv2d_workspace_bytes_grouped(int config, int ksplit, int kh, int kw, int stride, int B, int Cin, int Cout,
                                           int H, int W, int groups) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || groups < 1 || !supported_kernel(kh, kw, stride)) return -1;
    if (config < 0) config = pick_config(kh, stride, B, Cin, Cout, H, W);
    Geometry g = geometry_any(kh, stride, config, B, Cin, Cout, H, W);
    if (!g.ok) return -1;
    g.co_tiles *= groups;
    const int ks = resolve_ksplit(g, ksplit, nullptr);
    return ks > 1 ? (int64_t)ks * B * groups * Cout * H * W * (int64_t)sizeof(float) : 0;
}

int64_t spk_conv2d_dgrad_s2_workspace_bytes(int B, int Cin, int Cout, int Hin, int Win, int H, int W, int groups) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Hin <= 0 || Win <= 0 || H <= 0 || W <= 0 || groups < 1) return -1;
    if (spk_conv2d_dgrad_s2_config(B, Cin, Cout, Hin, Win) != kDgradS2Config) return 0;
    const int ks = dgrad_s2_ksplit(B, Cin, Cout, Hin, Win, groups);
    return ks > 1 ? (int64_t)ks * B * groups * Cout * H * W * (int64_t)sizeof(float) : 0;
}

int spk_conv2d_stats_slots(int config, int kh, int kw, int stride, int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || !supported_kernel(kh, kw, stride)) return -1;
    if (config < 0) config = pick_config(kh, stride, B, Cin, Cout, H, W);
    const Geometry g = geometry_any(kh, stride, config, B, Cin, Cout, H, W);
    if (!g.ok) return -1;
    return g.tiles_x * g.tiles_y * g.tiles_b;            // = gridDim.x of the launch
}

int64_t spk_conv2d_workspace_bytes(int config, int ksplit, int kh, int kw, int stride, int B, int Cin, int Cout,
                                   int H, int W) {
    return spk_conv2d_workspace_bytes_grouped(config, ksplit, kh, kw, stride, B, Cin, Cout, H, W, 1);
}

int spk_conv2d_pack_weights(const float* w, float* w_packed, int kh, int kw, int Cin, int Cout, int config,
                            int transpose_flip, void* stream) {
    return spk_conv2d_pack_weights_list(&w, 1, w_packed, kh, kw, Cin, Cout, config, transpose_flip, stream);
}

int spk_conv2d_pack_weights_list(const float* const* ws, int n, float* w_packed, int kh, int kw, int Cin, int Cout, int config,
                                 int transpose_flip, void* stream) {
    SPK_REQUIRE(ws && w_packed && n >= 1 && n <= SPK_PACK_LIST_MAX, "pack_weights: null pointer or bad list length (1..%d)", SPK_PACK_LIST_MAX);
    PackList list;
    for (int i = 0; i < SPK_PACK_LIST_MAX; ++i) {
        list.w[i] = ws[i < n ? i : 0];
        SPK_REQUIRE(list.w[i], "pack_weights: null pointer");
    }
    SPK_REQUIRE(config >= 0 && config < kNumConfigs, "pack_weights: bad config %d", config);
    SPK_REQUIRE(Cin > 0 && Cout > 0 && kh > 0 && kw > 0, "pack_weights: bad shape");
    SPK_REQUIRE(transpose_flip >= 0 && transpose_flip <= 3, "pack_weights: transpose_flip is 0, 1, 2 or 3");
    SPK_REQUIRE(transpose_flip != 2 || (kh == 3 && kw == 3), "pack_weights: the stride-2 data-gradient form packs a 3x3 kernel");
    SPK_REQUIRE(transpose_flip != 3 || (kh == 4 && kw == 4), "pack_weights: the transposed-conv form packs a 4x4 kernel");
    if (config == kGemmConfig) {        // row-major [opCout][opCin]: the weight itself, or its transpose for the data gradient
        SPK_REQUIRE(kh == 1 && kw == 1 && transpose_flip < 2, "pack_weights: config %d packs 1x1 kernels", kGemmConfig);
        const long long nf = (long long)Cin * Cout;
        hipLaunchKernelGGL(pack_rowmajor_kernel, dim3((unsigned)((nf + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, list,
                           w_packed, Cin, Cout, transpose_flip, nf);
        return spk::check_launch("pack_rowmajor_kernel");
    }
    if (is_gemm2(config)) {             // [co tile][k tile][16 k][CO_T] of the weight or of its transpose
        SPK_REQUIRE(kh == 1 && kw == 1 && transpose_flip < 2, "pack_weights: config %d packs 1x1 kernels", config);
        return pack_gemm2(list, n, w_packed, Cin, Cout, config, transpose_flip, (hipStream_t)stream);
    }
    if (config == kStemConfig) {        // [148 k][64 co], k = (ci, ky, kx)
        SPK_REQUIRE(kh == 7 && kw == 7, "pack_weights: config %d packs 7x7 kernels", config);
        return pack_stem(list, n, w_packed, Cin, Cout, transpose_flip, (hipStream_t)stream);
    }
    if (config == kDgradS2Config) {     // the transposed 3x3 operator itself, in 64-channel x 8-channel tiles (kDims[13])
        SPK_REQUIRE(transpose_flip == 2, "pack_weights: config %d packs the stride-2 data-gradient form (transpose_flip = 2)", kDgradS2Config);
        transpose_flip = 1;
    }
    const CfgDims& c = kDims[config];
    const int opCin = (transpose_flip == 1 || transpose_flip == 2) ? Cout : Cin;
    const int opCout = transpose_flip == 2 ? 4 * Cin : (transpose_flip == 3 ? 4 * Cout : (transpose_flip ? Cin : Cout));
    const int taps = transpose_flip >= 2 ? 4 : kh * kw;
    const int n_chunks = spk::ceil_div(opCin, c.ci_t);
    const long long total = (long long)spk::ceil_div(opCout, c.co_t) * n_chunks * taps * c.ci_t * c.co_t;
    const int threads = 256;
    const size_t tile_bytes = (transpose_flip ? (size_t)c.ci_t * ((c.co_t * taps) | 1) : (size_t)c.co_t * ((c.ci_t * taps) | 1)) * sizeof(float);
    if (transpose_flip <= 1 && tile_bytes <= 64 * 1024 && c.co_t * c.ci_t * taps <= 256 * 20) {
        const long long blocks = (long long)spk::ceil_div(opCout, c.co_t) * n_chunks;
        SPK_REQUIRE(blocks < (1ll << 31), "pack_weights: too many tiles");
        hipLaunchKernelGGL(pack_weights_tiled_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(threads), tile_bytes, (hipStream_t)stream, list,
                           w_packed, taps, Cin, opCin, opCout, c.co_t, c.ci_t, n_chunks, transpose_flip, total);
        return spk::check_launch("pack_weights_tiled_kernel");
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + threads - 1) / threads), (unsigned)n), dim3(threads), 0,
                       (hipStream_t)stream, list, w_packed, taps, transpose_flip == 3 ? Cout : Cin, opCin, opCout, c.co_t, c.ci_t,
                       n_chunks, transpose_flip, total);
    return spk::check_launch("pack_weights_kernel");
}

int spk_conv2d_fwd(const spk_conv2d_desc* d, void* stream) {
    SPK_REQUIRE(d, "conv2d: null descriptor");
    if (d->flags & SPK_CONV_BF16X3) return spk_conv2d_bf16x3_fwd(d, stream);
    if (d->flags & SPK_CONV_WINOGRAD) return spk_conv2d_wino_fwd(d, stream);
    SPK_REQUIRE(!(d->flags & SPK_EPI_TORGB), "conv2d: SPK_EPI_TORGB is an epilogue of the SPK_CONV_WINOGRAD launches only");
    SPK_REQUIRE(d->x && d->w_packed && d->y, "conv2d: null tensor pointer");
    SPK_REQUIRE(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->H > 0 && d->W > 0 && d->Hin > 0 && d->Win > 0, "conv2d: bad shape");
    if (d->flags & SPK_CONV_TRANSPOSE4X4_S2) {
        // forward of nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1): x [B,Cin,Hin,Win] -> y [B,Cout,2Hin,2Win]; the four
        // output-parity classes are 2x2 kernels over the (Hin+1) x (Win+1) window anchors, stored interleaved
        SPK_REQUIRE(d->kh == 4 && d->kw == 4 && d->stride == 2, "conv2d: TRANSPOSE4X4_S2 is a 4x4 stride-2 transposed conv");
        SPK_REQUIRE(!(d->flags & ~(SPK_CONV_TRANSPOSE4X4_S2 | SPK_EPI_BIAS | SPK_EPI_ACCUM)) && !d->out_scale_bc && !d->y_pre,
                    "conv2d: TRANSPOSE4X4_S2 takes SPK_EPI_BIAS / SPK_EPI_ACCUM only");
        SPK_REQUIRE(!(d->flags & SPK_EPI_BIAS) || d->bias, "conv2d: SPK_EPI_BIAS without bias");
        SPK_REQUIRE(d->H == 2 * d->Hin && d->W == 2 * d->Win, "conv2d: TRANSPOSE4X4_S2 output must be 2x the input (%dx%d vs %dx%d)",
                    d->H, d->W, d->Hin, d->Win);
        SPK_REQUIRE(d->groups <= 1, "conv2d: TRANSPOSE4X4_S2 is not grouped");
        SPK_REQUIRE((long long)d->B * d->Cout * d->H * d->W < (1ll << 40), "conv2d: tensor too large");
        spk_conv2d_desc dd = *d;
        dd.kh = dd.kw = 2; dd.stride = 1; dd.Cout = 4 * d->Cout; dd.H = d->Hin + 1; dd.W = d->Win + 1; dd.ksplit = 1;
        dd.flags = d->flags & (SPK_EPI_BIAS | SPK_EPI_ACCUM);
        if (dd.config < 0) dd.config = pick_config(2, 1, dd.B, dd.Cin, dd.Cout, dd.H, dd.W);
        SPK_REQUIRE(config_valid(dd.config, 2, 2, 1), "conv2d: TRANSPOSE4X4_S2 runs tile configs 0-3 (got %d)", dd.config);
        return run_2x2_parity(dd.config, &dd, d->H, d->W, 1, (hipStream_t)stream);
    }
    SPK_REQUIRE(supported_kernel(d->kh, d->kw, d->stride) && d->kh != 2, "conv2d: unsupported kernel %dx%d stride %d", d->kh, d->kw, d->stride);
    if (d->flags & SPK_CONV_DGRAD_S2) {
        // data gradient of a 3x3 stride-2 pad-1 conv: x = the output-side gradient [B, Cin, Hin, Win], y = the input-side
        // gradient [B, Cout, H, W]; run as the four output-parity classes (2x2 kernels, 4*Cout channels, interleaved stores)
        SPK_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 2, "conv2d: DGRAD_S2 is the data gradient of a 3x3 stride-2 conv");
        SPK_REQUIRE(!(d->flags & ~(SPK_CONV_DGRAD_S2 | SPK_EPI_ACCUM)) && !d->out_scale_bc && !d->y_pre,
                    "conv2d: DGRAD_S2 takes SPK_EPI_ACCUM only");
        SPK_REQUIRE((d->H == 2 * d->Hin || d->H == 2 * d->Hin - 1) && (d->W == 2 * d->Win || d->W == 2 * d->Win - 1),
                    "conv2d: DGRAD_S2 output %dx%d is not the input size of a stride-2 conv with output %dx%d", d->H, d->W, d->Hin, d->Win);
        SPK_REQUIRE(d->groups <= 1 || d->group_in_stride >= d->Cin, "conv2d: DGRAD_S2 groups read disjoint channels");
        SPK_REQUIRE((long long)d->B * d->Cout * d->H * d->W < (1ll << 40), "conv2d: tensor too large");
        spk_conv2d_desc dd = *d;
        dd.kh = dd.kw = 2; dd.stride = 1; dd.Cout = 4 * d->Cout; dd.H = d->Hin; dd.W = d->Win; dd.ksplit = 1;
        dd.flags = d->flags & SPK_EPI_ACCUM; dd.out_scale = d->out_scale;
        if (dd.config < 0) dd.config = spk_conv2d_dgrad_s2_config(d->B, d->Cin, d->Cout, d->Hin, d->Win);
        SPK_REQUIRE(config_valid(dd.config, 2, 2, 1), "conv2d: DGRAD_S2 runs tile configs 0-3 and %d (got %d)", kDgradS2Config, dd.config);
        if (dd.config == kDgradS2Config) return run_dgrad_s2_fused(d, (hipStream_t)stream);
        return run_2x2_parity(dd.config, &dd, d->H, d->W, 0, (hipStream_t)stream);
    }
    const bool ups = d->flags & SPK_CONV_UPSAMPLE2X, aff = d->flags & SPK_CONV_IN_AFFINE_RELU;
    const bool bsc = d->flags & SPK_CONV_IN_BATCH_SCALE;
    SPK_REQUIRE(!aff || (!ups && !bsc), "conv2d: IN_AFFINE_RELU excludes UPSAMPLE2X and IN_BATCH_SCALE");
    SPK_REQUIRE(!(d->flags & SPK_CONV_UP_FIR1331) || ups, "conv2d: UP_FIR1331 qualifies UPSAMPLE2X");
    SPK_REQUIRE(!bsc || (d->in_scale && d->kh == 3 && d->stride == 1), "conv2d: IN_BATCH_SCALE needs in_scale[B,Cin] and a 3x3 stride-1 kernel");
    SPK_REQUIRE(!ups || (d->kh == 3 && d->stride == 1), "conv2d: UPSAMPLE2X needs a 3x3 stride-1 kernel");
    const int pad = (d->kh - 1) / 2;
    if (ups) SPK_REQUIRE(d->H == 2 * d->Hin && d->W == 2 * d->Win, "conv2d: upsampled output must be 2x the input (%dx%d vs %dx%d)", d->H, d->W, d->Hin, d->Win);
    else SPK_REQUIRE(d->H == (d->Hin + 2 * pad - d->kh) / d->stride + 1 && d->W == (d->Win + 2 * pad - d->kw) / d->stride + 1,
                     "conv2d: output size %dx%d does not match input %dx%d (k=%d s=%d)", d->H, d->W, d->Hin, d->Win, d->kh, d->stride);
    SPK_REQUIRE(!(d->flags & SPK_EPI_BIAS) || d->bias, "conv2d: SPK_EPI_BIAS without bias");
    SPK_REQUIRE(!(d->flags & SPK_EPI_NOISE) || (d->noise && d->noise_w), "conv2d: SPK_EPI_NOISE without noise");
    SPK_REQUIRE(!(d->flags & SPK_EPI_STYLE) || d->style, "conv2d: SPK_EPI_STYLE without style");
    SPK_REQUIRE(!(d->flags & SPK_EPI_STATS) || d->stats, "conv2d: SPK_EPI_STATS without stats");
    SPK_REQUIRE(d->stats_slots >= 0 && d->stats_slots <= 65536, "conv2d: stats_slots must be in [0, 65536] (got %d)", d->stats_slots);
    SPK_REQUIRE(!aff || (d->in_scale && d->in_shift), "conv2d: IN_AFFINE_RELU without in_scale/in_shift");
    SPK_REQUIRE((long long)d->B * d->Cout * d->H * d->W < (1ll << 40), "conv2d: tensor too large");
    if (d->groups > 1) {
        SPK_REQUIRE(!(d->flags & ~(SPK_EPI_BIAS | SPK_EPI_LRELU | SPK_EPI_ACCUM | SPK_EPI_STATS | SPK_CONV_IN_AFFINE_RELU | SPK_EPI_ACCUM_HALF)) && !d->out_scale_bc,
                    "conv2d: a grouped launch takes bias / lrelu / accum / accum-half / stats / in-affine only");
        SPK_REQUIRE(d->group_in_stride == 0 || d->group_in_stride >= d->Cin, "conv2d: group_in_stride must be 0 (shared input) or >= Cin");
    }
    int cfg = d->config;
    if (cfg < 0) cfg = pick_config(d->kh, d->stride, d->B, d->Cin, d->Cout, d->H, d->W);
    SPK_REQUIRE(!bsc || cfg >= 4, "conv2d: IN_BATCH_SCALE is built for the half-depth configs (4-7)");
    SPK_REQUIRE(!d->out_scale_bc || bsc, "conv2d: out_scale_bc (demodulation) goes with SPK_CONV_IN_BATCH_SCALE (the modulated convolution)");
    SPK_REQUIRE(config_valid(cfg, d->kh, d->kw, d->stride), "conv2d: config %d is not built for %dx%d stride %d", cfg, d->kh, d->kw, d->stride);
    spk_conv2d_desc dd = *d;
    dd.config = cfg;
    const int mode = ups ? (bsc ? MODE_UPSAMPLE_BATCH_SCALE : MODE_UPSAMPLE) : (aff ? MODE_AFFINE_RELU : (bsc ? MODE_BATCH_SCALE : MODE_PLAIN));
    hipStream_t s = (hipStream_t)stream;
    SPK_REQUIRE(!d->out_scale_dev || !(cfg == kGemmConfig || is_gemm2(cfg)), "conv2d: out_scale_dev is built into the tap kernels (not the GEMM forms of a 1x1)");
    if (cfg == kGemmConfig) return run_1x1_gemm(&dd, s);
    if (is_gemm2(cfg)) return run_1x1_gemm2(&dd, s);
    if (cfg == kStemConfig) {
        SPK_REQUIRE(mode == MODE_PLAIN && !(d->flags & SPK_EPI_ACCUM_HALF), "conv2d: the stem form reads a plain input");
        return run_stem(&dd, s);
    }
    SPK_REQUIRE(!(d->flags & SPK_EPI_ACCUM_HALF), "conv2d: SPK_EPI_ACCUM_HALF is built into the GEMM form of a 1x1 (configs 14, 15)");
    if (d->kh == 1) return run_1x1(d->stride, cfg, mode, &dd, s);
    if (d->kh == 3 && d->stride == 1) {
        if ((d->flags & SPK_EPI_STATS) && mode != MODE_AFFINE_RELU) {
            SPK_REQUIRE(mode == MODE_PLAIN, "conv2d: SPK_EPI_STATS on a 3x3 stride-1 conv goes with a plain or BatchNorm-folded input (not upsample / batch scale)");
            return run_3x3s1_stats(cfg, &dd, s);
        }
        return cfg <= 3 ? run_3x3s1_a(cfg, mode, &dd, s) : run_3x3s1_b(cfg, mode, &dd, s);
    }
    return run_3x3s2_7x7s2(d->kh, cfg, mode, &dd, s);
}

}  // extern "C"
