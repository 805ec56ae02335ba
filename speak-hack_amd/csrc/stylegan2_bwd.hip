// Backward pieces of the build-defined StyleGAN2 decoder variant (formulas:
// reference/styleganv2.txt:1835,1912) that are NOT the MFMA conv / weight-gradient kernels themselves:
//
//   modconv_dx_finish   what is left of the modulated conv's data path after the MFMA data-gradient conv: the adjoint of
//                       upfirdn2d(up = 2, [1,3,3,1]) (x2 layers), the modulation factor s[b,ci] on the way out, and the
//                       modulation gradient d s[b,ci] = <dx~, up(x)> = <up^T(dx~), x> -- evaluated at the LOW resolution,
//                       so neither up(x) (268 MB at [8,128,256,256]) nor the rescaled gradient ever exists in HBM;
//   demod_bwd           the adjoint of d[b,co] = rsqrt(scale^2 sum_ci s^2 sum_k w^2 + eps) w.r.t. s and w (two small
//                       contractions; round 2 ran them as ATen GEMMs through autograd);
//   torgb_mod_bwd_data  the data gradient of the modulated 1x1 toRGB with the modulation folded into the LDS weights.
#include "spk_common.hpp"

#include <algorithm>
#include <cstdint>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ inline float block_sum_256(float v, float* red) {       // fixed order: lanes by xor-shuffle, then the 4 waves in order
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- x2 layers: dx[b,c,m,n] = s[b,c] * sum_{a,b' in -1..2} k[a] k[b'] g[b,c,2m+a,2n+b'],  k = (.25,.75,.75,.25), zero outside;
//      ds[b,c] = sum_{m,n} (dx / s)[m,n] * x[b,c,m,n].  One workgroup per (b,c) plane; a thread produces two adjacent
//      low-resolution pixels from four 16-byte row loads + 8 halo dwords of the x2 gradient (coalesced: thread q reads
//      columns 4q .. 4q+3).  Needs Ws % 2 == 0 and 16-byte aligned rows (W = 2 Ws % 4 == 0).
__global__ __launch_bounds__(256) void modconv_dx_finish_up_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                  const float* __restrict__ s, float* __restrict__ dx,
                                                                  float* __restrict__ ds, int Hs, int Ws) {
    __shared__ float red[4];
    const size_t plane = blockIdx.x;
    const int H = 2 * Hs, W = 2 * Ws, half_w = Ws / 2;
    const float* gp = g + plane * (size_t)H * W;
    const float* xp = x + plane * (size_t)Hs * Ws;
    float* dp = dx ? dx + plane * (size_t)Hs * Ws : nullptr;
    const float sv = s[plane];
    float dot = 0.f;
    const int items = Hs * half_w;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int m = it / half_w, q = it - m * half_w;          // low-res row m, low-res columns 2q, 2q+1
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int yy = 2 * m - 1 + r;
            if ((unsigned)yy >= (unsigned)H) continue;
            const float ky = (r == 0 || r == 3) ? 0.25f : 0.75f;
            const float* row = gp + (size_t)yy * W;
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q);
            const float l = q > 0 ? row[4 * q - 1] : 0.f;
            const float rr = 4 * q + 4 < W ? row[4 * q + 4] : 0.f;
            // low-res column 2q: x2 columns 4q-1 .. 4q+2; column 2q+1: 4q+1 .. 4q+4
            a0 += ky * (0.25f * l + 0.75f * v[0] + 0.75f * v[1] + 0.25f * v[2]);
            a1 += ky * (0.25f * v[1] + 0.75f * v[2] + 0.75f * v[3] + 0.25f * rr);
        }
        const f32x2 xv = *reinterpret_cast<const f32x2*>(xp + (size_t)m * Ws + 2 * q);
        dot += a0 * xv[0] + a1 * xv[1];
        if (dp) *reinterpret_cast<f32x2*>(dp + (size_t)m * Ws + 2 * q) = f32x2{a0 * sv, a1 * sv};
    }
    const float tot = block_sum_256(dot, red);
    if (threadIdx.x == 0) ds[plane] = tot;
}

// ---- same-resolution layers: dx = g * s[b,c] (in place allowed), ds[b,c] = <g, x>.  One workgroup per plane.
__global__ __launch_bounds__(256) void modconv_dx_finish_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                               const float* __restrict__ s, float* __restrict__ dx,
                                                               float* __restrict__ ds, long long HW, int vec) {
    __shared__ float red[4];
    const size_t plane = blockIdx.x;
    const float* gp = g + plane * (size_t)HW;
    const float* xp = x + plane * (size_t)HW;
    float* dp = dx ? dx + plane * (size_t)HW : nullptr;
    const float sv = s[plane];
    float dot = 0.f;
    if (vec) {
        for (long long i = threadIdx.x; i < HW / 4; i += 256) {
            const f32x4 gv = reinterpret_cast<const f32x4*>(gp)[i], xv = reinterpret_cast<const f32x4*>(xp)[i];
            dot += (gv[0] * xv[0] + gv[1] * xv[1]) + (gv[2] * xv[2] + gv[3] * xv[3]);
            if (dp) reinterpret_cast<f32x4*>(dp)[i] = gv * sv;
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += 256) {
            const float gv = gp[i];
            dot += gv * xp[i];
            if (dp) dp[i] = gv * sv;
        }
    }
    const float tot = block_sum_256(dot, red);
    if (threadIdx.x == 0) ds[plane] = tot;
}

// ---- demodulation adjoint.  d[b,co] = (scale^2 sum_ci s[b,ci]^2 w2[co,ci] + eps)^(-1/2), w2 = sum_k w^2.  With
//      e[b,co] = -dd[b,co] * d[b,co]^3 * scale^2:
//        ds[b,ci]      += s[b,ci] * sum_co e[b,co] * w2[co,ci]
//        dw[co,ci,k]   += w[co,ci,k] * sum_b e[b,co] * s[b,ci]^2
//      One pass over the weights: a workgroup owns 64 input channels (the lanes) x DBW_CO output channels (wave v takes
//      co0 + v, co0 + v + 4, ...); a weight element's taps are read once, w2 feeds the ds sums, m = sum_b e s^2 the dw update.
//      The ds sums of a workgroup's output-channel chunk go to partial[chunk][b][ci]; demod_bwd_finish_kernel adds the chunks
//      in order (no atomics: bitwise reproducible).  (The first version ran a thread per ci over ALL co: 2 workgroups for a
//      512-channel layer, 518 us; this one is ~10.)
constexpr int DBW_CO = 32, DBW_BT = 16;
__global__ __launch_bounds__(256) void demod_bwd_kernel(const float* __restrict__ w, const float* __restrict__ s,
                                                       const float* __restrict__ d, const float* __restrict__ dd,
                                                       float* __restrict__ partial, float* __restrict__ dw, int B, int Cin, int Cout,
                                                       int taps, float scale2) {
    __shared__ float e_s[DBW_BT][DBW_CO];
    __shared__ float red[4][DBW_BT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ci = blockIdx.x * 64 + lane, co0 = blockIdx.y * DBW_CO;
    const bool ci_ok = ci < Cin;
    float m[DBW_CO / 4];
#pragma unroll
    for (int i = 0; i < DBW_CO / 4; ++i) m[i] = 0.f;
    for (int b0 = 0; b0 < B; b0 += DBW_BT) {
        __syncthreads();
        for (int i = threadIdx.x; i < DBW_BT * DBW_CO; i += 256) {
            const int b = i / DBW_CO, c = i - b * DBW_CO;
            float v = 0.f;
            if (b0 + b < B && co0 + c < Cout) {
                const size_t idx = (size_t)(b0 + b) * Cout + co0 + c;
                const float dv = d[idx];
                v = -dd[idx] * dv * dv * dv * scale2;
            }
            e_s[b][c] = v;
        }
        __syncthreads();
        float s2[DBW_BT], acc[DBW_BT];
#pragma unroll
        for (int b = 0; b < DBW_BT; ++b) {
            const float sv = (ci_ok && b0 + b < B) ? s[(size_t)(b0 + b) * Cin + ci] : 0.f;
            s2[b] = sv * sv;
            acc[b] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < DBW_CO / 4; ++i) {
            const int c = wave + 4 * i, co = co0 + c;
            float w2 = 0.f;
            if (ci_ok && co < Cout) {
                const float* wr = w + ((size_t)co * Cin + ci) * taps;
                for (int k = 0; k < taps; ++k) w2 += wr[k] * wr[k];
            }
            float mm = 0.f;
#pragma unroll
            for (int b = 0; b < DBW_BT; ++b) {
                const float ev = e_s[b][c];
                acc[b] += ev * w2;
                mm += ev * s2[b];
            }
            m[i] += mm;
        }
        if (partial) {
#pragma unroll
            for (int b = 0; b < DBW_BT; ++b) red[wave][b][lane] = acc[b];
            __syncthreads();
            for (int i = threadIdx.x; i < DBW_BT * 64; i += 256) {
                const int b = i >> 6, l = i & 63;
                if (b0 + b < B && blockIdx.x * 64 + l < Cin)
                    partial[((size_t)blockIdx.y * B + b0 + b) * Cin + blockIdx.x * 64 + l] =
                        (red[0][b][l] + red[1][b][l]) + (red[2][b][l] + red[3][b][l]);
            }
        }
    }
    if (dw && ci_ok) {
#pragma unroll
        for (int i = 0; i < DBW_CO / 4; ++i) {
            const int co = co0 + wave + 4 * i;
            if (co < Cout) {
                const size_t base = ((size_t)co * Cin + ci) * taps;
                for (int k = 0; k < taps; ++k) dw[base + k] += w[base + k] * m[i];
            }
        }
    }
}

__global__ __launch_bounds__(256) void demod_bwd_finish_kernel(const float* __restrict__ partial, const float* __restrict__ s,
                                                              float* __restrict__ ds, int chunks, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int c = 0; c < chunks; ++c) v += partial[(size_t)c * n + i];
    ds[i] += s[i] * v;
}

// ---- everything a modulated conv's backward needs from the epilogue adjoint's plane sums, one launch (round 3's first form
//      did it in ~10 ATen launches per layer on [B,C] tensors).  sums[b][k][c], k: {sum dy*y, sum dy, sum dt, sum dt*noise}.
//        dprime[b,c] = d[b,c] * gain            (the factor on dz for the data / weight gradient; d = 1 without demodulation)
//        dd[b,c]     = (sum dy*y - gain*nw[c]*sum dt*noise - gain*bias[c]*sum dt) / d[b,c]
//        dbias[c]    = gain * sum_b sum dt,  dnw[c] = gain * sum_b sum dt*noise
__global__ __launch_bounds__(256) void modconv_epi_finish_kernel(const float* __restrict__ sums, const float* __restrict__ d,
                                                                const float* __restrict__ bias, const float* __restrict__ nw,
                                                                float gain, float* __restrict__ dd, float* __restrict__ dprime,
                                                                float* __restrict__ dbias, float* __restrict__ dnw, int B, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float bv = bias ? bias[c] : 0.f, nv = nw ? nw[c] : 0.f;
    float sb = 0.f, sn = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* row = sums + (size_t)b * 4 * C;
        const float s_dyy = row[c], s_dt = row[2 * C + c], s_dtn = row[3 * C + c];
        const float dv = d ? d[(size_t)b * C + c] : 1.f;
        dprime[(size_t)b * C + c] = dv * gain;
        if (dd) dd[(size_t)b * C + c] = (s_dyy - gain * nv * s_dtn - gain * bv * s_dt) / dv;
        sb += s_dt;
        sn += s_dtn;
    }
    if (dbias) dbias[c] = gain * sb;
    if (dnw) dnw[c] = gain * sn;
}

// ---- modulated toRGB, data gradient: dx[b,c,p] = in_scale * s[b,c] * sum_o w[o,c] dy[b,o,p]: the per-image weights
//      w[o,c] * in_scale * s[b,c] are formed in LDS (blockIdx.y = b), one streaming pass with 16-byte accesses.
template <bool VEC>
__global__ __launch_bounds__(256) void torgb_mod_bwd_data_kernel(const float* __restrict__ w, const float* __restrict__ mod,
                                                                const float* __restrict__ dy, float* __restrict__ dx, int C, int O,
                                                                long long HW, float in_scale) {
    extern __shared__ float w_s[];  // [O][C]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < O * C; i += blockDim.x) w_s[i] = w[i] * in_scale * mod[(size_t)b * C + (i % C)];
    __syncthreads();
    const float* dyb = dy + (size_t)b * O * HW;
    float* dxb = dx + (size_t)b * C * HW;
    if (VEC) {
        const long long n4 = HW / 4;
        for (long long p4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; p4 < n4; p4 += (long long)gridDim.x * blockDim.x) {
            f32x4 gq[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) gq[o] = o < O ? reinterpret_cast<const f32x4*>(dyb + (size_t)o * HW)[p4] : f32x4{0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < C; ++c) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < O) v += gq[o] * w_s[o * C + c];
                reinterpret_cast<f32x4*>(dxb + (size_t)c * HW)[p4] = v;
            }
        }
    } else {
        for (long long pp = (long long)blockIdx.x * blockDim.x + threadIdx.x; pp < HW; pp += (long long)gridDim.x * blockDim.x) {
            float gq[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) gq[o] = o < O ? dyb[(size_t)o * HW + pp] : 0.f;
            for (int c = 0; c < C; ++c) {
                float v = 0.f;
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < O) v += w_s[o * C + c] * gq[o];
                dxb[(size_t)c * HW + pp] = v;
            }
        }
    }
}

}  // namespace

extern "C" {

int spk_modconv_dx_finish(const float* g, const float* x, const float* s, float* dx, float* ds, int B, int C, int Hs, int Ws,
                          int upsample, void* stream) {
    SPK_REQUIRE(g && x && s && ds, "modconv_dx_finish: null pointer");
    SPK_REQUIRE(B > 0 && C > 0 && Hs > 0 && Ws > 0 && (long long)B * C < (1ll << 31), "modconv_dx_finish: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (upsample) {
        SPK_REQUIRE(Ws % 2 == 0 && ((uintptr_t)g % 16 == 0) && ((uintptr_t)x % 8 == 0) && (!dx || (uintptr_t)dx % 8 == 0),
                    "modconv_dx_finish: the x2 form needs an even low-resolution width and 16-byte aligned tensors");
        SPK_REQUIRE(dx != g, "modconv_dx_finish: the x2 form cannot run in place");
        hipLaunchKernelGGL(modconv_dx_finish_up_kernel, dim3((unsigned)(B * C)), dim3(256), 0, st, g, x, s, dx, ds, Hs, Ws);
        return spk::check_launch("modconv_dx_finish_up_kernel");
    }
    const long long HW = (long long)Hs * Ws;
    const int vec = (HW % 4 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)x % 16 == 0) && (!dx || (uintptr_t)dx % 16 == 0);
    hipLaunchKernelGGL(modconv_dx_finish_kernel, dim3((unsigned)(B * C)), dim3(256), 0, st, g, x, s, dx, ds, HW, vec);
    return spk::check_launch("modconv_dx_finish_kernel");
}

int spk_modconv_epi_finish(const float* sums, const float* d, const float* bias, const float* noise_w, float gain, float* dd,
                           float* dprime, float* dbias, float* dnw, int B, int C, void* stream) {
    SPK_REQUIRE(sums && dprime && B > 0 && C > 0, "modconv_epi_finish: null pointer or bad shape");
    SPK_REQUIRE(!dd || d, "modconv_epi_finish: dd (the demodulation gradient) needs d");
    SPK_REQUIRE((!dbias || bias) && (!dnw || noise_w), "modconv_epi_finish: dbias / dnw go with bias / noise_w");
    hipLaunchKernelGGL(modconv_epi_finish_kernel, dim3((unsigned)spk::ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, d, bias,
                       noise_w, gain, dd, dprime, dbias, dnw, B, C);
    return spk::check_launch("modconv_epi_finish_kernel");
}

int64_t spk_modconv_demod_bwd_workspace_bytes(int B, int Cin, int Cout) {
    if (B <= 0 || Cin <= 0 || Cout <= 0) return -1;
    return (int64_t)spk::ceil_div(Cout, DBW_CO) * B * Cin * (int64_t)sizeof(float);
}

int spk_modconv_demod_bwd(const float* w, const float* s, const float* d, const float* dd, float* ds, float* dw, void* workspace,
                          int64_t workspace_bytes, int B, int Cin, int Cout, int taps, float scale, void* stream) {
    SPK_REQUIRE(w && s && d && dd, "modconv_demod_bwd: null pointer");
    SPK_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && taps > 0, "modconv_demod_bwd: bad shape");
    if (!ds && !dw) return SPK_OK;
    const int chunks = spk::ceil_div(Cout, DBW_CO);
    SPK_REQUIRE(!ds || (workspace && workspace_bytes >= spk_modconv_demod_bwd_workspace_bytes(B, Cin, Cout)),
                "modconv_demod_bwd: ds needs a %lld-byte workspace (spk_modconv_demod_bwd_workspace_bytes)",
                (long long)spk_modconv_demod_bwd_workspace_bytes(B, Cin, Cout));
    hipStream_t st = (hipStream_t)stream;
    float* partial = ds ? static_cast<float*>(workspace) : nullptr;
    hipLaunchKernelGGL(demod_bwd_kernel, dim3((unsigned)spk::ceil_div(Cin, 64), (unsigned)chunks), dim3(256), 0, st, w, s, d, dd, partial, dw,
                       B, Cin, Cout, taps, scale * scale);
    int rc = spk::check_launch("demod_bwd_kernel");
    if (rc != SPK_OK || !ds) return rc;
    const long long n = (long long)B * Cin;
    hipLaunchKernelGGL(demod_bwd_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, s, ds, chunks, n);
    return spk::check_launch("demod_bwd_finish_kernel");
}

int spk_torgb_mod_bwd_data(const float* w, const float* mod, const float* dy, float* dx, int B, int C, int O, int64_t HW,
                           float in_scale, void* stream) {
    SPK_REQUIRE(w && mod && dy && dx, "torgb_mod_bwd_data: null pointer");
    SPK_REQUIRE(B > 0 && C > 0 && O > 0 && O <= 4 && HW > 0, "torgb_mod_bwd_data: bad shape (O must be <= 4)");
    SPK_REQUIRE((size_t)O * C * sizeof(float) <= 48 * 1024, "torgb_mod_bwd_data: weight too large for LDS");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (HW % 4 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0);
    const long long work = vec ? HW / 4 : HW;
    dim3 grid((unsigned)std::max(1ll, std::min((work + 255) / 256, 256ll)), (unsigned)B);
    const size_t lds = (size_t)O * C * sizeof(float);
    if (vec) hipLaunchKernelGGL(torgb_mod_bwd_data_kernel<true>, grid, dim3(256), lds, st, w, mod, dy, dx, C, O, (long long)HW, in_scale);
    else     hipLaunchKernelGGL(torgb_mod_bwd_data_kernel<false>, grid, dim3(256), lds, st, w, mod, dy, dx, C, O, (long long)HW, in_scale);
    return spk::check_launch("torgb_mod_bwd_data_kernel");
}

}  // extern "C"
