// HBM-bound streaming kernels around the conv core: decoder prologue, toRGB 1x1, stand-alone
// bilinear x2.  All are one pass over the data with 16-byte per-lane accesses where the shape
// allows.  See include/spk.h for the reference call sites each replaces.
#include "spk_common.hpp"

#include <algorithm>

namespace {

// y[b,c,p] = (x[b*xbs + c*HW + p] + bias[c] + noise_w[c]*noise[b,p]) * (s0[b,c]+1) + s1[b,c]
__global__ __launch_bounds__(256) void bias_noise_style_kernel(const float* __restrict__ cin, long long xbs,
                                                              const float* __restrict__ bias,
                                                            const float* __restrict__ nw, const float* __restrict__ noise,
                                                            const float* __restrict__ style, long long style_stride,
                                                            float* __restrict__ y, int B, int C, int HW) {
    const long long total = (long long)B * C * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(idx % HW);
        const int c = (int)((idx / HW) % C);
        const int b = (int)(idx / ((long long)HW * C));
        float v = cin[(size_t)b * xbs + (size_t)c * HW + p];
        if (bias) v += bias[c];
        if (noise) v += nw[c] * noise[(size_t)b * HW + p];
        if (style) {
            const float* st = style + (size_t)b * style_stride;
            v = v * (st[c] + 1.f) + st[C + c];
        }
        y[idx] = v;
    }
}

// StyleGAN2's skip connection: upfirdn2d(skip, up = 2, FIR [1,3,3,1] * 4 / 16 per axis, pad (2,1)) evaluated at output
// pixel(s) of a [2Hs, 2Ws] image -- per axis out[2i] = .25 in[i-1] + .75 in[i], out[2i+1] = .75 in[i] + .25 in[i+1],
// samples outside the image count as zero.  The 3-channel skip image is tiny next to the 64..512-channel activations
// the toRGB kernel streams, so the upsample + add ride in its epilogue instead of being two more passes.
__device__ __forceinline__ float skip_px(const float* __restrict__ sk, int Hs, int Ws, int r, int c) {
    return (r >= 0 && r < Hs && c >= 0 && c < Ws) ? sk[(size_t)r * Ws + c] : 0.f;
}
__device__ __forceinline__ float upfir2x_1(const float* __restrict__ sk, int Hs, int Ws, int y, int x) {
    const int ra = (y >> 1) - 1 + (y & 1), ca = (x >> 1) - 1 + (x & 1);
    const float wa = (y & 1) ? 0.75f : 0.25f, wb = 1.f - wa, ua = (x & 1) ? 0.75f : 0.25f, ub = 1.f - ua;
    return wa * (ua * skip_px(sk, Hs, Ws, ra, ca) + ub * skip_px(sk, Hs, Ws, ra, ca + 1)) +
           wb * (ua * skip_px(sk, Hs, Ws, ra + 1, ca) + ub * skip_px(sk, Hs, Ws, ra + 1, ca + 1));
}
// four consecutive pixels x .. x+3 of row y (x a multiple of 4)
__device__ __forceinline__ float4 upfir2x_4(const float* __restrict__ sk, int Hs, int Ws, int y, int x) {
    const int ra = (y >> 1) - 1 + (y & 1), c0 = x >> 1;
    const float wa = (y & 1) ? 0.75f : 0.25f, wb = 1.f - wa;
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = wa * skip_px(sk, Hs, Ws, ra, c0 - 1 + k) + wb * skip_px(sk, Hs, Ws, ra + 1, c0 - 1 + k);
    return make_float4(0.25f * r[0] + 0.75f * r[1], 0.75f * r[1] + 0.25f * r[2], 0.25f * r[1] + 0.75f * r[2], 0.75f * r[2] + 0.25f * r[3]);
}

// 1x1 conv with O <= 4 outputs.  VEC: each thread owns 4 consecutive pixels.  ``skip`` (optional, [B,O,H/2,W/2] with W the
// row length of y): y += upfir2x(skip).
template <bool VEC>
__global__ __launch_bounds__(256) void conv1x1_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C,
                                                           int O, long long HW, float in_scale,
                                                           const float* __restrict__ mod, const float* __restrict__ skip, int W) {
    extern __shared__ float w_s[];  // [O][C]; with ``mod`` [B,C] the weight is modulated per image (StyleGAN2 toRGB)
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < O * C; i += blockDim.x) w_s[i] = w[i] * in_scale * (mod ? mod[(size_t)b * C + (i % C)] : 1.f);
    __syncthreads();
    const float* xb = x + (size_t)b * C * HW;
    float* yb = y + (size_t)b * O * HW;
    if (VEC) {
        const long long n4 = HW / 4;
        for (long long p4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; p4 < n4;
             p4 += (long long)gridDim.x * blockDim.x) {
            float4 acc[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float bo = (bias && o < O) ? bias[o] : 0.f;
                acc[o] = make_float4(bo, bo, bo, bo);
            }
#pragma unroll 8
            for (int c = 0; c < C; ++c) {
                const float4 xv = reinterpret_cast<const float4*>(xb + (size_t)c * HW)[p4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (o < O) {
                        const float wv = w_s[o * C + c];
                        acc[o].x += wv * xv.x; acc[o].y += wv * xv.y; acc[o].z += wv * xv.z; acc[o].w += wv * xv.w;
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < O) {
                    if (skip) {
                        const int Hs = (int)(HW / W) >> 1, Ws = W >> 1;
                        const long long p = 4 * p4;
                        const float4 u = upfir2x_4(skip + ((size_t)b * O + o) * Hs * Ws, Hs, Ws, (int)(p / W), (int)(p % W));
                        acc[o].x += u.x; acc[o].y += u.y; acc[o].z += u.z; acc[o].w += u.w;
                    }
                    reinterpret_cast<float4*>(yb + (size_t)o * HW)[p4] = acc[o];
                }
            }
        }
    } else {
        for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < HW;
             p += (long long)gridDim.x * blockDim.x) {
            float acc[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = (bias && o < O) ? bias[o] : 0.f;
            for (int c = 0; c < C; ++c) {
                const float xv = xb[(size_t)c * HW + p];
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < O) acc[o] += w_s[o * C + c] * xv;
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < O) {
                    if (skip) {
                        const int Hs = (int)(HW / W) >> 1, Ws = W >> 1;
                        acc[o] += upfir2x_1(skip + ((size_t)b * O + o) * Hs * Ws, Hs, Ws, (int)(p / W), (int)(p % W));
                    }
                    yb[(size_t)o * HW + p] = acc[o];
                }
            }
        }
    }
}

// Same op for small images with many channels (StyleGAN2's toRGB at 4^2..128^2 with 128-512 channels): too few pixels to fill
// the chip with one thread per 4 pixels, and a thread that walks all C channels alone is bound by the LATENCY of its C
// dependent-looking loads (the first form: 64 pixel quads x 4 channel slices per workgroup, 128 channels per thread -- 48 us
// for a 4 x 4 image).  Here a workgroup takes Q <= 16 pixel quads and splits the channels 256 / Q ways (16 ... 64 slices,
// 8 ... 32 channels per thread, all of a thread's loads issued before the first FMA); the partial sums meet in LDS.
constexpr int CS_BATCH = 16;
__global__ __launch_bounds__(256) void conv1x1_small_csplit_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y, int C,
                                                                  int O, long long HW, float in_scale,
                                                                  const float* __restrict__ mod, const float* __restrict__ skip, int W,
                                                                  int lgQ) {
    extern __shared__ float sm[];           // [O][C] weights, then [slices][4 outputs][Q] float4 partials
    float* w_s = sm;
    float4* part = reinterpret_cast<float4*>(sm + ((O * C + 3) & ~3));
    const int Q = 1 << lgQ, NS = 256 >> lgQ;
    const int b = blockIdx.y, q = threadIdx.x & (Q - 1), slice = threadIdx.x >> lgQ;
    for (int i = threadIdx.x; i < O * C; i += 256) w_s[i] = w[i] * in_scale * (mod ? mod[(size_t)b * C + (i % C)] : 1.f);
    __syncthreads();
    const float* xb = x + (size_t)b * C * HW;
    float* yb = y + (size_t)b * O * HW;
    const long long n4 = HW / 4;
    const long long p4 = (long long)blockIdx.x * Q + q;
    const int c0 = (int)((long long)C * slice / NS), c1 = (int)((long long)C * (slice + 1) / NS);
    float4 acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p4 < n4) {
        for (int cb = c0; cb < c1; cb += CS_BATCH) {
            float4 xv[CS_BATCH];
#pragma unroll
            for (int k = 0; k < CS_BATCH; ++k)          // every load of the batch is in flight before the first use
                xv[k] = reinterpret_cast<const float4*>(xb + (size_t)min(cb + k, c1 - 1) * HW)[p4];
#pragma unroll
            for (int k = 0; k < CS_BATCH; ++k) {
                if (cb + k < c1) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (o < O) {
                            const float wv = w_s[o * C + cb + k];
                            acc[o].x += wv * xv[k].x; acc[o].y += wv * xv[k].y; acc[o].z += wv * xv[k].z; acc[o].w += wv * xv[k].w;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) part[(slice * 4 + o) * Q + q] = acc[o];
    __syncthreads();
    // thread t < O * Q finishes output channel t / Q of quad t % Q: the slices in a fixed order
    const int fo = threadIdx.x >> lgQ, fq = threadIdx.x & (Q - 1);
    const long long fp4 = (long long)blockIdx.x * Q + fq;
    if (fo < O && fp4 < n4) {
        const float bo = bias ? bias[fo] : 0.f;
        float4 r = make_float4(bo, bo, bo, bo);
        for (int k = 0; k < NS; ++k) {
            const float4 t = part[(k * 4 + fo) * Q + fq];
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        if (skip) {
            const int Hs = (int)(HW / W) >> 1, Ws = W >> 1;
            const long long p = 4 * fp4;
            const float4 u = upfir2x_4(skip + ((size_t)b * O + fo) * Hs * Ws, Hs, Ws, (int)(p / W), (int)(p % W));
            r.x += u.x; r.y += u.y; r.z += u.z; r.w += u.w;
        }
        reinterpret_cast<float4*>(yb + (size_t)fo * HW)[fp4] = r;
    }
}

// bilinear x2, align_corners=False (torch area_pixel_compute_source_index with scale 0.5).
// One thread makes 4 consecutive outputs of a row (Wo % 4 == 0 always: Wo = 2*Win, handled per pair otherwise).
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        long long planes, int Hin, int Win) {
    const int Ho = 2 * Hin, Wo = 2 * Win;
    const int Wq = (Wo + 3) / 4;
    const long long total = planes * Ho * Wq;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % Wq);
        const int uy = (int)((idx / Wq) % Ho);
        const long long pl = idx / ((long long)Wq * Ho);
        const float sy = fmaxf(0.5f * (uy + 0.5f) - 0.5f, 0.f);
        const int iy0 = (int)sy, iy1 = min(iy0 + 1, Hin - 1);
        const float ly1 = sy - iy0, ly0 = 1.f - ly1;
        const float* r0 = x + (size_t)pl * Hin * Win + (size_t)iy0 * Win;
        const float* r1 = x + (size_t)pl * Hin * Win + (size_t)iy1 * Win;
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ux = min(4 * q + k, Wo - 1);
            const float sx = fmaxf(0.5f * (ux + 0.5f) - 0.5f, 0.f);
            const int ix0 = (int)sx, ix1 = min(ix0 + 1, Win - 1);
            const float lx1 = sx - ix0, lx0 = 1.f - lx1;
            out[k] = ly0 * (lx0 * r0[ix0] + lx1 * r0[ix1]) + ly1 * (lx0 * r1[ix0] + lx1 * r1[ix1]);
        }
        float* dst = y + (size_t)pl * Ho * Wo + (size_t)uy * Wo + 4 * q;
        if (Wo % 4 == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
            for (int k = 0; k < 4 && 4 * q + k < Wo; ++k) dst[k] = out[k];
        }
    }
}

// The same image when Win % 4 == 0: a thread takes four source columns of one source row m and makes the 2 x 8 outputs under them
// (rows 2m, 2m + 1) from the 3 x 6 source values around them -- three 16-byte loads + six neighbours, all issued before the first
// use, four 16-byte stores; plane / row / column from the launch grid (no 64-bit division).  Element for element the arithmetic of
// upsample2x_kernel (same taps, same order).  The x2 image of a Winograd x2 layer (SPK_OP_UPSAMPLE2X) is 270 MB of writes per
// decoder step at B = 8: the generic kernel above ran it at ~2 TB/s, a step's worth in 0.32 ms.
// ZB: neighbours outside the image count as zero instead of being clamped -- upfirdn2d(up = 2, FIR [1,3,3,1], pad (2,1)), the x2 of
// the StyleGAN2 variant's styled convs: the same (.25, .75) taps.
template <bool ZB>
__global__ __launch_bounds__(256) void upsample2x_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int Hin, int Win) {
    const int Wq = Win >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;                 // over Hin * Wq
    if (t >= Hin * Wq) return;
    const int m = t / Wq, qb = t - m * Wq, c0 = 4 * qb;
    const float* xp = x + (size_t)blockIdx.y * Hin * Win;
    const int r0 = max(m - 1, 0), r2 = min(m + 1, Hin - 1), cl = max(c0 - 1, 0), cr = min(c0 + 4, Win - 1);
    const float4 a = *reinterpret_cast<const float4*>(xp + (size_t)r0 * Win + c0);
    const float4 b = *reinterpret_cast<const float4*>(xp + (size_t)m * Win + c0);
    const float4 c = *reinterpret_cast<const float4*>(xp + (size_t)r2 * Win + c0);
    float al = xp[(size_t)r0 * Win + cl], ar = xp[(size_t)r0 * Win + cr];
    float bl = xp[(size_t)m * Win + cl], br = xp[(size_t)m * Win + cr];
    float cl_ = xp[(size_t)r2 * Win + cl], cr_ = xp[(size_t)r2 * Win + cr];
    if constexpr (ZB) {
        if (c0 + 4 >= Win) ar = br = cr_ = 0.f;          // (the left edge: see hrow)
    }
    // horizontal taps: output 2c = l0 x[c-1] + l1 x[c] (c = 0: 1 x[0] + 0 x[1]); output 2c+1 = .75 x[c] + .25 x[min(c+1, Win-1)]
    const float e0 = c0 == 0 ? 1.f : 0.25f, e1 = c0 == 0 ? 0.f : 0.75f;
    auto hrow = [&](const float4& v, float l, float r, float (&o)[8]) {
        // (c0 == 0: the first even output is 1 * x[0] + 0 * x[1], exactly as upsample2x_kernel forms it)
        o[0] = c0 == 0 ? (ZB ? 0.75f * v.x : e0 * v.x + e1 * v.y) : 0.25f * l + 0.75f * v.x;
        o[1] = 0.75f * v.x + 0.25f * v.y;
        o[2] = 0.25f * v.x + 0.75f * v.y;
        o[3] = 0.75f * v.y + 0.25f * v.z;
        o[4] = 0.25f * v.y + 0.75f * v.z;
        o[5] = 0.75f * v.z + 0.25f * v.w;
        o[6] = 0.25f * v.z + 0.75f * v.w;
        o[7] = 0.75f * v.w + 0.25f * r;
    };
    float ha[8], hb[8], hc[8];
    hrow(a, al, ar, ha);
    hrow(b, bl, br, hb);
    hrow(c, cl_, cr_, hc);
    float* yp = y + (size_t)blockIdx.y * 4 * Hin * Win + (size_t)(2 * m) * (2 * Win) + 2 * c0;
    float ev[8], od[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // row 2m = .25 H[m-1] + .75 H[m] (m = 0: 1 H[0] + 0 H[1]); row 2m+1 = .75 H[m] + .25 H[min(m+1, Hin-1)]
        ev[k] = m == 0 ? (ZB ? 0.75f * hb[k] : 1.f * hb[k] + 0.f * hc[k]) : 0.25f * ha[k] + 0.75f * hb[k];
        od[k] = (ZB && m == Hin - 1) ? 0.75f * hb[k] : 0.75f * hb[k] + 0.25f * hc[k];
    }
    *reinterpret_cast<float4*>(yp) = make_float4(ev[0], ev[1], ev[2], ev[3]);
    *reinterpret_cast<float4*>(yp + 4) = make_float4(ev[4], ev[5], ev[6], ev[7]);
    *reinterpret_cast<float4*>(yp + 2 * Win) = make_float4(od[0], od[1], od[2], od[3]);
    *reinterpret_cast<float4*>(yp + 2 * Win + 4) = make_float4(od[4], od[5], od[6], od[7]);
}

// The same image with FULLY contiguous stores: a thread owns TWO input columns = one 16-byte group of each of its two output rows, so a
// store instruction of the wave writes 1 KB without gaps (the 4-column form above writes 32 bytes per thread as two 16-byte halves: every
// instruction touches every other 16 bytes -- 3.8 TB/s of the ~7 TB/s a plain fill reaches).  Same taps, same expressions, same bits.
template <bool ZB>
__global__ __launch_bounds__(256) void upsample2x_vec2_kernel(const float* __restrict__ x, float* __restrict__ y, int Hin, int Win) {
    const int Wq = Win >> 1;
    const int t_raw = blockIdx.x * 256 + threadIdx.x;             // over Hin * Wq
    const bool live = t_raw < Hin * Wq;
    const int t = live ? t_raw : Hin * Wq - 1;                    // (idle lanes of the last wave shadow the last element: they take part in the shuffles)
    const int m = t / Wq, qb = t - m * Wq, c0 = 2 * qb, lane = threadIdx.x & 63;
    const float* xp = x + (size_t)blockIdx.y * Hin * Win;
    const int r0 = max(m - 1, 0), r2 = min(m + 1, Hin - 1), cl = max(c0 - 1, 0), cr = min(c0 + 2, Win - 1);
    const float2 a = *reinterpret_cast<const float2*>(xp + (size_t)r0 * Win + c0);
    const float2 b = *reinterpret_cast<const float2*>(xp + (size_t)m * Win + c0);
    const float2 c = *reinterpret_cast<const float2*>(xp + (size_t)r2 * Win + c0);
    // the column left / right of the pair: the neighbour lane's value (three 8-byte loads per thread instead of nine loads); the
    // lanes at a row's or the wave's edge load theirs
    float al = __shfl_up(a.y, 1), bl = __shfl_up(b.y, 1), cl_ = __shfl_up(c.y, 1);
    float ar = __shfl_down(a.x, 1), br = __shfl_down(b.x, 1), cr_ = __shfl_down(c.x, 1);
    if (qb == 0 || lane == 0) {
        al = xp[(size_t)r0 * Win + cl]; bl = xp[(size_t)m * Win + cl]; cl_ = xp[(size_t)r2 * Win + cl];
    }
    if (qb == Wq - 1 || lane == 63) {
        ar = xp[(size_t)r0 * Win + cr]; br = xp[(size_t)m * Win + cr]; cr_ = xp[(size_t)r2 * Win + cr];
    }
    if constexpr (ZB) {
        if (c0 + 2 >= Win) ar = br = cr_ = 0.f;
    }
    const float e0 = c0 == 0 ? 1.f : 0.25f, e1 = c0 == 0 ? 0.f : 0.75f;
    auto hrow = [&](const float2& v, float l, float r, float (&o)[4]) {
        o[0] = c0 == 0 ? (ZB ? 0.75f * v.x : e0 * v.x + e1 * v.y) : 0.25f * l + 0.75f * v.x;
        o[1] = 0.75f * v.x + 0.25f * v.y;
        o[2] = 0.25f * v.x + 0.75f * v.y;
        o[3] = 0.75f * v.y + 0.25f * r;
    };
    float ha[4], hb[4], hc[4];
    hrow(a, al, ar, ha);
    hrow(b, bl, br, hb);
    hrow(c, cl_, cr_, hc);
    float* yp = y + (size_t)blockIdx.y * 4 * Hin * Win + (size_t)(2 * m) * (2 * Win) + 2 * c0;
    float ev[4], od[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ev[k] = m == 0 ? (ZB ? 0.75f * hb[k] : 1.f * hb[k] + 0.f * hc[k]) : 0.25f * ha[k] + 0.75f * hb[k];
        od[k] = (ZB && m == Hin - 1) ? 0.75f * hb[k] : 0.75f * hb[k] + 0.25f * hc[k];
    }
    if (!live) return;
    *reinterpret_cast<float4*>(yp) = make_float4(ev[0], ev[1], ev[2], ev[3]);
    *reinterpret_cast<float4*>(yp + 2 * Win) = make_float4(od[0], od[1], od[2], od[3]);
}

// ... and TWO input rows per thread (Hin even): four row loads feed four output rows (1 load per 16-byte store instead of 1.5).
template <bool ZB>
__global__ __launch_bounds__(256) void upsample2x_vec2x2_kernel(const float* __restrict__ x, float* __restrict__ y, int Hin, int Win) {
    const int Wq = Win >> 1, Hq = Hin >> 1;
    const int t_raw = blockIdx.x * 256 + threadIdx.x;             // over Hq * Wq
    const bool live = t_raw < Hq * Wq;
    const int t = live ? t_raw : Hq * Wq - 1;
    const int mp = t / Wq, qb = t - mp * Wq, c0 = 2 * qb, lane = threadIdx.x & 63, m0 = 2 * mp;
    const float* xp = x + (size_t)blockIdx.y * Hin * Win;
    const int cl = max(c0 - 1, 0), cr = min(c0 + 2, Win - 1);
    const int rows[4] = {max(m0 - 1, 0), m0, m0 + 1, min(m0 + 2, Hin - 1)};
    float2 v[4];
    float vl[4], vr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float2*>(xp + (size_t)rows[k] * Win + c0);
#pragma unroll
    for (int k = 0; k < 4; ++k) { vl[k] = __shfl_up(v[k].y, 1); vr[k] = __shfl_down(v[k].x, 1); }
    if (qb == 0 || lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) vl[k] = xp[(size_t)rows[k] * Win + cl];
    }
    if (qb == Wq - 1 || lane == 63) {
#pragma unroll
        for (int k = 0; k < 4; ++k) vr[k] = xp[(size_t)rows[k] * Win + cr];
    }
    if constexpr (ZB) {
        if (c0 + 2 >= Win) { vr[0] = vr[1] = vr[2] = vr[3] = 0.f; }
    }
    const float e0 = c0 == 0 ? 1.f : 0.25f, e1 = c0 == 0 ? 0.f : 0.75f;
    float h[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k][0] = c0 == 0 ? (ZB ? 0.75f * v[k].x : e0 * v[k].x + e1 * v[k].y) : 0.25f * vl[k] + 0.75f * v[k].x;
        h[k][1] = 0.75f * v[k].x + 0.25f * v[k].y;
        h[k][2] = 0.25f * v[k].x + 0.75f * v[k].y;
        h[k][3] = 0.75f * v[k].y + 0.25f * vr[k];
    }
    if (!live) return;
    float* yp = y + (size_t)blockIdx.y * 4 * Hin * Win + (size_t)(2 * m0) * (2 * Win) + 2 * c0;
    float o[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // input row m0 (h[1], above h[0], below h[2]) and row m0 + 1 (h[2], above h[1], below h[3])
        o[0][k] = m0 == 0 ? (ZB ? 0.75f * h[1][k] : 1.f * h[1][k] + 0.f * h[2][k]) : 0.25f * h[0][k] + 0.75f * h[1][k];
        o[1][k] = 0.75f * h[1][k] + 0.25f * h[2][k];
        o[2][k] = 0.25f * h[1][k] + 0.75f * h[2][k];
        o[3][k] = (ZB && m0 + 1 == Hin - 1) ? 0.75f * h[2][k] : 0.75f * h[2][k] + 0.25f * h[3][k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(yp + (size_t)r * 2 * Win) = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
}

// 1x1 conv FROM <= 4 channels (the discriminator's fromRGB, styleganv1.py:675: 3 -> 64 at 256^2): a pure store stream -- 134 MB of
// output for 6 MB of input at B = 8 -- that the tap kernel ran at 1.8 TB/s (its tiles are built for long contractions).  A thread owns
// four pixels: C 16-byte loads, then one fully contiguous 16-byte store per output channel (bias, LeakyReLU fused); `scale_dev`: an
// optional device scalar on the weights (1 / sigma of a spectrally normalised layer).
__global__ __launch_bounds__(256) void conv1x1_expand_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ scale_dev, float* __restrict__ y, int C, int O,
                                                            long long HW, float slope) {
    extern __shared__ float w_s[];                       // [O][4] weights (zero padded to 4 inputs) + [O] bias
    const float sc = scale_dev ? *scale_dev : 1.f;
    for (int i = threadIdx.x; i < O * 4; i += blockDim.x) w_s[i] = (i & 3) < C ? w[(i >> 2) * C + (i & 3)] * sc : 0.f;
    for (int i = threadIdx.x; i < O; i += blockDim.x) w_s[4 * O + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int b = blockIdx.y;
    const long long p4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p4 >= HW / 4) return;
    float4 xv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xv[c] = c < C ? reinterpret_cast<const float4*>(x + ((size_t)b * C + c) * HW)[p4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4* yb = reinterpret_cast<float4*>(y + (size_t)b * O * HW) + p4;
    const size_t plane4 = (size_t)(HW / 4);
#pragma unroll 4
    for (int o = 0; o < O; ++o) {
        const float4 wv = *reinterpret_cast<const float4*>(w_s + 4 * o);
        const float bo = w_s[4 * O + o];
        float4 v;
        v.x = bo + wv.x * xv[0].x + wv.y * xv[1].x + wv.z * xv[2].x + wv.w * xv[3].x;
        v.y = bo + wv.x * xv[0].y + wv.y * xv[1].y + wv.z * xv[2].y + wv.w * xv[3].y;
        v.z = bo + wv.x * xv[0].z + wv.y * xv[1].z + wv.z * xv[2].z + wv.w * xv[3].z;
        v.w = bo + wv.x * xv[0].w + wv.y * xv[1].w + wv.z * xv[2].w + wv.w * xv[3].w;
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        yb[(size_t)o * plane4] = v;
    }
}

inline unsigned stream_grid(long long work_items, int threads) {
    const long long blocks = (work_items + threads - 1) / threads;
    return (unsigned)std::max(1ll, std::min(blocks, 256ll * 8));
}

}  // namespace

extern "C" {

int spk_bias_noise_style_fwd(const float* x, int64_t x_batch_stride, const float* bias, const float* noise_w,
                             const float* noise, const float* style, int64_t style_stride, float* y, int B, int C,
                             int HW, void* stream) {
    SPK_REQUIRE(x && y, "bias_noise_style: null pointer");
    SPK_REQUIRE(B > 0 && C > 0 && HW > 0, "bias_noise_style: bad shape");
    SPK_REQUIRE(!noise || noise_w, "bias_noise_style: noise without noise_w");
    hipLaunchKernelGGL(bias_noise_style_kernel, dim3(stream_grid((long long)B * C * HW, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, (long long)x_batch_stride, bias, noise_w, noise, style,
                       (long long)style_stride, y, B, C, HW);
    return spk::check_launch("bias_noise_style_kernel");
}

static int conv1x1_small_launch(const float* x, const float* w, const float* mod, const float* bias, float* y, int B, int C, int O,
                                int64_t HW, float in_scale, void* stream, const float* skip = nullptr, int W = 0) {
    SPK_REQUIRE(x && w && y, "conv1x1_small: null pointer");
    SPK_REQUIRE(B > 0 && C > 0 && O > 0 && O <= 4 && HW > 0, "conv1x1_small: bad shape (O must be <= 4)");
    SPK_REQUIRE((size_t)O * C * sizeof(float) <= 48 * 1024, "conv1x1_small: weight too large for LDS");
    SPK_REQUIRE(!skip || (W > 0 && W % 2 == 0 && HW % W == 0 && (HW / W) % 2 == 0), "conv1x1_small: skip needs an even H x W image");
    // with a skip image a pixel quad must stay inside one row
    const bool vec = (HW % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && (!skip || W % 4 == 0);
    const size_t lds = (size_t)O * C * sizeof(float);
    if (vec && C >= 64 && (HW / 4 + 255) / 256 * B < 512) {   // too few pixel quads for one thread each: split the channels
        const long long n4 = HW / 4;
        const int lgQ = n4 >= 16 ? 4 : (n4 >= 8 ? 3 : (n4 >= 4 ? 2 : (n4 >= 2 ? 1 : 0)));      // Q = min(16, pow2 <= n4 ...)
        const int Q = 1 << lgQ, NS = 256 >> lgQ;
        dim3 grid((unsigned)((n4 + Q - 1) / Q), (unsigned)B);
        const size_t lds2 = ((size_t)((O * C + 3) & ~3) + (size_t)NS * 4 * Q * 4) * sizeof(float);
        hipLaunchKernelGGL(conv1x1_small_csplit_kernel, grid, dim3(256), lds2, (hipStream_t)stream, x, w, bias, y, C, O,
                           (long long)HW, in_scale, mod, skip, W, lgQ);
        return spk::check_launch("conv1x1_small_csplit_kernel");
    }
    if (vec) {
        dim3 grid(stream_grid(HW / 4, 256), (unsigned)B);
        hipLaunchKernelGGL(conv1x1_small_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, y, C, O,
                           (long long)HW, in_scale, mod, skip, W);
    } else {
        dim3 grid(stream_grid(HW, 256), (unsigned)B);
        hipLaunchKernelGGL(conv1x1_small_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, y, C, O,
                           (long long)HW, in_scale, mod, skip, W);
    }
    return spk::check_launch("conv1x1_small_kernel");
}

int spk_conv1x1_expand_fwd(const float* x, const float* w, const float* bias, const float* scale_dev, float* y, int B, int C, int O,
                           int64_t HW, float slope, void* stream) {
    SPK_REQUIRE(x && w && y, "conv1x1_expand: null pointer");
    SPK_REQUIRE(B > 0 && B < 65536 && C > 0 && C <= 4 && O > 0 && O <= 2048 && HW > 0 && HW % 4 == 0, "conv1x1_expand: C <= 4 inputs, HW a multiple of 4");
    SPK_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "conv1x1_expand: 16-byte aligned tensors");
    dim3 grid((unsigned)((HW / 4 + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(conv1x1_expand_kernel, grid, dim3(256), (size_t)O * 5 * sizeof(float), (hipStream_t)stream, x, w, bias, scale_dev, y, C, O,
                       (long long)HW, slope);
    return spk::check_launch("conv1x1_expand_kernel");
}

int spk_conv1x1_small_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int O, int64_t HW,
                          float in_scale, void* stream) {
    return conv1x1_small_launch(x, w, nullptr, bias, y, B, C, O, HW, in_scale, stream);
}

int spk_conv1x1_small_mod_fwd(const float* x, const float* w, const float* mod, const float* bias, float* y, int B, int C, int O,
                              int64_t HW, float in_scale, void* stream) {
    SPK_REQUIRE(mod, "conv1x1_small_mod: null modulation");
    return conv1x1_small_launch(x, w, mod, bias, y, B, C, O, HW, in_scale, stream);
}

int spk_torgb_mod_skip_fwd(const float* x, const float* w, const float* mod, const float* bias, const float* skip, float* y, int B,
                           int C, int O, int H, int W, float in_scale, void* stream) {
    SPK_REQUIRE(mod, "torgb_mod_skip: null modulation");
    SPK_REQUIRE(H > 0 && W > 0, "torgb_mod_skip: bad shape");
    return conv1x1_small_launch(x, w, mod, bias, y, B, C, O, (int64_t)H * W, in_scale, stream, skip, W);
}

int spk_upsample2x_fwd(const float* x, float* y, int64_t planes, int Hin, int Win, int zero_border, void* stream) {
    SPK_REQUIRE(x && y, "upsample2x: null pointer");
    SPK_REQUIRE(planes > 0 && Hin > 0 && Win > 0, "upsample2x: bad shape");
    const bool vec = Win % 4 == 0 && planes < 65536 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (!zero_border && !vec) return spk_upsample2x_bilinear_fwd(x, y, planes, Hin, Win, stream);
    SPK_REQUIRE(vec, "upsample2x: the zero-border (upfirdn2d [1,3,3,1]) form needs Win %% 4 == 0, 16-byte aligned tensors, < 65536 planes");
    static const int form = [] { const char* e = getenv("SPK_UPSAMPLE_FORM"); return e ? atoi(e) : 22; }();    // lab: 4 = the 4-column threads, 2 = two columns x one row
    if (form == 22 && Hin % 2 == 0) {
        dim3 grid3((unsigned)(((Hin >> 1) * (Win >> 1) + 255) / 256), (unsigned)planes);
        if (zero_border) hipLaunchKernelGGL(upsample2x_vec2x2_kernel<true>, grid3, dim3(256), 0, (hipStream_t)stream, x, y, Hin, Win);
        else hipLaunchKernelGGL(upsample2x_vec2x2_kernel<false>, grid3, dim3(256), 0, (hipStream_t)stream, x, y, Hin, Win);
        return spk::check_launch("upsample2x_vec2x2_kernel");
    }
    if (form == 2 || form == 22) {
        dim3 grid2((unsigned)((Hin * (Win >> 1) + 255) / 256), (unsigned)planes);
        if (zero_border) hipLaunchKernelGGL(upsample2x_vec2_kernel<true>, grid2, dim3(256), 0, (hipStream_t)stream, x, y, Hin, Win);
        else hipLaunchKernelGGL(upsample2x_vec2_kernel<false>, grid2, dim3(256), 0, (hipStream_t)stream, x, y, Hin, Win);
        return spk::check_launch("upsample2x_vec2_kernel");
    }
    dim3 grid((unsigned)((Hin * (Win >> 2) + 255) / 256), (unsigned)planes);
    if (zero_border) hipLaunchKernelGGL(upsample2x_vec_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, y, Hin, Win);
    else hipLaunchKernelGGL(upsample2x_vec_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, y, Hin, Win);
    return spk::check_launch("upsample2x_vec_kernel");
}

int spk_upsample2x_bilinear_fwd(const float* x, float* y, int64_t planes, int Hin, int Win, void* stream) {
    SPK_REQUIRE(x && y, "upsample2x: null pointer");
    SPK_REQUIRE(planes > 0 && Hin > 0 && Win > 0, "upsample2x: bad shape");
    if (Win % 4 == 0 && planes < 65536 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
        return spk_upsample2x_fwd(x, y, planes, Hin, Win, 0, stream);      // (one kernel for both entry points: the same bits)
    hipLaunchKernelGGL(upsample2x_kernel, dim3(stream_grid(planes * 2ll * Hin * ((2 * Win + 3) / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, (long long)planes, Hin, Win);
    return spk::check_launch("upsample2x_kernel");
}

const char* spk_version(void) { return "spk-hip 0.1 (gfx950)"; }
const char* spk_last_error(void) { return spk::err_buf(); }

}  // extern "C"
