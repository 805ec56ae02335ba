// Backward pieces of the decoder that are not convolutions: the fused-epilogue adjoint, the bilinear
// x2 adjoint, toRGB backward, FC backward.  All HBM/L2-bound single passes.  The conv data gradient
// reuses the forward MFMA kernel (transpose_flip packing), the weight gradient is wgrad_mfma_f32.hip.
#include "spk_common.hpp"

#include <algorithm>

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// One workgroup per (b,c) plane.  y = a*(s0+1)+s1, a = lrelu(t), t = conv + bias + nw*noise:
//   dt = dy*(s0+1)*(a>0 ? 1 : slope);  sums = {sum dy, sum dy*a, sum dt, sum dt*noise}
__global__ __launch_bounds__(256) void epilogue_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                          const float* __restrict__ noise, const float* __restrict__ style,
                                                          long long style_stride, float slope, float* __restrict__ dt,
                                                          float* __restrict__ sums, int C, long long HW) {
    __shared__ float red[4];
    const long long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const long long b = plane / C;
    const float g = style ? style[b * style_stride + c] + 1.f : 1.f;
    const float* dyp = dy + plane * HW;
    const float* ap = a ? a + plane * HW : nullptr;
    const float* np = noise ? noise + b * HW : nullptr;
    float* dtp = dt + plane * HW;
    float s_dy = 0.f, s_dya = 0.f, s_dt = 0.f, s_dtn = 0.f;
    const bool vec = (HW % 4 == 0) && (((uintptr_t)dy | (uintptr_t)dt | (uintptr_t)(a ? a : dy) | (uintptr_t)(noise ? noise : dy)) % 16 == 0);
    if (vec) {
        const long long n4 = HW / 4;
#pragma unroll 2
        for (long long i = threadIdx.x; i < n4; i += 256) {
            const float4 d = reinterpret_cast<const float4*>(dyp)[i];
            const float4 av = ap ? reinterpret_cast<const float4*>(ap)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 nv = np ? reinterpret_cast<const float4*>(np)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 t;
            t.x = d.x * g * ((!ap || av.x > 0.f) ? 1.f : slope);
            t.y = d.y * g * ((!ap || av.y > 0.f) ? 1.f : slope);
            t.z = d.z * g * ((!ap || av.z > 0.f) ? 1.f : slope);
            t.w = d.w * g * ((!ap || av.w > 0.f) ? 1.f : slope);
            reinterpret_cast<float4*>(dtp)[i] = t;
            s_dy += (d.x + d.y) + (d.z + d.w);
            s_dya += (d.x * av.x + d.y * av.y) + (d.z * av.z + d.w * av.w);
            s_dt += (t.x + t.y) + (t.z + t.w);
            s_dtn += (t.x * nv.x + t.y * nv.y) + (t.z * nv.z + t.w * nv.w);
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += 256) {
            const float d = dyp[i];
            const float av = ap ? ap[i] : 0.f;
            const float t = d * g * ((!ap || av > 0.f) ? 1.f : slope);
            dtp[i] = t;
            s_dy += d;
            s_dya += d * av;
            s_dt += t;
            if (np) s_dtn += t * np[i];
        }
    }
    s_dy = block_sum(s_dy, red);
    s_dya = block_sum(s_dya, red);
    s_dt = block_sum(s_dt, red);
    s_dtn = block_sum(s_dtn, red);
    if (threadIdx.x == 0) {
        // [B][4][C], component order {dy*a, dy, dt, dt*noise}: rows 0-1 of an image ARE its style gradient [d s0 | d s1]
        float* o = sums + (size_t)b * 4 * C + c;
        o[0] = s_dya; o[C] = s_dy; o[2 * (size_t)C] = s_dt; o[3 * (size_t)C] = s_dtn;
    }
}

// out[c] (+)= sum_b sums[b][row][c]: fixed order over the batch
__global__ __launch_bounds__(256) void plane_sums_reduce_kernel(const float* __restrict__ sums, int B, int rows, int C, int row,
                                                               float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sums[((size_t)b * rows + row) * C + c];
    out[c] = accumulate ? out[c] + s : s;
}

// adjoint of bilinear x2 (align_corners=False): every source pixel gathers from the <= 4x4 upsampled
// pixels whose forward stencil touches it, re-evaluating the forward index/lambda rule (edges included).
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                            long long planes, int Hin, int Win) {
    const int Ho = 2 * Hin, Wo = 2 * Win;
    const long long total = planes * Hin * Win;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(idx % Win), iy = (int)((idx / Win) % Hin);
        const long long pl = idx / ((long long)Win * Hin);
        const float* g = dy + pl * Ho * Wo;
        // rows 2iy-1 .. 2iy+2 (cols likewise) are the only ones whose forward stencil can touch (iy, ix)
        float wy[4], wx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int uy = 2 * iy - 1 + k, ux = 2 * ix - 1 + k;
            wy[k] = 0.f; wx[k] = 0.f;
            if (uy >= 0 && uy < Ho) {
                const int y0 = uy == 0 ? 0 : (uy - 1) >> 1, y1 = min(y0 + 1, Hin - 1);
                const float l1 = uy == 0 ? 0.f : ((uy & 1) ? 0.25f : 0.75f);
                wy[k] = (y0 == iy ? 1.f - l1 : 0.f) + (y1 == iy ? l1 : 0.f);
            }
            if (ux >= 0 && ux < Wo) {
                const int x0 = ux == 0 ? 0 : (ux - 1) >> 1, x1 = min(x0 + 1, Win - 1);
                const float l1 = ux == 0 ? 0.f : ((ux & 1) ? 0.25f : 0.75f);
                wx[k] = (x0 == ix ? 1.f - l1 : 0.f) + (x1 == ix ? l1 : 0.f);
            }
        }
        // all 16 loads are issued before the first use: rows / columns outside the image are clamped to a valid address and
        // carry weight 0.  (With a branch per zero weight the loads of a pixel ran one memory latency after the other:
        // 476 us for the 256^2 layer's 670 MB, 1.4 TB/s.)
        float v[4][4];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const float* row = g + (size_t)min(max(2 * iy - 1 + ky, 0), Ho - 1) * Wo;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v[ky][kx] = row[min(max(2 * ix - 1 + kx, 0), Wo - 1)];
        }
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            float r = 0.f;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) r += wx[kx] * v[ky][kx];
            acc += wy[ky] * r;
        }
        dx[idx] = acc;
    }
}

// The same adjoint, FOUR source pixels of a row per thread (Win % 4 == 0, 16-byte aligned planes): per upsampled row two 16-byte
// loads + the two neighbours instead of 16 stride-2 dwords per pixel (4 loads per output instead of 16; the 256^2 layer's 670 MB
// took 320 us = 2.1 TB/s with the scalar form).  Tap k of source index i (upsampled index 2 i - 1 + k) weighs
// {i > 0 ? .25 : 0, i > 0 ? .75 : 1, i < n - 1 ? .75 : 1, i < n - 1 ? .25 : 0}: the forward's clamped taps at the borders.
__global__ __launch_bounds__(256) void upsample2x_bwd_vec_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                long long planes, int Hin, int Win) {
    const int Ho = 2 * Hin, Wo = 2 * Win, W4 = Win >> 2;
    const long long total = planes * Hin * W4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(idx % W4) * 4, iy = (int)((idx / W4) % Hin);
        const long long pl = idx / ((long long)W4 * Hin);
        const float* g = dy + pl * Ho * Wo;
        float4 a[4], b[4];
        float l[4], r[4];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {             // every load before the first use; rows / neighbours outside carry weight 0
            const float* row = g + (size_t)min(max(2 * iy - 1 + ky, 0), Ho - 1) * Wo + 2 * ix;
            a[ky] = *reinterpret_cast<const float4*>(row);
            b[ky] = *reinterpret_cast<const float4*>(row + 4);
            l[ky] = row[ix > 0 ? -1 : 0];
            r[ky] = row[ix + 4 < Win ? 8 : 7];
        }
        const float wy[4] = {iy > 0 ? 0.25f : 0.f, iy > 0 ? 0.75f : 1.f, iy < Hin - 1 ? 0.75f : 1.f, iy < Hin - 1 ? 0.25f : 0.f};
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const float e[10] = {l[ky], a[ky].x, a[ky].y, a[ky].z, a[ky].w, b[ky].x, b[ky].y, b[ky].z, b[ky].w, r[ky]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = ix + j;
                const float h = (i > 0 ? 0.25f : 0.f) * e[2 * j] + (i > 0 ? 0.75f : 1.f) * e[2 * j + 1] +
                                (i < Win - 1 ? 0.75f : 1.f) * e[2 * j + 2] + (i < Win - 1 ? 0.25f : 0.f) * e[2 * j + 3];
                acc[j] += wy[ky] * h;
            }
        }
        *reinterpret_cast<float4*>(dx + (pl * Hin + iy) * (long long)Win + ix) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ... and TWO source rows per thread (Hin even): six upsampled rows feed two output rows (3 row reads per output row instead of 4),
// the columns left / right of a thread's eight by lane shuffle (twelve 16-byte loads per thread and no dword gathers: the form above
// issues sixteen loads per output row and ran the 256^2 layer's 335 MB at 1.9 TB/s).
__global__ __launch_bounds__(256) void upsample2x_bwd_vec2_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                 long long planes, int Hin, int Win) {
    const int Ho = 2 * Hin, Wo = 2 * Win, W4 = Win >> 2, H2 = Hin >> 1, lane = threadIdx.x & 63;
    const long long total = planes * H2 * W4;
    const long long rounds = (total + (long long)gridDim.x * blockDim.x - 1) / ((long long)gridDim.x * blockDim.x);
    for (long long it = 0; it < rounds; ++it) {          // (every lane runs every round: the shuffles below are wave-wide)
        const long long idx_raw = it * (long long)gridDim.x * blockDim.x + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = idx_raw < total;
        const long long idx = live ? idx_raw : total - 1;
        const int q = (int)(idx % W4), ix = q * 4, iy = (int)((idx / W4) % H2) * 2;
        const long long pl = idx / ((long long)W4 * H2);
        const float* g = dy + pl * Ho * Wo;
        float4 a[6], b[6];
        float l[6], r[6];
#pragma unroll
        for (int ky = 0; ky < 6; ++ky) {
            const float* row = g + (size_t)min(max(2 * iy - 1 + ky, 0), Ho - 1) * Wo + 2 * ix;
            a[ky] = *reinterpret_cast<const float4*>(row);
            b[ky] = *reinterpret_cast<const float4*>(row + 4);
        }
#pragma unroll
        for (int ky = 0; ky < 6; ++ky) { l[ky] = __shfl_up(b[ky].w, 1); r[ky] = __shfl_down(a[ky].x, 1); }
        if (q == 0 || lane == 0) {                       // (a row's first thread: weight 0; a wave's first lane: its own load)
#pragma unroll
            for (int ky = 0; ky < 6; ++ky) l[ky] = (g + (size_t)min(max(2 * iy - 1 + ky, 0), Ho - 1) * Wo + 2 * ix)[ix > 0 ? -1 : 0];
        }
        if (q == W4 - 1 || lane == 63) {
#pragma unroll
            for (int ky = 0; ky < 6; ++ky) r[ky] = (g + (size_t)min(max(2 * iy - 1 + ky, 0), Ho - 1) * Wo + 2 * ix)[ix + 4 < Win ? 8 : 7];
        }
        float h[6][4];
#pragma unroll
        for (int ky = 0; ky < 6; ++ky) {
            const float e[10] = {l[ky], a[ky].x, a[ky].y, a[ky].z, a[ky].w, b[ky].x, b[ky].y, b[ky].z, b[ky].w, r[ky]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = ix + j;
                h[ky][j] = (i > 0 ? 0.25f : 0.f) * e[2 * j] + (i > 0 ? 0.75f : 1.f) * e[2 * j + 1] +
                           (i < Win - 1 ? 0.75f : 1.f) * e[2 * j + 2] + (i < Win - 1 ? 0.25f : 0.f) * e[2 * j + 3];
            }
        }
        if (!live) continue;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int y = iy + rr;
            const float wy[4] = {y > 0 ? 0.25f : 0.f, y > 0 ? 0.75f : 1.f, y < Hin - 1 ? 0.75f : 1.f, y < Hin - 1 ? 0.25f : 0.f};
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += wy[ky] * h[2 * rr + ky][j];
            *reinterpret_cast<float4*>(dx + (pl * Hin + y) * (long long)Win + ix) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
}

// toRGB backward, data half: dx[b,c,p] = in_scale * sum_o w[o,c] * dy[b,o,p]   (one streaming pass, 16-B accesses)
template <bool VEC>
__global__ __launch_bounds__(256) void conv1x1_small_bwd_data_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                                    float* __restrict__ dx, int C, int O, long long HW,
                                                                    float in_scale) {
    extern __shared__ float w_s[];  // [O][C]
    for (int i = threadIdx.x; i < O * C; i += blockDim.x) w_s[i] = w[i] * in_scale;
    __syncthreads();
    const int b = blockIdx.y;
    const float* dyb = dy + (size_t)b * O * HW;
    float* dxb = dx + (size_t)b * C * HW;
    if (VEC) {
        const long long n4 = HW / 4;
        for (long long p4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; p4 < n4; p4 += (long long)gridDim.x * blockDim.x) {
            float4 g[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) g[o] = o < O ? reinterpret_cast<const float4*>(dyb + (size_t)o * HW)[p4] : make_float4(0, 0, 0, 0);
            for (int c = 0; c < C; ++c) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < O) {
                        const float wv = w_s[o * C + c];
                        v.x += wv * g[o].x; v.y += wv * g[o].y; v.z += wv * g[o].z; v.w += wv * g[o].w;
                    }
                reinterpret_cast<float4*>(dxb + (size_t)c * HW)[p4] = v;
            }
        }
    } else {
        for (long long pp = (long long)blockIdx.x * blockDim.x + threadIdx.x; pp < HW; pp += (long long)gridDim.x * blockDim.x) {
            float g[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) g[o] = o < O ? dyb[(size_t)o * HW + pp] : 0.f;
            for (int c = 0; c < C; ++c) {
                float v = 0.f;
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < O) v += w_s[o * C + c] * g[o];
                dxb[(size_t)c * HW + pp] = v;
            }
        }
    }
}

// toRGB backward, weight half.  A workgroup walks a pixel range of one image in 128-pixel tiles for ONE 64-channel group
// (blockIdx.z; round 2 looped the groups inside the workgroup, which left the 512-channel 4^2..32^2 toRGB layers of the
// StyleGAN2 variant on 8 workgroups: 300 us each); the x tile is staged through LDS (coalesced HBM reads, odd row pitch) and
// thread (c, q) accumulates sum_p dy[o,p]*x[c,p] over its quarter of the tile.  partial[blk][o*C + c], partial[blk][O*C + o]
// (d bias, written by the first channel group).
__global__ __launch_bounds__(256) void conv1x1_small_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                      float* __restrict__ partial, int C, int O, long long HW,
                                                                      long long px_per_block, float in_scale) {
    __shared__ float x_s[64 * 129];
    __shared__ float dy_s[4 * 128];
    __shared__ float red[4];
    const int b = blockIdx.y, tid = threadIdx.x, c_l = tid & 63, q = tid >> 6;
    const long long p_begin = (long long)blockIdx.x * px_per_block, p_end = min(HW, p_begin + px_per_block);
    const float* xb = x + (size_t)b * C * HW;
    const float* dyb = dy + (size_t)b * O * HW;
    float* part = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (O * C + O);
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec_ok = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;       // (p_begin is a multiple of 128)
    {
        const int cg = blockIdx.z * 64;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (long long p0 = p_begin; p0 < p_end; p0 += 128) {
            __syncthreads();
            if (vec_ok && cg + 64 <= C && p0 + 128 <= p_end) {
                // a whole tile: eight 16-byte loads per thread, all in flight before the first LDS store (the dword loop below
                // waits for every load before it issues the next: 32 memory latencies per tile, 209 us for the 256^2 toRGB)
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    v[k] = *reinterpret_cast<const float4*>(xb + (size_t)(cg + (tid >> 5) + 8 * k) * HW + p0 + 4 * (tid & 31));
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float* d = x_s + ((tid >> 5) + 8 * k) * 129 + 4 * (tid & 31);
                    d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
                }
            } else {
                for (int e = tid; e < 64 * 128; e += 256) {
                    const int c = e >> 7, pp = e & 127;
                    x_s[c * 129 + pp] = (cg + c < C && p0 + pp < p_end) ? xb[(size_t)(cg + c) * HW + p0 + pp] : 0.f;
                }
            }
            for (int o = 0; o < O; ++o) {
                const float g = (tid < 128 && p0 + tid < p_end) ? dyb[(size_t)o * HW + p0 + tid] : 0.f;
                if (tid < 128) dy_s[o * 128 + tid] = g;
                if (cg == 0) dbacc[o] += g;
            }
            __syncthreads();
            const float* xr = x_s + c_l * 129 + q * 32;
#pragma unroll 8
            for (int pp = 0; pp < 32; ++pp) {
                const float xv = xr[pp];
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < O) acc[o] += dy_s[o * 128 + q * 32 + pp] * xv;
            }
        }
        // reduce the four pixel quarters (waves) of each channel through LDS
        __syncthreads();
        for (int o = 0; o < O; ++o) x_s[(o * 4 + q) * 64 + c_l] = acc[o];
        __syncthreads();
        if (q == 0 && cg + c_l < C)
            for (int o = 0; o < O; ++o)
                part[o * C + cg + c_l] = (x_s[(o * 4 + 0) * 64 + c_l] + x_s[(o * 4 + 1) * 64 + c_l] + x_s[(o * 4 + 2) * 64 + c_l] +
                                          x_s[(o * 4 + 3) * 64 + c_l]) * in_scale;
    }
    if (blockIdx.z != 0) return;
    for (int o = 0; o < O; ++o) {
        const float t = block_sum(dbacc[o], red);
        if (tid == 0) part[O * C + o] = t;
    }
}

// dx[b,i] = wmul * sum_o dz[b,o] * w[o,i],  dz = dout * (out > 0 ? 1 : slope).  A workgroup owns 64 consecutive i
// (one per lane); its 4 waves split the o range; dz for a tile of 8 batch rows sits in LDS (broadcast reads), so
// every weight element is read once per batch tile.
constexpr int FCB_BT = 8;
__global__ __launch_bounds__(256) void fc_bwd_input_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                          const float* __restrict__ w, float* __restrict__ dx,
                                                          long long dx_stride, int B, int I, int O, float wmul, float slope) {
    extern __shared__ float dz_s[];            // [FCB_BT][O]
    __shared__ float red[4][FCB_BT][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 64 + lane;
    for (int b0 = 0; b0 < B; b0 += FCB_BT) {
        __syncthreads();
        for (int e = tid; e < FCB_BT * O; e += 256) {
            const int b = e / O, o = e - b * O;
            float v = 0.f;
            if (b0 + b < B) {
                const size_t idx = (size_t)(b0 + b) * O + o;
                v = dout[idx] * (out[idx] > 0.f ? 1.f : slope);
            }
            dz_s[e] = v;
        }
        __syncthreads();
        float acc[FCB_BT];
#pragma unroll
        for (int b = 0; b < FCB_BT; ++b) acc[b] = 0.f;
        if (i < I) {
            int o = wave;
            // 32 independent row loads in flight (8 waves' worth of CUs run this kernel: it is latency, and with 8 loads per
            // round a 512 x 512 layer took 16 rounds = 33 us)
            for (; o + 124 < O; o += 128) {
                float wv[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) wv[u] = w[(size_t)(o + 4 * u) * I + i];
#pragma unroll
                for (int u = 0; u < 32; ++u)
#pragma unroll
                    for (int b = 0; b < FCB_BT; ++b) acc[b] += dz_s[b * O + o + 4 * u] * wv[u];
            }
            for (; o + 28 < O; o += 32) {            // 8 independent row loads in flight
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(o + 4 * u) * I + i];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int b = 0; b < FCB_BT; ++b) acc[b] += dz_s[b * O + o + 4 * u] * wv[u];
            }
            for (; o < O; o += 4) {
                const float wv = w[(size_t)o * I + i];
#pragma unroll
                for (int b = 0; b < FCB_BT; ++b) acc[b] += dz_s[b * O + o] * wv;
            }
        }
#pragma unroll
        for (int b = 0; b < FCB_BT; ++b) red[wave][b][lane] = acc[b];
        __syncthreads();
        if (wave == 0 && i < I) {
#pragma unroll
            for (int b = 0; b < FCB_BT; ++b)
                if (b0 + b < B)
                    dx[(size_t)(b0 + b) * dx_stride + i] = (red[0][b][lane] + red[1][b][lane] + red[2][b][lane] + red[3][b][lane]) * wmul;
        }
    }
}

// The same gradient with SIXTEEN columns per workgroup (O % 512 == 0, I % 16 == 0, 16-byte aligned rows): lane (r = lane / 4,
// c = lane % 4) of wave w reads the 16-byte pieces w[128 w' + r + 16 j][i0 + 4 c .. + 3] -- all of them in flight before the first
// use -- so a 512 x 512 layer is 32 workgroups and ONE memory latency instead of 8 workgroups and 16 (42 us per mapping layer
// at the pair decoder's 16 rows: 0.34 ms of a G step).  Fixed summation order: o ascending within a lane, then the 16 lanes of
// a column group by xor-shuffles, then the four waves.
constexpr int FCI_BT = 16;
__global__ __launch_bounds__(256) void fc_bwd_input_cols_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                               const float* __restrict__ w, float* __restrict__ dx,
                                                               long long dx_stride, int B, int I, int O, float wmul, float slope) {
    extern __shared__ float dz_s[];            // [FCI_BT][O]
    __shared__ float red[4][FCI_BT][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 2, c4 = lane & 3;
    const int i0 = blockIdx.x * 16;
    const int o_per_wave = O >> 2;             // rows of this wave: [wave * O/4, + O/4), r + 16 j within
    for (int b0 = 0; b0 < B; b0 += FCI_BT) {
        __syncthreads();
        for (int e = tid; e < FCI_BT * O; e += 256) {
            const int b = e / O, o = e - b * O;
            float v = 0.f;
            if (b0 + b < B) {
                const size_t idx = (size_t)(b0 + b) * O + o;
                v = dout[idx] * (out[idx] > 0.f ? 1.f : slope);
            }
            dz_s[e] = v;
        }
        __syncthreads();
        float4 acc[FCI_BT];
#pragma unroll
        for (int b = 0; b < FCI_BT; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int o0 = wave * o_per_wave; o0 < (wave + 1) * o_per_wave; o0 += 128) {
            float4 wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = *reinterpret_cast<const float4*>(w + (size_t)(o0 + r + 16 * j) * I + i0 + 4 * c4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int b = 0; b < FCI_BT; ++b) {
                    const float z = dz_s[b * O + o0 + r + 16 * j];
                    acc[b].x += z * wv[j].x; acc[b].y += z * wv[j].y; acc[b].z += z * wv[j].z; acc[b].w += z * wv[j].w;
                }
        }
#pragma unroll
        for (int b = 0; b < FCI_BT; ++b) {
#pragma unroll
            for (int off = 4; off <= 32; off <<= 1) {
                acc[b].x += __shfl_xor(acc[b].x, off); acc[b].y += __shfl_xor(acc[b].y, off);
                acc[b].z += __shfl_xor(acc[b].z, off); acc[b].w += __shfl_xor(acc[b].w, off);
            }
            if (lane < 4) {
                red[wave][b][4 * lane] = acc[b].x; red[wave][b][4 * lane + 1] = acc[b].y;
                red[wave][b][4 * lane + 2] = acc[b].z; red[wave][b][4 * lane + 3] = acc[b].w;
            }
        }
        __syncthreads();
        {
            const int b = tid >> 4, c = tid & 15;      // 16 rows x 16 columns
            if (b0 + b < B)
                dx[(size_t)(b0 + b) * dx_stride + i0 + c] = ((red[0][b][c] + red[1][b][c]) + (red[2][b][c] + red[3][b][c])) * wmul;
        }
    }
}

// dw[o,i] = wmul * sum_b dz[b,o] * x[b,i];  db[o] = bmul * sum_b dz[b,o]
__global__ __launch_bounds__(256) void fc_bwd_weight_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                           const float* __restrict__ x, long long x_stride,
                                                           float* __restrict__ dw, float* __restrict__ db, int B, int I,
                                                           int O, float wmul, float bmul, float slope) {
    const int o = blockIdx.y;
    float dzs = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < I; i += gridDim.x * 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float ov = out[(size_t)b * O + o];
            acc += dout[(size_t)b * O + o] * (ov > 0.f ? 1.f : slope) * x[(size_t)b * x_stride + i];
        }
        dw[(size_t)o * I + i] = acc * wmul;
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        for (int b = 0; b < B; ++b) {
            const float ov = out[(size_t)b * O + o];
            dzs += dout[(size_t)b * O + o] * (ov > 0.f ? 1.f : slope);
        }
        db[o] = dzs * bmul;
    }
}

// ---- the same two kernels over up to SPK_FC_MAX_GROUPS independent FCs (the 13 style affines of a training pass:
// one launch each instead of 13 pairs; the groups share B) ----
struct FcBwdGroups { spk_fc_bwd_group g[SPK_FC_MAX_GROUPS]; int n; };

__global__ __launch_bounds__(256) void fc_grouped_bwd_input_kernel(const FcBwdGroups a, int B) {
    extern __shared__ float dz_s[];            // [FCB_BT][O]
    __shared__ float red[4][FCB_BT][64];
    const spk_fc_bwd_group& g = a.g[blockIdx.y];
    const int I = g.I, O = g.O;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 64 + lane;
    if (!g.dx || (int)blockIdx.x * 64 >= I) return;          // uniform
    for (int b0 = 0; b0 < B; b0 += FCB_BT) {
        __syncthreads();
        for (int e = tid; e < FCB_BT * O; e += 256) {
            const int b = e / O, o = e - b * O;
            float v = 0.f;
            if (b0 + b < B) {
                const size_t idx = (size_t)(b0 + b) * O + o;
                v = g.dout[(size_t)(b0 + b) * g.dout_stride + o] * (g.out[idx] > 0.f ? 1.f : g.slope);
            }
            dz_s[e] = v;
        }
        __syncthreads();
        float acc[FCB_BT];
#pragma unroll
        for (int b = 0; b < FCB_BT; ++b) acc[b] = 0.f;
        if (i < I) {
            int o = wave;
            for (; o + 28 < O; o += 32) {            // 8 independent row loads in flight
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = g.w[(size_t)(o + 4 * u) * I + i];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int b = 0; b < FCB_BT; ++b) acc[b] += dz_s[b * O + o + 4 * u] * wv[u];
            }
            for (; o < O; o += 4) {
                const float wv = g.w[(size_t)o * I + i];
#pragma unroll
                for (int b = 0; b < FCB_BT; ++b) acc[b] += dz_s[b * O + o] * wv;
            }
        }
#pragma unroll
        for (int b = 0; b < FCB_BT; ++b) red[wave][b][lane] = acc[b];
        __syncthreads();
        if (wave == 0 && i < I) {
#pragma unroll
            for (int b = 0; b < FCB_BT; ++b)
                if (b0 + b < B)
                    g.dx[(size_t)(b0 + b) * g.dx_stride + i] = (red[0][b][lane] + red[1][b][lane] + red[2][b][lane] + red[3][b][lane]) * g.wmul;
        }
    }
}

__global__ __launch_bounds__(256) void fc_grouped_bwd_weight_kernel(const FcBwdGroups a, int B) {
    const spk_fc_bwd_group& g = a.g[blockIdx.z];
    const int o = blockIdx.y, I = g.I, O = g.O;
    if (!g.dw || o >= O) return;                              // uniform
    for (int i = blockIdx.x * 256 + threadIdx.x; i < I; i += gridDim.x * 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float ov = g.out[(size_t)b * O + o];
            acc += g.dout[(size_t)b * g.dout_stride + o] * (ov > 0.f ? 1.f : g.slope) * g.x[(size_t)b * g.x_stride + i];
        }
        g.dw[(size_t)o * I + i] = acc * g.wmul;
    }
    if (g.db && blockIdx.x == 0 && threadIdx.x == 0) {
        float dzs = 0.f;
        for (int b = 0; b < B; ++b) {
            const float ov = g.out[(size_t)b * O + o];
            dzs += g.dout[(size_t)b * g.dout_stride + o] * (ov > 0.f ? 1.f : g.slope);
        }
        g.db[o] = dzs * g.bmul;
    }
}

}  // namespace

extern "C" {

int spk_plane_sums_reduce(const float* sums, int B, int rows, int C, int row, float* out, int accumulate, void* stream) {
    SPK_REQUIRE(sums && out && B > 0 && rows > 0 && C > 0 && row >= 0 && row < rows, "plane_sums_reduce: bad arguments");
    hipLaunchKernelGGL(plane_sums_reduce_kernel, dim3((unsigned)spk::ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, B, rows, C, row,
                       out, accumulate);
    return spk::check_launch("plane_sums_reduce_kernel");
}

int spk_epilogue_bwd(const float* dy, const float* a, const float* noise, const float* style, int64_t style_stride,
                     float slope, float* dt, float* sums, int B, int C, int64_t HW, void* stream) {
    SPK_REQUIRE(dy && dt && sums && B > 0 && C > 0 && HW > 0, "epilogue_bwd: bad arguments");
    SPK_REQUIRE((long long)B * C < (1ll << 31), "epilogue_bwd: too many planes");
    hipLaunchKernelGGL(epilogue_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, dy, a, noise, style,
                       (long long)style_stride, slope, dt, sums, C, (long long)HW);
    return spk::check_launch("epilogue_bwd_kernel");
}

int spk_upsample2x_bilinear_bwd(const float* dy, float* dx, int64_t planes, int Hin, int Win, void* stream) {
    SPK_REQUIRE(dy && dx && planes > 0 && Hin > 0 && Win > 0, "upsample2x_bwd: bad arguments");
    const long long total = planes * Hin * Win;
    static const int form = [] { const char* e = getenv("SPK_UPSAMPLE_BWD_FORM"); return e ? atoi(e) : 2; }();       // lab: 1 = one source row per thread
    if (form == 2 && Win % 4 == 0 && Hin % 2 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0) {
        hipLaunchKernelGGL(upsample2x_bwd_vec2_kernel, dim3((unsigned)std::min((total / 8 + 255) / 256, 256ll * 256)), dim3(256), 0,
                           (hipStream_t)stream, dy, dx, (long long)planes, Hin, Win);
        return spk::check_launch("upsample2x_bwd_vec2_kernel");
    }
    if (Win % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0) {
        hipLaunchKernelGGL(upsample2x_bwd_vec_kernel, dim3((unsigned)std::min((total / 4 + 255) / 256, 256ll * 256)), dim3(256), 0,
                           (hipStream_t)stream, dy, dx, (long long)planes, Hin, Win);
        return spk::check_launch("upsample2x_bwd_vec_kernel");
    }
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)std::min((total + 255) / 256, 256ll * 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, (long long)planes, Hin, Win);
    return spk::check_launch("upsample2x_bwd_kernel");
}

int spk_conv1x1_small_bwd_blocks(int B, int64_t HW) {
    const long long per_img = std::max(1ll, std::min((long long)(HW + 511) / 512, 128ll));
    return (int)(per_img * B);
}

int spk_conv1x1_small_bwd(const float* x, const float* w, const float* dy, float* dx, float* partial, int B, int C, int O,
                          int64_t HW, float in_scale, void* stream) {
    SPK_REQUIRE(x && w && dy && partial, "conv1x1_small_bwd: null pointer");
    SPK_REQUIRE(B > 0 && C > 0 && O > 0 && O <= 4 && HW > 0, "conv1x1_small_bwd: bad shape (O must be <= 4)");
    SPK_REQUIRE((size_t)O * C * sizeof(float) <= 48 * 1024, "conv1x1_small_bwd: weight too large for LDS");
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        const bool vec = (HW % 4 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0);
        const long long work = vec ? HW / 4 : HW;
        dim3 grid((unsigned)std::max(1ll, std::min((work + 255) / 256, 256ll)), (unsigned)B);
        const size_t lds = (size_t)O * C * sizeof(float);
        if (vec) hipLaunchKernelGGL(conv1x1_small_bwd_data_kernel<true>, grid, dim3(256), lds, s, w, dy, dx, C, O, (long long)HW, in_scale);
        else     hipLaunchKernelGGL(conv1x1_small_bwd_data_kernel<false>, grid, dim3(256), lds, s, w, dy, dx, C, O, (long long)HW, in_scale);
        int rc = spk::check_launch("conv1x1_small_bwd_data_kernel");
        if (rc != SPK_OK) return rc;
    }
    const int per_img = spk_conv1x1_small_bwd_blocks(B, HW) / B;
    const long long px_per_block = ((HW + per_img - 1) / per_img + 127) / 128 * 128;
    hipLaunchKernelGGL(conv1x1_small_bwd_weight_kernel, dim3((unsigned)per_img, (unsigned)B, (unsigned)spk::ceil_div(C, 64)), dim3(256), 0, s, x, dy, partial, C,
                       O, (long long)HW, px_per_block, in_scale);
    return spk::check_launch("conv1x1_small_bwd_weight_kernel");
}

int spk_fc_bwd(const float* dout, const float* out, const float* x, int64_t x_stride, const float* w, float* dx,
               int64_t dx_stride, float* dw, float* db, int B, int I, int O, float wmul, float bmul, float slope,
               void* stream) {
    SPK_REQUIRE(dout && out && x && w, "fc_bwd: null pointer");
    SPK_REQUIRE(B > 0 && I > 0 && O > 0, "fc_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        if (O % 512 == 0 && I % 16 == 0 && (size_t)FCI_BT * O * sizeof(float) <= 48 * 1024 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
            hipLaunchKernelGGL(fc_bwd_input_cols_kernel, dim3((unsigned)(I / 16)), dim3(256), (size_t)FCI_BT * O * sizeof(float), s, dout,
                               out, w, dx, (long long)dx_stride, B, I, O, wmul, slope);
            int rc = spk::check_launch("fc_bwd_input_cols_kernel");
            if (rc != SPK_OK) return rc;
        } else {
            SPK_REQUIRE((size_t)FCB_BT * O * sizeof(float) <= 60 * 1024, "fc_bwd: O too large for the LDS tile");
            hipLaunchKernelGGL(fc_bwd_input_kernel, dim3((unsigned)spk::ceil_div(I, 64)), dim3(256), (size_t)FCB_BT * O * sizeof(float),
                               s, dout, out, w, dx, (long long)dx_stride, B, I, O, wmul, slope);
            int rc = spk::check_launch("fc_bwd_input_kernel");
            if (rc != SPK_OK) return rc;
        }
    }
    if (dw) {
        hipLaunchKernelGGL(fc_bwd_weight_kernel, dim3((unsigned)std::min(spk::ceil_div(I, 256), 8), (unsigned)O), dim3(256), 0,
                           s, dout, out, x, (long long)x_stride, dw, db, B, I, O, wmul, bmul, slope);
        return spk::check_launch("fc_bwd_weight_kernel");
    }
    return SPK_OK;
}

int spk_fc_grouped_bwd(const spk_fc_bwd_group* groups, int n_groups, int B, void* stream) {
    SPK_REQUIRE(groups && n_groups > 0 && n_groups <= SPK_FC_MAX_GROUPS && B > 0, "fc_grouped_bwd: bad arguments (1..%d groups)",
                SPK_FC_MAX_GROUPS);
    FcBwdGroups a;
    a.n = n_groups;
    int maxI = 0, maxO = 0;
    bool any_dx = false, any_dw = false;
    for (int i = 0; i < n_groups; ++i) {
        const spk_fc_bwd_group& g = groups[i];
        SPK_REQUIRE(g.dout && g.out && g.I > 0 && g.O > 0 && g.dout_stride >= g.O, "fc_grouped_bwd: group %d: bad shape", i);
        SPK_REQUIRE(!g.dx || (g.w && g.dx_stride >= g.I), "fc_grouped_bwd: group %d: dx needs w and dx_stride >= I", i);
        SPK_REQUIRE(!g.dw || (g.x && g.x_stride >= g.I), "fc_grouped_bwd: group %d: dw needs x and x_stride >= I", i);
        SPK_REQUIRE(!g.db || g.dw, "fc_grouped_bwd: group %d: db comes with dw", i);
        SPK_REQUIRE((size_t)FCB_BT * g.O * sizeof(float) <= 60 * 1024, "fc_grouped_bwd: group %d: O too large for the LDS tile", i);
        a.g[i] = g;
        maxI = std::max(maxI, g.I); maxO = std::max(maxO, g.O);
        any_dx = any_dx || g.dx; any_dw = any_dw || g.dw;
    }
    hipStream_t s = (hipStream_t)stream;
    if (any_dx) {
        hipLaunchKernelGGL(fc_grouped_bwd_input_kernel, dim3((unsigned)spk::ceil_div(maxI, 64), (unsigned)n_groups), dim3(256),
                           (size_t)FCB_BT * maxO * sizeof(float), s, a, B);
        int rc = spk::check_launch("fc_grouped_bwd_input_kernel");
        if (rc != SPK_OK) return rc;
    }
    if (any_dw) {
        SPK_REQUIRE(maxO < 65536, "fc_grouped_bwd: O too large for the grid");
        hipLaunchKernelGGL(fc_grouped_bwd_weight_kernel, dim3((unsigned)std::min(spk::ceil_div(maxI, 256), 8), (unsigned)maxO, (unsigned)n_groups),
                           dim3(256), 0, s, a, B);
        return spk::check_launch("fc_grouped_bwd_weight_kernel");
    }
    return SPK_OK;
}

}  // extern "C"
