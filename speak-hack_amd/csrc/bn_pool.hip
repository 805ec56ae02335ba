// BatchNorm2d pieces and pooling of the ResNet-50 trunk (torchvision resnet50, built at
// model.py:60-62).  BatchNorm is split MI355X-style: the producing conv accumulates the batch sums
// in its epilogue (SPK_EPI_STATS), bn_finalize turns them into a per-channel affine (and updates the
// running statistics), and the *consumer* applies affine+ReLU while staging its input -- the
// normalised tensor never exists in HBM.  Only the block output (bn3 + identity + ReLU) is
// materialised, by the HBM-bound bn_add_relu pass.
#include "spk_common.hpp"

#include <algorithm>
#include <cstdint>
#include <initializer_list>

namespace {

// nn.BatchNorm2d's training-mode momentum update, spelled with explicit roundings so that every kernel that performs it
// (the finalize of a pass, the replayed second update) rounds identically whatever the compiler would contract
__device__ __forceinline__ float bn_momentum_update(float running, float batch_value, float momentum) {
    return __fmaf_rn(momentum, batch_value, __fmul_rn(1.f - momentum, running));
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(double* __restrict__ stats, int slots, long long count,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* running_mean, float* running_var, float momentum,
                                                         float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                         float* save_mean, float* save_invstd, int C) {
    // 16 channels per workgroup (128-byte rows); the `slots` copies of the sums (one per pixel tile of the producing
    // conv) are shared out over 16 thread groups and meet in LDS
    __shared__ double part[2][16][17];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    if (stats) {
        double s1 = 0.0, s2 = 0.0;
        if (c < C)
            for (int k = q; k < slots; k += 16) {
                s1 += stats[(size_t)k * 2 * C + c];
                s2 += stats[(size_t)k * 2 * C + C + c];
            }
        part[0][q][cl] = s1;
        part[1][q][cl] = s2;
        __syncthreads();
    }
    if (c >= C || q != 0) return;
    float mean, var;
    if (stats) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            s1 += part[0][k][cl];
            s2 += part[1][k][cl];
        }
        if (slots > 1) {                 // copy 0 now holds the totals: a later finalize of the same sums passes 1 copy
            stats[c] = s1;
            stats[C + c] = s2;
        }
        const double m = s1 / (double)count;
        double v = s2 / (double)count - m * m;
        v = v > 0.0 ? v : 0.0;
        mean = (float)m;
        var = (float)v;
        if (momentum > 0.f && running_mean && running_var) {   // nn.BatchNorm2d training-mode update
            const double unbiased = count > 1 ? v * (double)count / (double)(count - 1) : v;
            running_mean[c] = bn_momentum_update(running_mean[c], mean, momentum);
            running_var[c] = bn_momentum_update(running_var[c], (float)unbiased, momentum);
        }
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - mean * sc;
    if (save_mean) save_mean[c] = mean;
    if (save_invstd) save_invstd[c] = invstd;
}

// y = [relu](a*sa[c] + ba[c] + (b ? b*sb[c] + bb[c] : 0)); grid.y = b*C + c planes, x over HW
template <bool VEC>
__global__ __launch_bounds__(256) void bn_add_relu_kernel(const float* __restrict__ a, const float* __restrict__ sa,
                                                         const float* __restrict__ ba, const float* __restrict__ b,
                                                         const float* __restrict__ sb, const float* __restrict__ bb,
                                                         float* __restrict__ y, int C, long long HW, int relu) {
    const long long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float s1 = sa ? sa[c] : 1.f, o1 = ba ? ba[c] : 0.f;
    const float s2 = sb ? sb[c] : 1.f, o2 = bb ? bb[c] : 0.f;
    const float* ap = a + plane * HW;
    const float* bp = b ? b + plane * HW : nullptr;
    float* yp = y + plane * HW;
    if (VEC) {
        const long long n4 = HW / 4;
        for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.y * blockDim.x) {
            float4 v = reinterpret_cast<const float4*>(ap)[i];
            v.x = v.x * s1 + o1; v.y = v.y * s1 + o1; v.z = v.z * s1 + o1; v.w = v.w * s1 + o1;
            if (bp) {
                const float4 w = reinterpret_cast<const float4*>(bp)[i];
                v.x += w.x * s2 + o2; v.y += w.y * s2 + o2; v.z += w.z * s2 + o2; v.w += w.w * s2 + o2;
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            reinterpret_cast<float4*>(yp)[i] = v;
        }
    } else {
        for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.y * blockDim.x) {
            float v = ap[i] * s1 + o1;
            if (bp) v += bp[i] * s2 + o2;
            if (relu) v = fmaxf(v, 0.f);
            yp[i] = v;
        }
    }
}

// 3x3 s2 p1 max pool, input optionally max(x*s+b, 0) on the fly
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, float* __restrict__ y, int C,
                                                          int Hin, int Win, int Ho, int Wo, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo);
        const int oy = (int)((idx / Wo) % Ho);
        const long long plane = idx / ((long long)Wo * Ho);
        const int c = (int)(plane % C);
        const float s = sc ? sc[c] : 1.f, o = sc ? sh[c] : 0.f;
        const float* xp = x + plane * Hin * Win;
        // nine loads in flight at once (clamped address, -inf where the window leaves the image): with a branch per tap they
        // ran one memory latency after the other
        float v[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                v[ky][kx] = xp[min(max(oy * 2 + ky - 1, 0), Hin - 1) * Win + min(max(ox * 2 + kx - 1, 0), Win - 1)];
        float m = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bool in = (unsigned)(oy * 2 + ky - 1) < (unsigned)Hin && (unsigned)(ox * 2 + kx - 1) < (unsigned)Win;
                float t = v[ky][kx];
                if (sc) t = fmaxf(t * s + o, 0.f);
                m = fmaxf(m, in ? t : -INFINITY);
            }
        y[idx] = m;
    }
}

// one wave per (b,c) plane
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            long long planes, long long HW) {
    const long long plane = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const float* xp = x + plane * HW;
    float s = 0.f;
    for (long long i = lane; i < HW; i += 64) s += xp[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) y[plane] = s / (float)HW;
}


// ---- BatchNorm backward (training mode), split like the forward --------------------------------------
// z = r*scale[c] + shift[c] is the BN output of raw conv output r.  The incoming gradient g is w.r.t.
//   MASK_RECOMPUTE: relu(z)            -> dz = g * (z > 0)          (conv -> bn -> relu chains)
//   MASK_TENSOR   : relu(z + identity) -> dz = g * (mask_src > 0)   (block end; mask_src = the block output)
//   MASK_NONE     : z                  -> dz = g                    (downsample branch; g already masked)
// pass 1 (one workgroup per (b,c) plane): sums[b,:,c] = { sum dz, sum dz * rhat } (layout [B][2][C]),  rhat = (r - mean)*invstd
// pass 2: dr = gamma*invstd * (dz - c1[c] - rhat*c2[c]),  c1 = sum_bhw dz / N, c2 = sum_bhw dz*rhat / N;
//         optionally also writes dz (the identity branch's gradient at a block end).
enum { MASK_NONE = 0, MASK_RECOMPUTE = 1, MASK_TENSOR = 2 };

__device__ __forceinline__ float bn_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ r,
                                                           const float* __restrict__ mask_src, int mask_mode,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float g_scale, long long g_plane_stride, float* __restrict__ sums,
                                                           int C, long long HW) {
    __shared__ float red[4];
    const long long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    // g_plane_stride == 0: g holds ONE value per plane (the global-average-pool gradient, broadcast)
    const float* gp = g + (g_plane_stride ? plane * g_plane_stride : plane);
    const float* rp = r + plane * HW;
    const float* mp = mask_src ? mask_src + plane * HW : nullptr;
    float s0 = 0.f, s1 = 0.f;
    for (long long i = threadIdx.x; i < HW; i += 256) {
        const float rv = rp[i];
        float d = (g_plane_stride ? gp[i] : gp[0]) * g_scale;
        if (mask_mode == MASK_RECOMPUTE) d = (rv * sc + sh > 0.f) ? d : 0.f;
        else if (mask_mode == MASK_TENSOR) d = (mp[i] > 0.f) ? d : 0.f;
        s0 += d;
        s1 += d * (rv - mu) * is;
    }
    s0 = bn_block_sum(s0, red);
    s1 = bn_block_sum(s1, red);
    if (threadIdx.x == 0) { const long long b = plane / C; sums[(2 * b) * C + c] = s0; sums[(2 * b + 1) * C + c] = s1; }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ r,
                                                          const float* __restrict__ mask_src, int mask_mode,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ csum, float inv_count, float g_scale,
                                                          long long g_plane_stride, float* __restrict__ dr, float* __restrict__ dz_out,
                                                          int C, long long HW, int n_sums, float* __restrict__ csum_out, float stat_w) {
    const long long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    // csum [n_sums][2][C]: the per-image sums of the reduce pass (n_sums = B: added up here, every workgroup for itself --
    // 2 B floats out of L2 instead of a reduction launch) or the batch totals (n_sums = 1)
    float t1 = 0.f, t2 = 0.f;
    for (int b = 0; b < n_sums; ++b) { t1 += csum[(size_t)(2 * b) * C + c]; t2 += csum[(size_t)(2 * b + 1) * C + c]; }
    if (csum_out && plane < C && blockIdx.y == 0 && threadIdx.x == 0) { csum_out[c] = t1; csum_out[C + c] = t2; }
    const float c1 = t1 * inv_count * stat_w, c2 = t2 * inv_count * stat_w;
    const float* gp = g + (g_plane_stride ? plane * g_plane_stride : plane);
    const float* rp = r + plane * HW;
    const float* mp = mask_src ? mask_src + plane * HW : nullptr;
    for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.y * blockDim.x) {
        const float rv = rp[i];
        float d = (g_plane_stride ? gp[i] : gp[0]) * g_scale;
        if (mask_mode == MASK_RECOMPUTE) d = (rv * sc + sh > 0.f) ? d : 0.f;
        else if (mask_mode == MASK_TENSOR) d = (mp[i] > 0.f) ? d : 0.f;
        if (dz_out) dz_out[plane * HW + i] = d;
        dr[plane * HW + i] = sc * (d - c1 - (rv - mu) * is * c2);     // sc = gamma * invstd
    }
}

// 16-byte forms of the two passes (HW % 4 == 0, 16-byte aligned tensors -- every layer of the trunk): TPP threads share a
// plane -- a whole workgroup for large planes, ONE WAVE for planes of <= 1024 floats (the 16^2 and 8^2 layers are 29 of
// the trunk's 53 BatchNorms: a 256-thread workgroup per 64-float plane left most lanes idle) -- and a thread of the apply
// pass handles at least four vectors (the dword form launched one workgroup per 256 elements and reached 43 % of HBM).
__device__ __forceinline__ float bn_mask(float d, float rv, float mv, int mask_mode, float sc, float sh) {
    if (mask_mode == MASK_RECOMPUTE) return (rv * sc + sh > 0.f) ? d : 0.f;
    if (mask_mode == MASK_TENSOR) return (mv > 0.f) ? d : 0.f;
    return d;
}

template <int TPP>
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const float* __restrict__ g, const float* __restrict__ r,
                                                               const float* __restrict__ mask_src, int mask_mode,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               float g_scale, long long g_plane_stride, float* __restrict__ sums,
                                                               int C, long long HW, long long planes) {
    __shared__ float red[4];
    const int t = threadIdx.x % TPP;
    const long long plane = (long long)blockIdx.x * (256 / TPP) + threadIdx.x / TPP;
    const bool live = plane < planes;                       // whole waves (TPP >= 64)
    float s0 = 0.f, s1 = 0.f;
    if (live) {
        const int c = (int)(plane % C);
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        const float* rp = r + plane * HW;
        const float* mp = mask_src ? mask_src + plane * HW : nullptr;
        const float g1 = g_plane_stride ? 0.f : g[plane] * g_scale;
        const float* gp = g + plane * g_plane_stride;
        for (long long i = (long long)t * 4; i < HW; i += TPP * 4) {
            const float4 rv = *reinterpret_cast<const float4*>(rp + i);
            float4 d = make_float4(g1, g1, g1, g1), mv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_plane_stride) {
                d = *reinterpret_cast<const float4*>(gp + i);
                d.x *= g_scale; d.y *= g_scale; d.z *= g_scale; d.w *= g_scale;
            }
            if (mask_mode == MASK_TENSOR) mv = *reinterpret_cast<const float4*>(mp + i);
            d.x = bn_mask(d.x, rv.x, mv.x, mask_mode, sc, sh); d.y = bn_mask(d.y, rv.y, mv.y, mask_mode, sc, sh);
            d.z = bn_mask(d.z, rv.z, mv.z, mask_mode, sc, sh); d.w = bn_mask(d.w, rv.w, mv.w, mask_mode, sc, sh);
            s0 += (d.x + d.y) + (d.z + d.w);
            s1 += (d.x * (rv.x - mu) + d.y * (rv.y - mu) + d.z * (rv.z - mu) + d.w * (rv.w - mu)) * is;
        }
    }
    if (TPP == 256) {
        s0 = bn_block_sum(s0, red);
        s1 = bn_block_sum(s1, red);
    } else {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    }
    if (live && t == 0) { const long long b = plane / C; const int c = (int)(plane % C); sums[(2 * b) * C + c] = s0; sums[(2 * b + 1) * C + c] = s1; }
}

template <int TPP>
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const float* __restrict__ g, const float* __restrict__ r,
                                                              const float* __restrict__ mask_src, int mask_mode,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ csum, float inv_count, float g_scale,
                                                              long long g_plane_stride, float* __restrict__ dr,
                                                              float* __restrict__ dz_out, int C, long long HW, long long planes,
                                                              int n_sums, float* __restrict__ csum_out, float stat_w) {
    const int t = threadIdx.x % TPP;
    const long long plane = (long long)blockIdx.x * (256 / TPP) + threadIdx.x / TPP;
    if (plane >= planes) return;
    const int c = (int)(plane % C);
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    float t1 = 0.f, t2 = 0.f;                               // (see bn_bwd_apply_kernel)
    for (int b = 0; b < n_sums; ++b) { t1 += csum[(size_t)(2 * b) * C + c]; t2 += csum[(size_t)(2 * b + 1) * C + c]; }
    if (csum_out && plane < C && blockIdx.y == 0 && t == 0) { csum_out[c] = t1; csum_out[C + c] = t2; }
    const float c1 = t1 * inv_count * stat_w, c2 = t2 * inv_count * stat_w;
    const float* rp = r + plane * HW;
    const float* mp = mask_src ? mask_src + plane * HW : nullptr;
    const float g1 = g_plane_stride ? 0.f : g[plane] * g_scale;
    const float* gp = g + plane * g_plane_stride;
    for (long long i = ((long long)blockIdx.y * TPP + t) * 4; i < HW; i += (long long)gridDim.y * TPP * 4) {
        const float4 rv = *reinterpret_cast<const float4*>(rp + i);
        float4 d = make_float4(g1, g1, g1, g1), mv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_plane_stride) {
            d = *reinterpret_cast<const float4*>(gp + i);
            d.x *= g_scale; d.y *= g_scale; d.z *= g_scale; d.w *= g_scale;
        }
        if (mask_mode == MASK_TENSOR) mv = *reinterpret_cast<const float4*>(mp + i);
        d.x = bn_mask(d.x, rv.x, mv.x, mask_mode, sc, sh); d.y = bn_mask(d.y, rv.y, mv.y, mask_mode, sc, sh);
        d.z = bn_mask(d.z, rv.z, mv.z, mask_mode, sc, sh); d.w = bn_mask(d.w, rv.w, mv.w, mask_mode, sc, sh);
        if (dz_out) *reinterpret_cast<float4*>(dz_out + plane * HW + i) = d;
        float4 o;
        o.x = sc * (d.x - c1 - (rv.x - mu) * is * c2); o.y = sc * (d.y - c1 - (rv.y - mu) * is * c2);     // sc = gamma * invstd
        o.z = sc * (d.z - c1 - (rv.z - mu) * is * c2); o.w = sc * (d.w - c1 - (rv.w - mu) * is * c2);
        *reinterpret_cast<float4*>(dr + plane * HW + i) = o;
    }
}

inline bool bn_vec_ok(long long HW, std::initializer_list<const void*> ptrs) {
    if (HW % 4) return false;
    for (const void* q : ptrs)
        if (q && (reinterpret_cast<uintptr_t>(q) & 15)) return false;
    return true;
}

// y[2h, 2w] = x[h, w], zeros elsewhere ([planes,H,W] -> [planes,Ho,Wo], Ho in {2H-1, 2H}): turns the data gradient
// of a stride-2 conv into a stride-1 conv with flipped weights.
__global__ __launch_bounds__(256) void dilate2x_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes,
                                                      int H, int W, int Ho, int Wo) {
    const long long total = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho);
        const long long pl = idx / ((long long)Wo * Ho);
        float v = 0.f;
        if (!(ox & 1) && !(oy & 1) && (oy >> 1) < H && (ox >> 1) < W) v = x[(pl * H + (oy >> 1)) * W + (ox >> 1)];
        y[idx] = v;
    }
}

// adjoint of maxpool3x3s2 (input optionally max(x*s+b,0) on the fly): every input pixel collects dy from the
// <= 4 windows in which it is the (first, row-major) maximum -- torch's tie rule.  A workgroup owns 32x32 input pixels
// of one plane: it first finds the argmax of the 17x17 windows that touch them (once each, code ky*3+kx in LDS; a
// per-pixel search would redo every window four times), then every pixel looks its windows up.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                              const float* __restrict__ sh, const float* __restrict__ dy,
                                                              float* __restrict__ dx, int C, int Hin, int Win, int Ho, int Wo,
                                                              int tiles_x, int tiles_y) {
    constexpr int TO = 16, TW = TO + 1;            // windows per tile side (+1: the odd last row / column reaches the next)
    constexpr int TI = 2 * TO + 3, TP = TI + 2;    // staged input rows / columns (2 oy0 - 1 .. 2 oy0 + 33: window 16 ends there), LDS pitch
    __shared__ unsigned char code[TW * TW];
    __shared__ float xs[TI * TP];
    __shared__ float gs[TW * TW];
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const long long plane = bx / tiles_y;
    const int oy0 = ty * TO, ox0 = tx * TO;
    const int c = (int)(plane % C);
    const float s = sc ? sc[c] : 1.f, o = sc ? sh[c] : 0.f;
    const float* xp = x + plane * Hin * Win;
    const float* gp = dy + plane * Ho * Wo;
    // the tile's input pixels once, row-contiguous, (folded BatchNorm + ReLU applied) into LDS, and the 17 x 17 gradients it can
    // receive: ALL global loads of the workgroup are issued before the first one is used (the first version searched the
    // windows with 2.25 strided gathers per pixel and looked the gradients up under a branch: 375 us for the stem's 450 MB)
    constexpr int NXL = (TI * TI + 255) / 256, NGL = (TW * TW + 255) / 256;
    float xv[NXL], gv[NGL];
#pragma unroll
    for (int k = 0; k < NXL; ++k) {
        const int e = min((int)threadIdx.x + 256 * k, TI * TI - 1);
        const int r = e / TI, q = e - r * TI;
        xv[k] = xp[min(max(2 * oy0 - 1 + r, 0), Hin - 1) * Win + min(max(2 * ox0 - 1 + q, 0), Win - 1)];
    }
#pragma unroll
    for (int k = 0; k < NGL; ++k) {
        const int w = min((int)threadIdx.x + 256 * k, TW * TW - 1);
        const int wy = w / TW, wx = w - wy * TW;
        gv[k] = gp[min(oy0 + wy, Ho - 1) * Wo + min(ox0 + wx, Wo - 1)];
    }
#pragma unroll
    for (int k = 0; k < NXL; ++k) {
        const int e = threadIdx.x + 256 * k;
        if (e < TI * TI) {
            const int r = e / TI, q = e - r * TI;
            const int yy = 2 * oy0 - 1 + r, xx = 2 * ox0 - 1 + q;
            float v = xv[k];
            if (sc) v = fmaxf(v * s + o, 0.f);
            xs[r * TP + q] = ((unsigned)yy < (unsigned)Hin && (unsigned)xx < (unsigned)Win) ? v : -INFINITY;
        }
    }
#pragma unroll
    for (int k = 0; k < NGL; ++k) {
        const int w = threadIdx.x + 256 * k;
        if (w < TW * TW) gs[w] = gv[k];
    }
    __syncthreads();
    for (int w = threadIdx.x; w < TW * TW; w += 256) {
        const int wy = w / TW, wx = w - wy * TW;
        const int oy = oy0 + wy, ox = ox0 + wx;
        int best = 255;
        if (oy < Ho && ox < Wo) {
            float m = -INFINITY;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = xs[(2 * wy + ky) * TP + 2 * wx + kx];      // (-inf outside the image: never the maximum)
                    if (v > m) { m = v; best = ky * 3 + kx; }
                }
        }
        code[w] = (unsigned char)best;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i;
        const int ly = e >> 5, lx = e & 31;
        const int iy = 2 * oy0 + ly, ix = 2 * ox0 + lx;
        if (iy >= Hin || ix >= Win) continue;
        float acc = 0.f;
        for (int wy = ly >> 1; wy <= (ly + 1) >> 1; ++wy) {
            const int ky = ly - 2 * wy + 1;
            for (int wx = lx >> 1; wx <= (lx + 1) >> 1; ++wx) {
                const int kx = lx - 2 * wx + 1;
                if (code[wy * TW + wx] == ky * 3 + kx) acc += gs[wy * TW + wx];                 // 255 = no such window
            }
        }
        dx[plane * Hin * Win + (long long)iy * Win + ix] = acc;
    }
}

}  // namespace

extern "C" {

int spk_bn_finalize(double* stats, int stats_slots, int64_t count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, float* scale, float* shift, float* save_mean,
                    float* save_invstd, int C, void* stream) {
    SPK_REQUIRE(scale && shift && C > 0, "bn_finalize: bad arguments");
    SPK_REQUIRE(stats || (running_mean && running_var), "bn_finalize: need batch sums or running statistics");
    SPK_REQUIRE(!stats || count > 0, "bn_finalize: element count must be positive");
    SPK_REQUIRE(stats_slots >= 0 && stats_slots <= 65536, "bn_finalize: bad stats_slots %d", stats_slots);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)spk::ceil_div(C, 16)), dim3(256), 0, (hipStream_t)stream, stats,
                       stats_slots > 1 ? stats_slots : 1, (long long)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean,
                       save_invstd, C);
    return spk::check_launch("bn_finalize_kernel");
}

namespace {
struct BnReplayList { spk_bn_replay_item it[SPK_BN_LIST_MAX]; };
// one workgroup per BatchNorm; the arithmetic of bn_finalize_kernel's running-statistics update, term for term
__global__ __launch_bounds__(256) void bn_replay_list_kernel(const BnReplayList L, float momentum) {
    const spk_bn_replay_item it = L.it[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += 256) {
        const double s1 = it.stats[c], s2 = it.stats[it.C + c];
        const double m = s1 / (double)it.count;
        double v = s2 / (double)it.count - m * m;
        v = v > 0.0 ? v : 0.0;
        const float mean = (float)m;
        const double unbiased = it.count > 1 ? v * (double)it.count / (double)(it.count - 1) : v;
        it.running_mean[c] = bn_momentum_update(it.running_mean[c], mean, momentum);
        it.running_var[c] = bn_momentum_update(it.running_var[c], (float)unbiased, momentum);
    }
}
}  // namespace

int spk_bn_replay_list(const spk_bn_replay_item* items_host, int n, float momentum, void* stream) {
    SPK_REQUIRE(items_host && n >= 1 && n <= SPK_BN_LIST_MAX, "bn_replay_list: 1..%d items", SPK_BN_LIST_MAX);
    BnReplayList L;
    for (int i = 0; i < n; ++i) {
        L.it[i] = items_host[i];
        SPK_REQUIRE(L.it[i].stats && L.it[i].running_mean && L.it[i].running_var && L.it[i].C > 0 && L.it[i].count > 0,
                    "bn_replay_list: bad item %d", i);
    }
    hipLaunchKernelGGL(bn_replay_list_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, L, momentum);
    return spk::check_launch("bn_replay_list_kernel");
}

int spk_bn_add_relu_fwd(const float* a, const float* sa, const float* ba, const float* b, const float* sb,
                        const float* bb, float* y, int B, int C, int64_t HW, int relu, void* stream) {
    SPK_REQUIRE(a && y && B > 0 && C > 0 && HW > 0, "bn_add_relu: bad arguments");
    SPK_REQUIRE((long long)B * C < (1ll << 31), "bn_add_relu: too many planes");
    const bool vec = HW % 4 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)y % 16 == 0 && (!b || (uintptr_t)b % 16 == 0);
    const long long work = vec ? HW / 4 : HW;
    dim3 grid((unsigned)(B * C), (unsigned)std::max(1ll, std::min((work + 255) / 256, 64ll)));   // planes on x (no 65535 limit)
    if (vec) hipLaunchKernelGGL(bn_add_relu_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, sa, ba, b, sb, bb, y, C, (long long)HW, relu);
    else     hipLaunchKernelGGL(bn_add_relu_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, sa, ba, b, sb, bb, y, C, (long long)HW, relu);
    return spk::check_launch("bn_add_relu_kernel");
}

int spk_maxpool3x3s2_fwd(const float* x, const float* in_scale, const float* in_shift, float* y, int B, int C,
                         int Hin, int Win, void* stream) {
    SPK_REQUIRE(x && y && B > 0 && C > 0 && Hin > 0 && Win > 0, "maxpool: bad arguments");
    SPK_REQUIRE(!in_scale == !in_shift, "maxpool: in_scale and in_shift go together");
    const int Ho = (Hin - 1) / 2 + 1, Wo = (Win - 1) / 2 + 1;
    const long long total = (long long)B * C * Ho * Wo;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)std::min((total + 255) / 256, 256ll * 16)), dim3(256), 0,
                       (hipStream_t)stream, x, in_scale, in_shift, y, C, Hin, Win, Ho, Wo, total);
    return spk::check_launch("maxpool3x3s2_kernel");
}

int spk_global_avgpool_fwd(const float* x, float* y, int64_t planes, int64_t HW, void* stream) {
    SPK_REQUIRE(x && y && planes > 0 && HW > 0, "avgpool: bad arguments");
    hipLaunchKernelGGL(global_avgpool_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y,
                       (long long)planes, (long long)HW);
    return spk::check_launch("global_avgpool_kernel");
}

int spk_bn_bwd_reduce(const float* g, const float* r, const float* mask_src, int mask_mode, const float* scale,
                      const float* shift, const float* mean, const float* invstd, float g_scale, int g_per_plane,
                      float* sums, int B, int C, int64_t HW, void* stream) {
    SPK_REQUIRE(g && r && scale && shift && mean && invstd && sums && B > 0 && C > 0 && HW > 0, "bn_bwd_reduce: bad arguments");
    SPK_REQUIRE(mask_mode != MASK_TENSOR || mask_src, "bn_bwd_reduce: MASK_TENSOR without mask tensor");
    const long long planes = (long long)B * C, gps = g_per_plane ? 0 : HW;
    if (bn_vec_ok(HW, {g_per_plane ? nullptr : g, r, mask_src})) {
        if (HW <= 1024)
            hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel<64>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, r,
                               mask_src, mask_mode, scale, shift, mean, invstd, g_scale, gps, sums, C, (long long)HW, planes);
        else
            hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel<256>, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, g, r, mask_src,
                               mask_mode, scale, shift, mean, invstd, g_scale, gps, sums, C, (long long)HW, planes);
        return spk::check_launch("bn_bwd_reduce_vec_kernel");
    }
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, g, r, mask_src, mask_mode,
                       scale, shift, mean, invstd, g_scale, gps, sums, C, (long long)HW);
    return spk::check_launch("bn_bwd_reduce_kernel");
}

static int bn_bwd_apply_impl(const float* g, const float* r, const float* mask_src, int mask_mode, const float* scale,
                             const float* shift, const float* mean, const float* invstd, const float* csum, int n_sums,
                             float* csum_out, float stat_w, int64_t count, float g_scale, int g_per_plane, float* dr,
                             float* dz_out, int B, int C, int64_t HW, void* stream) {
    SPK_REQUIRE(g && r && scale && shift && mean && invstd && csum && dr && B > 0 && C > 0 && HW > 0 && count > 0 && n_sums >= 1,
                "bn_bwd_apply: bad arguments");
    SPK_REQUIRE(mask_mode != MASK_TENSOR || mask_src, "bn_bwd_apply: MASK_TENSOR without mask tensor");
    const long long planes = (long long)B * C, gps = g_per_plane ? 0 : HW;
    if (bn_vec_ok(HW, {g_per_plane ? nullptr : g, r, mask_src, dr, dz_out})) {
        if (HW <= 1024) {
            hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<64>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, r,
                               mask_src, mask_mode, scale, shift, mean, invstd, csum, 1.0f / (float)count, g_scale, gps, dr, dz_out,
                               C, (long long)HW, planes, n_sums, csum_out, stat_w);
        } else {
            const unsigned chunks = (unsigned)std::max(1ll, std::min((long long)HW / (256 * 4 * 4), 16ll));   // >= 4 vectors per thread
            hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<256>, dim3((unsigned)planes, chunks), dim3(256), 0, (hipStream_t)stream, g, r,
                               mask_src, mask_mode, scale, shift, mean, invstd, csum, 1.0f / (float)count, g_scale, gps, dr, dz_out,
                               C, (long long)HW, planes, n_sums, csum_out, stat_w);
        }
        return spk::check_launch("bn_bwd_apply_vec_kernel");
    }
    dim3 grid((unsigned)(B * C), (unsigned)std::max(1ll, std::min(((long long)HW + 255) / 256, 64ll)));
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, r, mask_src, mask_mode, scale, shift, mean,
                       invstd, csum, 1.0f / (float)count, g_scale, (long long)(g_per_plane ? 0 : HW), dr, dz_out, C, (long long)HW,
                       n_sums, csum_out, stat_w);
    return spk::check_launch("bn_bwd_apply_kernel");
}

int spk_bn_bwd_apply(const float* g, const float* r, const float* mask_src, int mask_mode, const float* scale,
                     const float* shift, const float* mean, const float* invstd, const float* csum, int64_t count,
                     float g_scale, int g_per_plane, float* dr, float* dz_out, int B, int C, int64_t HW, void* stream) {
    return bn_bwd_apply_impl(g, r, mask_src, mask_mode, scale, shift, mean, invstd, csum, 1, nullptr, 1.f, count, g_scale, g_per_plane,
                             dr, dz_out, B, C, HW, stream);
}

int spk_bn_bwd_apply_sums(const float* g, const float* r, const float* mask_src, int mask_mode, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const float* sums, float* csum_out,
                          int batch_stats, int64_t count, float g_scale, int g_per_plane, float* dr, float* dz_out, int B, int C,
                          int64_t HW, void* stream) {
    SPK_REQUIRE(csum_out, "bn_bwd_apply_sums: null csum_out");
    return bn_bwd_apply_impl(g, r, mask_src, mask_mode, scale, shift, mean, invstd, sums, B, csum_out, batch_stats ? 1.f : 0.f, count,
                             g_scale, g_per_plane, dr, dz_out, B, C, HW, stream);
}

int spk_dilate2x(const float* x, float* y, int64_t planes, int H, int W, int Ho, int Wo, void* stream) {
    SPK_REQUIRE(x && y && planes > 0 && H > 0 && W > 0, "dilate2x: bad arguments");
    SPK_REQUIRE((Ho == 2 * H || Ho == 2 * H - 1) && (Wo == 2 * W || Wo == 2 * W - 1), "dilate2x: output must be 2H or 2H-1");
    const long long total = planes * Ho * Wo;
    hipLaunchKernelGGL(dilate2x_kernel, dim3((unsigned)std::min((total + 255) / 256, 256ll * 16)), dim3(256), 0,
                       (hipStream_t)stream, x, y, (long long)planes, H, W, Ho, Wo);
    return spk::check_launch("dilate2x_kernel");
}

int spk_maxpool3x3s2_bwd(const float* x, const float* in_scale, const float* in_shift, const float* dy, float* dx, int B,
                         int C, int Hin, int Win, void* stream) {
    SPK_REQUIRE(x && dy && dx && B > 0 && C > 0 && Hin > 0 && Win > 0, "maxpool_bwd: bad arguments");
    SPK_REQUIRE(!in_scale == !in_shift, "maxpool_bwd: in_scale and in_shift go together");
    const int Ho = (Hin - 1) / 2 + 1, Wo = (Win - 1) / 2 + 1;
    const int tiles_x = spk::ceil_div(Win, 32), tiles_y = spk::ceil_div(Hin, 32);
    const long long blocks = (long long)B * C * tiles_x * tiles_y;
    SPK_REQUIRE(blocks < (1ll << 31), "maxpool_bwd: too many tiles");
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, in_scale,
                       in_shift, dy, dx, C, Hin, Win, Ho, Wo, tiles_x, tiles_y);
    return spk::check_launch("maxpool3x3s2_bwd_kernel");
}

}  // extern "C"
