// BatchNorm2d pieces and pooling of the ResNet-50 trunk (torchvision resnet50, built at
// model.py:60-62).  BatchNorm is split MI355X-style: the producing conv accumulates the batch sums
// in its epilogue (SPK_EPI_STATS), bn_finalize turns them into a per-channel affine (and updates the
// running statistics), and the *consumer* applies affine+ReLU while staging its input -- the
// normalised tensor never exists in HBM.  Only the block output (bn3 + identity + ReLU) is
// materialised, by the HBM-bound bn_add_relu pass.
#include "spk_common.hpp"

#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ stats, long long count,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* running_mean, float* running_var, float momentum,
                                                         float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                         float* save_mean, float* save_invstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (stats) {
        const double m = stats[c] / (double)count;
        double v = stats[C + c] / (double)count - m * m;
        v = v > 0.0 ? v : 0.0;
        mean = (float)m;
        var = (float)v;
        if (momentum > 0.f && running_mean && running_var) {   // nn.BatchNorm2d training-mode update
            const double unbiased = count > 1 ? v * (double)count / (double)(count - 1) : v;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
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
    const long long plane = blockIdx.y;
    const int c = (int)(plane % C);
    const float s1 = sa ? sa[c] : 1.f, o1 = ba ? ba[c] : 0.f;
    const float s2 = sb ? sb[c] : 1.f, o2 = bb ? bb[c] : 0.f;
    const float* ap = a + plane * HW;
    const float* bp = b ? b + plane * HW : nullptr;
    float* yp = y + plane * HW;
    if (VEC) {
        const long long n4 = HW / 4;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
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
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
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
        float m = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 + ky - 1;
            if (iy < 0 || iy >= Hin) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 + kx - 1;
                if (ix < 0 || ix >= Win) continue;
                float v = xp[iy * Win + ix];
                if (sc) v = fmaxf(v * s + o, 0.f);
                m = fmaxf(m, v);
            }
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

}  // namespace

extern "C" {

int spk_bn_finalize(const double* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, float* scale, float* shift, float* save_mean,
                    float* save_invstd, int C, void* stream) {
    SPK_REQUIRE(scale && shift && C > 0, "bn_finalize: bad arguments");
    SPK_REQUIRE(stats || (running_mean && running_var), "bn_finalize: need batch sums or running statistics");
    SPK_REQUIRE(!stats || count > 0, "bn_finalize: element count must be positive");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)spk::ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, stats,
                       (long long)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean,
                       save_invstd, C);
    return spk::check_launch("bn_finalize_kernel");
}

int spk_bn_add_relu_fwd(const float* a, const float* sa, const float* ba, const float* b, const float* sb,
                        const float* bb, float* y, int B, int C, int64_t HW, int relu, void* stream) {
    SPK_REQUIRE(a && y && B > 0 && C > 0 && HW > 0, "bn_add_relu: bad arguments");
    SPK_REQUIRE((long long)B * C < (1ll << 31), "bn_add_relu: too many planes");
    const bool vec = HW % 4 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)y % 16 == 0 && (!b || (uintptr_t)b % 16 == 0);
    const long long work = vec ? HW / 4 : HW;
    dim3 grid((unsigned)std::max(1ll, std::min((work + 255) / 256, 64ll)), (unsigned)(B * C));
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

}  // extern "C"
