// Stand-alone StyleGAN1 / ProGAN ops the reference defines (the north-star's "PixelNorm / upfirdn" family):
// PixelNorm (both spellings), InstanceNorm / AdaIN, Blur2d (depthwise FIR, stride 1/2), Upscale2d (nearest),
// the fade-in blend.  All HBM-bound single kernels; citations per entry point in include/spk.h.
#include "spk_common.hpp"

#include <algorithm>

namespace {

// y[b,c,p] = x[b,c,p] * rsqrt(mean_c x^2 + eps)   (mode 0)   or   x / sqrt(mean_c x^2 + eps)   (mode 1)
__global__ __launch_bounds__(256) void pixelnorm_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                       long long HW, long long total_px, float eps, int mode) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_px; idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW, p = idx % HW;
        const float* xp = x + b * C * HW + p;
        float s = 0.f;
        for (int c = 0; c < C; ++c) { const float v = xp[(size_t)c * HW]; s += v * v; }
        const float m = s / (float)C + eps;
        const float f = mode == 0 ? rsqrtf(m) : 1.f / sqrtf(m);
        float* yp = y + b * C * HW + p;
        for (int c = 0; c < C; ++c) yp[(size_t)c * HW] = xp[(size_t)c * HW] * f;
    }
}

__device__ __forceinline__ float lg_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// few pixels, many channels (the mapping network's [B,6144] latent): one workgroup per pixel, channels across the lanes
__global__ __launch_bounds__(256) void pixelnorm_wide_kernel(const float* __restrict__ x, float* __restrict__ y, int C, long long HW,
                                                            float eps, int mode) {
    __shared__ float red[4];
    const long long b = blockIdx.x / HW, p = blockIdx.x % HW;
    const float* xp = x + b * C * HW + p;
    float* yp = y + b * C * HW + p;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float v = xp[(size_t)c * HW]; s += v * v; }
    const float m = lg_block_sum(s, red) / (float)C + eps;
    const float f = mode == 0 ? rsqrtf(m) : 1.f / sqrtf(m);
    for (int c = threadIdx.x; c < C; c += 256) yp[(size_t)c * HW] = xp[(size_t)c * HW] * f;
}

// one workgroup per (b,c) plane: mean, biased variance (two-pass), y = (x-mean)*rsqrt(var+eps)*scale[b,c] + bias[b,c]
__global__ __launch_bounds__(256) void instance_norm_affine_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  const float* __restrict__ scale, const float* __restrict__ bias,
                                                                  long long sb_stride, int C, long long HW, float eps) {
    __shared__ float red[4];
    const long long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const long long b = plane / C;
    const float* xp = x + plane * HW;
    float* yp = y + plane * HW;
    float s = 0.f;
    for (long long i = threadIdx.x; i < HW; i += 256) s += xp[i];
    const float mean = lg_block_sum(s, red) / (float)HW;
    float v = 0.f;
    for (long long i = threadIdx.x; i < HW; i += 256) { const float d = xp[i] - mean; v += d * d; }
    const float invstd = rsqrtf(lg_block_sum(v, red) / (float)HW + eps);
    const float g = scale ? scale[b * sb_stride + c] : 1.f, o = bias ? bias[b * sb_stride + c] : 0.f;
    for (long long i = threadIdx.x; i < HW; i += 256) yp[i] = (xp[i] - mean) * invstd * g + o;
}

struct Fir { float f[49]; int k; };

// Adjoint of instance_norm_affine_kernel, one workgroup per (b,c) plane.  With xh = (x-mean)*invstd, g = scale[b,c]:
//   dx = g * invstd * (dy - mean(dy) - xh * mean(dy*xh)),   dscale[b,c] = sum dy*xh,   dbias[b,c] = sum dy
__global__ __launch_bounds__(256) void instance_norm_affine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                      const float* __restrict__ scale, long long sb_stride,
                                                                      float* __restrict__ dx, float* __restrict__ dscale,
                                                                      float* __restrict__ dbias, int C, long long HW, float eps) {
    __shared__ float red[4];
    const long long plane = blockIdx.x;
    const int c = (int)(plane % C);
    const long long b = plane / C;
    const float* xp = x + plane * HW;
    const float* gp = dy + plane * HW;
    float s = 0.f;
    for (long long i = threadIdx.x; i < HW; i += 256) s += xp[i];
    const float mean = lg_block_sum(s, red) / (float)HW;
    float v = 0.f;
    for (long long i = threadIdx.x; i < HW; i += 256) { const float d = xp[i] - mean; v += d * d; }
    const float invstd = rsqrtf(lg_block_sum(v, red) / (float)HW + eps);
    float s1 = 0.f, s2 = 0.f;
    for (long long i = threadIdx.x; i < HW; i += 256) { const float g = gp[i]; s1 += g; s2 += g * (xp[i] - mean) * invstd; }
    const float sum_dy = lg_block_sum(s1, red), sum_dyxh = lg_block_sum(s2, red);
    if (threadIdx.x == 0) {
        if (dscale) dscale[plane] = sum_dyxh;
        if (dbias) dbias[plane] = sum_dy;
    }
    if (dx) {
        const float g = scale ? scale[b * sb_stride + c] : 1.f;
        const float m1 = sum_dy / (float)HW, m2 = sum_dyxh / (float)HW;
        float* dp = dx + plane * HW;
        for (long long i = threadIdx.x; i < HW; i += 256) dp[i] = g * invstd * (gp[i] - m1 - (xp[i] - mean) * invstd * m2);
    }
}

// depthwise k x k FIR, zero padding (k-1)/2, stride 1 or 2
__global__ __launch_bounds__(256) void blur2d_kernel(const float* __restrict__ x, float* __restrict__ y, Fir fir, long long planes,
                                                    int H, int W, int Ho, int Wo, int stride) {
    const int k = fir.k, pad = (k - 1) / 2;
    const long long total = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho);
        const long long pl = idx / ((long long)Wo * Ho);
        const float* xp = x + pl * H * W;
        float acc = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride + ky - pad;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride + kx - pad;
                if (ix >= 0 && ix < W) acc += fir.f[ky * k + kx] * xp[(size_t)iy * W + ix];
            }
        }
        y[idx] = acc;
    }
}

__global__ __launch_bounds__(256) void upscale2d_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H,
                                                       int W, int factor, float gain) {
    const int Ho = H * factor, Wo = W * factor;
    const long long total = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho);
        const long long pl = idx / ((long long)Wo * Ho);
        y[idx] = x[(pl * H + oy / factor) * W + ox / factor] * gain;
    }
}

// Adjoint of pixelnorm: with r = rsqrt(mean_c x^2 + eps):  dx_j = r*dy_j - x_j * r^3 * mean_c(x*dy)   (both spellings of the
// forward are the same function).  One thread per pixel, channels strided by HW (coalesced across the wave).
__global__ __launch_bounds__(256) void pixelnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int C, long long HW, long long total_px, float eps) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_px; idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW, p = idx % HW;
        const float* xp = x + b * C * HW + p;
        const float* gp = dy + b * C * HW + p;
        float s = 0.f, t = 0.f;
        for (int c = 0; c < C; ++c) { const float v = xp[(size_t)c * HW]; s += v * v; t += v * gp[(size_t)c * HW]; }
        const float r = rsqrtf(s / (float)C + eps);
        const float k = r * r * r * (t / (float)C);
        float* dp = dx + b * C * HW + p;
        for (int c = 0; c < C; ++c) dp[(size_t)c * HW] = r * gp[(size_t)c * HW] - xp[(size_t)c * HW] * k;
    }
}

// few pixels, many channels: one workgroup per pixel
__global__ __launch_bounds__(256) void pixelnorm_bwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ dx, int C, long long HW, float eps) {
    __shared__ float red[4];
    const long long b = blockIdx.x / HW, p = blockIdx.x % HW;
    const float* xp = x + b * C * HW + p;
    const float* gp = dy + b * C * HW + p;
    float s = 0.f, t = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float v = xp[(size_t)c * HW]; s += v * v; t += v * gp[(size_t)c * HW]; }
    const float ss = lg_block_sum(s, red), tt = lg_block_sum(t, red);
    const float r = rsqrtf(ss / (float)C + eps);
    const float k = r * r * r * (tt / (float)C);
    float* dp = dx + b * C * HW + p;
    for (int c = threadIdx.x; c < C; c += 256) dp[(size_t)c * HW] = r * gp[(size_t)c * HW] - xp[(size_t)c * HW] * k;
}

// Adjoint of blur2d_kernel (gather form): dx[iy,ix] = sum_{ky,kx} f[ky,kx] * dy[(iy+pad-ky)/s, (ix+pad-kx)/s] over the taps
// whose quotient is exact and inside the output.
__global__ __launch_bounds__(256) void blur2d_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, Fir fir, long long planes,
                                                        int H, int W, int Ho, int Wo, int stride) {
    const int k = fir.k, pad = (k - 1) / 2;
    const long long total = planes * H * W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(idx % W), iy = (int)((idx / W) % H);
        const long long pl = idx / ((long long)W * H);
        const float* gp = dy + pl * Ho * Wo;
        float acc = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int ny = iy + pad - ky;
            if (ny < 0 || ny % stride) continue;
            const int oy = ny / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int nx = ix + pad - kx;
                if (nx < 0 || nx % stride) continue;
                const int ox = nx / stride;
                if (ox < Wo) acc += fir.f[ky * k + kx] * gp[(size_t)oy * Wo + ox];
            }
        }
        dx[idx] = acc;
    }
}

// Adjoint of upscale2d_kernel: dx[y,x] = gain * sum of the factor x factor block of dy
__global__ __launch_bounds__(256) void upscale2d_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long planes, int H,
                                                           int W, int factor, float gain) {
    const int Wo = W * factor;
    const long long total = planes * H * W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(idx % W), iy = (int)((idx / W) % H);
        const long long pl = idx / ((long long)W * H);
        const float* gp = dy + (pl * H * factor + (long long)iy * factor) * Wo + (long long)ix * factor;
        float acc = 0.f;
        for (int a = 0; a < factor; ++a)
            for (int b = 0; b < factor; ++b) acc += gp[(size_t)a * Wo + b];
        dx[idx] = acc * gain;
    }
}

__global__ __launch_bounds__(256) void fade_in_tanh_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                          float alpha, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = tanhf(alpha * a[i] + (1.f - alpha) * b[i]);
}

inline unsigned sgrid(long long n) { return (unsigned)std::max(1ll, std::min((n + 255) / 256, 256ll * 16)); }

}  // namespace

extern "C" {

int spk_pixelnorm_fwd(const float* x, float* y, int B, int C, int64_t HW, float eps, int sqrt_form, void* stream) {
    SPK_REQUIRE(x && y && B > 0 && C > 0 && HW > 0, "pixelnorm: bad arguments");
    if ((long long)B * HW <= 4096 && C >= 256) {
        hipLaunchKernelGGL(pixelnorm_wide_kernel, dim3((unsigned)(B * HW)), dim3(256), 0, (hipStream_t)stream, x, y, C, (long long)HW,
                           eps, sqrt_form ? 1 : 0);
        return spk::check_launch("pixelnorm_wide_kernel");
    }
    hipLaunchKernelGGL(pixelnorm_kernel, dim3(sgrid((long long)B * HW)), dim3(256), 0, (hipStream_t)stream, x, y, C, (long long)HW,
                       (long long)B * HW, eps, sqrt_form ? 1 : 0);
    return spk::check_launch("pixelnorm_kernel");
}

int spk_instance_norm_affine_fwd(const float* x, float* y, const float* scale, const float* bias, int64_t sb_stride, int B, int C,
                                 int64_t HW, float eps, void* stream) {
    SPK_REQUIRE(x && y && B > 0 && C > 0 && HW > 0, "instance_norm: bad arguments");
    hipLaunchKernelGGL(instance_norm_affine_kernel, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, x, y, scale, bias,
                       (long long)sb_stride, C, (long long)HW, eps);
    return spk::check_launch("instance_norm_affine_kernel");
}

int spk_instance_norm_affine_bwd(const float* x, const float* dy, const float* scale, int64_t sb_stride, float* dx, float* dscale,
                                 float* dbias, int B, int C, int64_t HW, float eps, void* stream) {
    SPK_REQUIRE(x && dy && B > 0 && C > 0 && HW > 0, "instance_norm_bwd: bad arguments");
    hipLaunchKernelGGL(instance_norm_affine_bwd_kernel, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, x, dy, scale,
                       (long long)sb_stride, dx, dscale, dbias, C, (long long)HW, eps);
    return spk::check_launch("instance_norm_affine_bwd_kernel");
}

int spk_blur2d_fwd(const float* x, float* y, const float* filter_host, int k, int64_t planes, int H, int W, int stride, void* stream) {
    SPK_REQUIRE(x && y && filter_host && k >= 1 && k <= 7 && planes > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2),
                "blur2d: bad arguments (k <= 7, stride 1 or 2)");
    Fir fir;
    fir.k = k;
    for (int i = 0; i < k * k; ++i) fir.f[i] = filter_host[i];
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(blur2d_kernel, dim3(sgrid((long long)planes * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, y, fir,
                       (long long)planes, H, W, Ho, Wo, stride);
    return spk::check_launch("blur2d_kernel");
}

int spk_upscale2d_nearest_fwd(const float* x, float* y, int64_t planes, int H, int W, int factor, float gain, void* stream) {
    SPK_REQUIRE(x && y && planes > 0 && H > 0 && W > 0 && factor >= 1, "upscale2d: bad arguments");
    hipLaunchKernelGGL(upscale2d_kernel, dim3(sgrid((long long)planes * H * W * factor * factor)), dim3(256), 0, (hipStream_t)stream,
                       x, y, (long long)planes, H, W, factor, gain);
    return spk::check_launch("upscale2d_kernel");
}

int spk_pixelnorm_bwd(const float* x, const float* dy, float* dx, int B, int C, int64_t HW, float eps, void* stream) {
    SPK_REQUIRE(x && dy && dx && B > 0 && C > 0 && HW > 0, "pixelnorm_bwd: bad arguments");
    if ((long long)B * HW <= 4096 && C >= 256) {
        hipLaunchKernelGGL(pixelnorm_bwd_wide_kernel, dim3((unsigned)(B * HW)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, C,
                           (long long)HW, eps);
        return spk::check_launch("pixelnorm_bwd_wide_kernel");
    }
    hipLaunchKernelGGL(pixelnorm_bwd_kernel, dim3(sgrid((long long)B * HW)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, C,
                       (long long)HW, (long long)B * HW, eps);
    return spk::check_launch("pixelnorm_bwd_kernel");
}

int spk_blur2d_bwd(const float* dy, float* dx, const float* filter_host, int k, int64_t planes, int H, int W, int stride, void* stream) {
    SPK_REQUIRE(dy && dx && filter_host && k >= 1 && k <= 7 && planes > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2),
                "blur2d_bwd: bad arguments (k <= 7, stride 1 or 2)");
    Fir fir;
    fir.k = k;
    for (int i = 0; i < k * k; ++i) fir.f[i] = filter_host[i];
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(blur2d_bwd_kernel, dim3(sgrid((long long)planes * H * W)), dim3(256), 0, (hipStream_t)stream, dy, dx, fir,
                       (long long)planes, H, W, Ho, Wo, stride);
    return spk::check_launch("blur2d_bwd_kernel");
}

int spk_upscale2d_nearest_bwd(const float* dy, float* dx, int64_t planes, int H, int W, int factor, float gain, void* stream) {
    SPK_REQUIRE(dy && dx && planes > 0 && H > 0 && W > 0 && factor >= 1, "upscale2d_bwd: bad arguments");
    hipLaunchKernelGGL(upscale2d_bwd_kernel, dim3(sgrid((long long)planes * H * W)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                       (long long)planes, H, W, factor, gain);
    return spk::check_launch("upscale2d_bwd_kernel");
}

int spk_fade_in_tanh_fwd(const float* a, const float* b, float* y, float alpha, int64_t n, void* stream) {
    SPK_REQUIRE(a && b && y && n > 0, "fade_in: bad arguments");
    hipLaunchKernelGGL(fade_in_tanh_kernel, dim3(sgrid(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, alpha, (long long)n);
    return spk::check_launch("fade_in_tanh_kernel");
}

}  // extern "C"
