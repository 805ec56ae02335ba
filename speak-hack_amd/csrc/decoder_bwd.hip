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
    s_dy = block_sum(s_dy, red);
    s_dya = block_sum(s_dya, red);
    s_dt = block_sum(s_dt, red);
    s_dtn = block_sum(s_dtn, red);
    if (threadIdx.x == 0) {
        float* o = sums + plane * 4;
        o[0] = s_dy; o[1] = s_dya; o[2] = s_dt; o[3] = s_dtn;
    }
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
        float acc = 0.f;
        for (int uy = max(2 * iy - 2, 0); uy <= min(2 * iy + 2, Ho - 1); ++uy) {
            const int y0 = uy == 0 ? 0 : (uy - 1) >> 1, y1 = min(y0 + 1, Hin - 1);
            const float ly1 = uy == 0 ? 0.f : ((uy & 1) ? 0.25f : 0.75f);
            const float wy = (y0 == iy ? 1.f - ly1 : 0.f) + (y1 == iy ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int ux = max(2 * ix - 2, 0); ux <= min(2 * ix + 2, Wo - 1); ++ux) {
                const int x0 = ux == 0 ? 0 : (ux - 1) >> 1, x1 = min(x0 + 1, Win - 1);
                const float lx1 = ux == 0 ? 0.f : ((ux & 1) ? 0.25f : 0.75f);
                const float wx = (x0 == ix ? 1.f - lx1 : 0.f) + (x1 == ix ? lx1 : 0.f);
                if (wx != 0.f) acc += wy * wx * g[(size_t)uy * Wo + ux];
            }
        }
        dx[idx] = acc;
    }
}

// toRGB backward.  Each workgroup owns a pixel range of one image:
//   dx[b,c,p] = in_scale * sum_o w[o,c] * dy[b,o,p]                      (written)
//   partial[blk][o*C + c] = in_scale * sum_p dy[b,o,p] * x[b,c,p];  partial[blk][O*C + o] = sum_p dy[b,o,p]
__global__ __launch_bounds__(256) void conv1x1_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ dy, float* __restrict__ dx,
                                                               float* __restrict__ partial, int C, int O, long long HW,
                                                               long long px_per_block, float in_scale) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long long p_begin = (long long)blockIdx.x * px_per_block, p_end = min(HW, p_begin + px_per_block);
    const float* xb = x + (size_t)b * C * HW;
    const float* dyb = dy + (size_t)b * O * HW;
    float* dxb = dx ? dx + (size_t)b * C * HW : nullptr;
    float* part = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (O * C + O);
    for (int c = 0; c < C; ++c) {
        float wc[4], s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) wc[o] = o < O ? w[o * C + c] * in_scale : 0.f;
        for (long long pp = p_begin + threadIdx.x; pp < p_end; pp += 256) {
            const float xv = xb[(size_t)c * HW + pp];
            float d = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < O) {
                    const float g = dyb[(size_t)o * HW + pp];
                    d += wc[o] * g;
                    s[o] += g * xv;
                }
            if (dxb) dxb[(size_t)c * HW + pp] = d;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < O) {
                const float t = block_sum(s[o], red);
                if (threadIdx.x == 0) part[o * C + c] = t * in_scale;
            }
    }
    for (int o = 0; o < O; ++o) {
        float s = 0.f;
        for (long long pp = p_begin + threadIdx.x; pp < p_end; pp += 256) s += dyb[(size_t)o * HW + pp];
        s = block_sum(s, red);
        if (threadIdx.x == 0) part[O * C + o] = s;
    }
}

// dx[b,i] = wmul * sum_o dz[b,o] * w[o,i],  dz = dout * (out > 0 ? 1 : slope).  Lanes run along i.
__global__ __launch_bounds__(256) void fc_bwd_input_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                          const float* __restrict__ w, float* __restrict__ dx,
                                                          long long dx_stride, int B, int I, int O, float wmul, float slope) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= I) return;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) {
        const float ov = out[(size_t)b * O + o];
        const float dz = dout[(size_t)b * O + o] * (ov > 0.f ? 1.f : slope);
        acc += dz * w[(size_t)o * I + i];
    }
    dx[(size_t)b * dx_stride + i] = acc * wmul;
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

}  // namespace

extern "C" {

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
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)std::min((total + 255) / 256, 256ll * 16)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, (long long)planes, Hin, Win);
    return spk::check_launch("upsample2x_bwd_kernel");
}

int spk_conv1x1_small_bwd_blocks(int B, int64_t HW) {
    const long long per_img = std::max(1ll, std::min((long long)(HW + 4095) / 4096, 64ll));
    return (int)(per_img * B);
}

int spk_conv1x1_small_bwd(const float* x, const float* w, const float* dy, float* dx, float* partial, int B, int C, int O,
                          int64_t HW, float in_scale, void* stream) {
    SPK_REQUIRE(x && w && dy && partial, "conv1x1_small_bwd: null pointer");
    SPK_REQUIRE(B > 0 && C > 0 && O > 0 && O <= 4 && HW > 0, "conv1x1_small_bwd: bad shape (O must be <= 4)");
    const int per_img = spk_conv1x1_small_bwd_blocks(B, HW) / B;
    const long long px_per_block = (HW + per_img - 1) / per_img;
    hipLaunchKernelGGL(conv1x1_small_bwd_kernel, dim3((unsigned)per_img, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, w,
                       dy, dx, partial, C, O, (long long)HW, px_per_block, in_scale);
    return spk::check_launch("conv1x1_small_bwd_kernel");
}

int spk_fc_bwd(const float* dout, const float* out, const float* x, int64_t x_stride, const float* w, float* dx,
               int64_t dx_stride, float* dw, float* db, int B, int I, int O, float wmul, float bmul, float slope,
               void* stream) {
    SPK_REQUIRE(dout && out && x && w, "fc_bwd: null pointer");
    SPK_REQUIRE(B > 0 && I > 0 && O > 0, "fc_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        hipLaunchKernelGGL(fc_bwd_input_kernel, dim3((unsigned)spk::ceil_div(I, 256), (unsigned)B), dim3(256), 0, s, dout, out,
                           w, dx, (long long)dx_stride, B, I, O, wmul, slope);
        int rc = spk::check_launch("fc_bwd_input_kernel");
        if (rc != SPK_OK) return rc;
    }
    if (dw) {
        hipLaunchKernelGGL(fc_bwd_weight_kernel, dim3((unsigned)std::min(spk::ceil_div(I, 256), 8), (unsigned)O), dim3(256), 0,
                           s, dout, out, x, (long long)x_stride, dw, db, B, I, O, wmul, bmul, slope);
        return spk::check_launch("fc_bwd_weight_kernel");
    }
    return SPK_OK;
}

}  // extern "C"
