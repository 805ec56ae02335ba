// Fully connected + LeakyReLU for tiny batch (mapping network, style affines): a weight-streaming
// GEMV-like kernel.  One wave per output row: the 64 lanes read the row of W with 16-byte loads
// (coalesced, each weight byte leaves HBM once), keep BT batch accumulators in registers and finish
// with a wave butterfly.  HBM/L2-bound: 2*I*O*B flops over 4*I*O weight bytes.
// Replaces FC.forward (styleganv1.py:489-495) -- see include/spk.h.
#include "spk_common.hpp"

namespace {

constexpr int FC_BT = 8;       // batch rows accumulated per pass over the weight row
constexpr int FC_WAVES = 4;    // waves (= output rows) per workgroup

template <bool VEC>
__global__ __launch_bounds__(FC_WAVES * 64) void fc_kernel(const float* __restrict__ x, long long x_stride,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, long long out_stride, int B, int I,
                                                          int O, float wmul, float bmul, float slope) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * FC_WAVES + (threadIdx.x >> 6);
    if (o >= O) return;
    const float* wr = w + (size_t)o * I;
    const float bb = bias ? bias[o] * bmul : 0.f;
    for (int b0 = 0; b0 < B; b0 += FC_BT) {
        float acc[FC_BT];
#pragma unroll
        for (int b = 0; b < FC_BT; ++b) acc[b] = 0.f;
        if (VEC) {
            for (int i = lane * 4; i < I; i += 256) {
                const float4 wv = *reinterpret_cast<const float4*>(wr + i);
#pragma unroll
                for (int b = 0; b < FC_BT; ++b) {
                    if (b0 + b < B) {
                        const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)(b0 + b) * x_stride + i);
                        acc[b] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
                    }
                }
            }
        } else {
            for (int i = lane; i < I; i += 64) {
                const float wv = wr[i];
#pragma unroll
                for (int b = 0; b < FC_BT; ++b)
                    if (b0 + b < B) acc[b] += wv * x[(size_t)(b0 + b) * x_stride + i];
            }
        }
#pragma unroll
        for (int b = 0; b < FC_BT; ++b) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[b] += __shfl_xor(acc[b], off);
        }
        // lane b publishes batch row b0+b
        float v = 0.f;
#pragma unroll
        for (int b = 0; b < FC_BT; ++b)
            if (lane == b) v = acc[b];
        if (lane < FC_BT && b0 + lane < B) {
            v = v * wmul + bb;
            v = v > 0.f ? v : v * slope;
            out[(size_t)(b0 + lane) * out_stride + o] = v;
        }
    }
}

}  // namespace

extern "C" int spk_fc_fwd(const float* x, int64_t x_stride, const float* w, const float* bias, float* out,
                          int64_t out_stride, int B, int I, int O, float wmul, float bmul, float slope, void* stream) {
    SPK_REQUIRE(x && w && out, "fc: null pointer");
    SPK_REQUIRE(B > 0 && I > 0 && O > 0, "fc: bad shape B=%d I=%d O=%d", B, I, O);
    SPK_REQUIRE(x_stride >= I && out_stride >= O, "fc: row stride smaller than row");
    const bool vec = (I % 4 == 0) && (x_stride % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0);
    dim3 grid((unsigned)spk::ceil_div(O, FC_WAVES)), block(FC_WAVES * 64);
    if (vec)
        hipLaunchKernelGGL(fc_kernel<true>, grid, block, 0, (hipStream_t)stream, x, (long long)x_stride, w, bias, out,
                           (long long)out_stride, B, I, O, wmul, bmul, slope);
    else
        hipLaunchKernelGGL(fc_kernel<false>, grid, block, 0, (hipStream_t)stream, x, (long long)x_stride, w, bias, out,
                           (long long)out_stride, B, I, O, wmul, bmul, slope);
    return spk::check_launch("fc_kernel");
}
