// Fully connected + LeakyReLU for tiny batch (mapping network, style affines): a weight-streaming
// GEMV-like kernel.  One wave per output row: the 64 lanes read the row of W with 16-byte loads
// (coalesced, each weight byte leaves HBM once), keep BT batch accumulators in registers and finish
// with a wave butterfly.  HBM/L2-bound: 2*I*O*B flops over 4*I*O weight bytes.
// Replaces FC.forward (styleganv1.py:489-495) -- see include/spk.h.
#include "spk_common.hpp"

namespace {

constexpr int FC_BT = 8;       // batch rows accumulated per pass over the weight row
constexpr int FC_WAVES = 4;    // waves (= output rows) per workgroup

// The mapping network's and the style affines' 512-wide rows: both weight loads and all 2 x FC_BT latent loads of a lane in flight
// at once (the generic loop runs its two iterations one memory latency after the other; such a layer is latency, not bandwidth:
// 1 MB of weights).  One body for fc_kernel and fc_grouped_kernel: their results are bitwise equal.
__device__ __forceinline__ void fc_dot512(const float* __restrict__ wr, const float* __restrict__ x, long long x_stride, int b0, int B,
                                          int lane, float (&acc)[FC_BT]) {
    const float4 w0 = *reinterpret_cast<const float4*>(wr + lane * 4);
    const float4 w1 = *reinterpret_cast<const float4*>(wr + 256 + lane * 4);
    float4 x0[FC_BT], x1[FC_BT];
#pragma unroll
    for (int b = 0; b < FC_BT; ++b) {
        const float* xr = x + (size_t)min(b0 + b, B - 1) * x_stride + lane * 4;
        x0[b] = *reinterpret_cast<const float4*>(xr);
        x1[b] = *reinterpret_cast<const float4*>(xr + 256);
    }
#pragma unroll
    for (int b = 0; b < FC_BT; ++b) {
        acc[b] += w0.x * x0[b].x + w0.y * x0[b].y + w0.z * x0[b].z + w0.w * x0[b].w;
        acc[b] += w1.x * x1[b].x + w1.y * x1[b].y + w1.z * x1[b].z + w1.w * x1[b].w;
    }
}

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
        if (VEC && I == 512) {
            fc_dot512(wr, x, x_stride, b0, B, lane, acc);
        } else if (VEC) {
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


// Long rows (the 6144 -> 512 first mapping layer: 12.6 MB of weights behind only 512 rows): the whole workgroup takes
// ONE output row, every lane keeps FC_WIDE_U 16-byte weight loads in flight, and the four wave sums meet in LDS --
// 4x the workgroups and 4-6x the bytes in flight of the wave-per-row kernel.
constexpr int FC_WIDE_U = 6;
__global__ __launch_bounds__(256) void fc_wide_kernel(const float* __restrict__ x, long long x_stride, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, long long out_stride,
                                                     int B, int I, int O, float wmul, float bmul, float slope) {
    __shared__ float red[4][FC_BT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x;
    const float* wr = w + (size_t)o * I;
    for (int b0 = 0; b0 < B; b0 += FC_BT) {
        float acc[FC_BT];
#pragma unroll
        for (int b = 0; b < FC_BT; ++b) acc[b] = 0.f;
        for (int i0 = tid * 4; i0 < I; i0 += 1024 * FC_WIDE_U) {
            float4 wv[FC_WIDE_U];
#pragma unroll
            for (int u = 0; u < FC_WIDE_U; ++u) {
                const int i = i0 + u * 1024;
                wv[u] = i < I ? *reinterpret_cast<const float4*>(wr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < FC_WIDE_U; ++u) {
                const int i = i0 + u * 1024;
                if (i < I) {
#pragma unroll
                    for (int b = 0; b < FC_BT; ++b) {
                        if (b0 + b < B) {
                            const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)(b0 + b) * x_stride + i);
                            acc[b] += wv[u].x * xv.x + wv[u].y * xv.y + wv[u].z * xv.z + wv[u].w * xv.w;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < FC_BT; ++b) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[b] += __shfl_xor(acc[b], off);
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < FC_BT; ++b) red[wave][b] = acc[b];
        }
        __syncthreads();
        if (tid < FC_BT && b0 + tid < B) {
            float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            v = v * wmul + (bias ? bias[o] * bmul : 0.f);
            v = v > 0.f ? v : v * slope;
            out[(size_t)(b0 + tid) * out_stride + o] = v;
        }
    }
}

// The same long rows, FOUR output rows per workgroup (I = 1024 U, U <= FC_WIDE_U, O % 4 == 0): all 4 U weight loads of a lane are
// issued first (one HBM latency for the whole 12.6 MB layer), then U rounds of FC_BT latent loads serve four rows each -- the
// one-row kernel issued its latent loads under a branch per chunk, six L2 latencies in a row, and re-read the latents per row
// (24 us for the 6144 -> 512 layer; this one: ~8).  Same per-lane summation order as fc_wide_kernel.
__global__ __launch_bounds__(256) void fc_wide4_kernel(const float* __restrict__ x, long long x_stride, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, long long out_stride,
                                                      int B, int I, int O, float wmul, float bmul, float slope) {
    __shared__ float red[4][4][FC_BT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o0 = blockIdx.x * 4;
    const int U = I >> 10;
    float4 wv[4][FC_WIDE_U];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < FC_WIDE_U; ++u)
            wv[r][u] = *reinterpret_cast<const float4*>(w + (size_t)(o0 + r) * I + tid * 4 + min(u, U - 1) * 1024);
    for (int b0 = 0; b0 < B; b0 += FC_BT) {
        float acc[4][FC_BT];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < FC_BT; ++b) acc[r][b] = 0.f;
#pragma unroll
        for (int u = 0; u < FC_WIDE_U; ++u) {
            if (u < U) {
                float4 xv[FC_BT];
#pragma unroll
                for (int b = 0; b < FC_BT; ++b)
                    xv[b] = *reinterpret_cast<const float4*>(x + (size_t)min(b0 + b, B - 1) * x_stride + tid * 4 + u * 1024);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int b = 0; b < FC_BT; ++b)
                        acc[r][b] += wv[r][u].x * xv[b].x + wv[r][u].y * xv[b].y + wv[r][u].z * xv[b].z + wv[r][u].w * xv[b].w;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < FC_BT; ++b) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) acc[r][b] += __shfl_xor(acc[r][b], off);
            }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < FC_BT; ++b) red[wave][r][b] = acc[r][b];
        }
        __syncthreads();
        if (tid < 4 * FC_BT) {
            const int r = tid / FC_BT, b = tid % FC_BT;
            if (b0 + b < B) {
                float v = (red[0][r][b] + red[1][r][b]) + (red[2][r][b] + red[3][r][b]);
                v = v * wmul + (bias ? bias[o0 + r] * bmul : 0.f);
                v = v > 0.f ? v : v * slope;
                out[(size_t)(b0 + b) * out_stride + o0 + r] = v;
            }
        }
    }
}

// Several independent FCs on one launch (the 13 style affines of a decoder step all depend only on the dlatents):
// same wave-per-row body as fc_kernel, the (group, row) pair comes from a prefix table in the kernel arguments.
struct FcGroups {
    spk_fc_group g[SPK_FC_MAX_GROUPS];
    int row_start[SPK_FC_MAX_GROUPS + 1];   // in units of FC_WAVES-row blocks
    int n;
};

__global__ __launch_bounds__(FC_WAVES * 64) void fc_grouped_kernel(const FcGroups a, int B) {
    int gi = 0;
    while (gi + 1 < a.n && (int)blockIdx.x >= a.row_start[gi + 1]) ++gi;
    const spk_fc_group& g = a.g[gi];
    const int lane = threadIdx.x & 63;
    const int o = ((int)blockIdx.x - a.row_start[gi]) * FC_WAVES + (threadIdx.x >> 6);
    if (o >= g.O) return;
    const float* wr = g.w + (size_t)o * g.I;
    const float bb = g.bias ? g.bias[o] * g.bmul : 0.f;
    for (int b0 = 0; b0 < B; b0 += FC_BT) {
        float acc[FC_BT];
#pragma unroll
        for (int b = 0; b < FC_BT; ++b) acc[b] = 0.f;
        if (g.I == 512) {
            fc_dot512(wr, g.x, g.x_stride, b0, B, lane, acc);
        } else {
            for (int i = lane * 4; i < g.I; i += 256) {
                const float4 wv = *reinterpret_cast<const float4*>(wr + i);
#pragma unroll
                for (int b = 0; b < FC_BT; ++b) {
                    if (b0 + b < B) {
                        const float4 xv = *reinterpret_cast<const float4*>(g.x + (size_t)(b0 + b) * g.x_stride + i);
                        acc[b] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < FC_BT; ++b) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[b] += __shfl_xor(acc[b], off);
        }
        float v = 0.f;
#pragma unroll
        for (int b = 0; b < FC_BT; ++b)
            if (lane == b) v = acc[b];
        if (lane < FC_BT && b0 + lane < B) {
            v = v * g.wmul + bb;
            v = v > 0.f ? v : v * g.slope;
            g.out[(size_t)(b0 + lane) * g.out_stride + o] = v;
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
    if (vec && I >= 2048 && I % 1024 == 0 && I <= 1024 * FC_WIDE_U && O % 4 == 0) {
        hipLaunchKernelGGL(fc_wide4_kernel, dim3((unsigned)(O / 4)), dim3(256), 0, (hipStream_t)stream, x, (long long)x_stride, w, bias, out,
                           (long long)out_stride, B, I, O, wmul, bmul, slope);
        return spk::check_launch("fc_wide4_kernel");
    }
    if (vec && I >= 2048) {
        hipLaunchKernelGGL(fc_wide_kernel, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, x, (long long)x_stride, w, bias, out,
                           (long long)out_stride, B, I, O, wmul, bmul, slope);
        return spk::check_launch("fc_wide_kernel");
    }
    dim3 grid((unsigned)spk::ceil_div(O, FC_WAVES)), block(FC_WAVES * 64);
    if (vec)
        hipLaunchKernelGGL(fc_kernel<true>, grid, block, 0, (hipStream_t)stream, x, (long long)x_stride, w, bias, out,
                           (long long)out_stride, B, I, O, wmul, bmul, slope);
    else
        hipLaunchKernelGGL(fc_kernel<false>, grid, block, 0, (hipStream_t)stream, x, (long long)x_stride, w, bias, out,
                           (long long)out_stride, B, I, O, wmul, bmul, slope);
    return spk::check_launch("fc_kernel");
}

extern "C" int spk_fc_grouped_fwd(const spk_fc_group* groups, int n_groups, int B, void* stream) {
    SPK_REQUIRE(groups && n_groups > 0 && n_groups <= SPK_FC_MAX_GROUPS && B > 0, "fc_grouped: bad arguments (1..%d groups)",
                SPK_FC_MAX_GROUPS);
    FcGroups a;
    a.n = n_groups;
    a.row_start[0] = 0;
    for (int i = 0; i < n_groups; ++i) {
        const spk_fc_group& g = groups[i];
        SPK_REQUIRE(g.x && g.w && g.out && g.I > 0 && g.O > 0 && g.x_stride >= g.I && g.out_stride >= g.O, "fc_grouped: group %d: bad shape", i);
        SPK_REQUIRE(g.I % 4 == 0 && g.x_stride % 4 == 0 && (uintptr_t)g.x % 16 == 0 && (uintptr_t)g.w % 16 == 0,
                    "fc_grouped: group %d: rows must be 16-byte aligned multiples of 4 floats", i);
        a.g[i] = g;
        a.row_start[i + 1] = a.row_start[i] + spk::ceil_div(g.O, FC_WAVES);
    }
    hipLaunchKernelGGL(fc_grouped_kernel, dim3((unsigned)a.row_start[n_groups]), dim3(FC_WAVES * 64), 0, (hipStream_t)stream, a, B);
    return spk::check_launch("fc_grouped_kernel");
}
