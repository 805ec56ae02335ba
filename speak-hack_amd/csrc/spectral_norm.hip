// Spectral normalisation of every wrapped layer of StyleDiscriminator in a handful of grouped launches.
//
// The reference wraps 16 layers with torch.nn.utils.spectral_norm (styleganv1.py:644-657,662-672): before EVERY forward each
// layer runs one power iteration -- v = normalize(W^T u), u = normalize(W v), sigma = u^T W v, W_hat = W / sigma on its
// [Cout, Cin*k*k] matrix -- about a dozen tiny ATen launches per layer, ~200 per discriminator pass and six passes per
// training iteration (train.py:160-177): a thousand launches that move 76 MB of weights three times each.  Here one call
// handles all layers: two passes over the weights (W^T u by row chunks with a fixed-order chunk sum, then W v), the
// normalisations, and the division written straight into the W_hat tensors.  No atomics: every sum has a fixed order, so
// replicas that start from the same u, v stay bit-identical (no buffer broadcast needed in data-parallel training).
// Backward (autograd of W / sigma with u, v constant, as torch's hook has it): dW = (G - <G, W_hat> u v^T) / sigma.
#include "spk_common.hpp"

#include <algorithm>

namespace {

constexpr int RC = 32;          // rows per chunk of the W^T u pass

struct SnGroups {
    spk_sn_group g[SPK_SN_MAX_GROUPS];
    int blk_a[SPK_SN_MAX_GROUPS + 1];      // block prefix of pass A: (row chunks) x (256-column blocks)
    int blk_c[SPK_SN_MAX_GROUPS + 1];      // block prefix over 256-column blocks
    int blk_r[SPK_SN_MAX_GROUPS + 1];      // block prefix over 4-row blocks (wave per row)
    int blk_e[SPK_SN_MAX_GROUPS + 1];      // block prefix over 1024-element blocks of the flat matrix
    long long off_p[SPK_SN_MAX_GROUPS];    // float offsets of the group's areas in the workspace
    long long off_t[SPK_SN_MAX_GROUPS];
    long long off_s[SPK_SN_MAX_GROUPS];
    long long off_q[SPK_SN_MAX_GROUPS];    // per-column-block sums of squares of t / per-block partial dots (backward)
    int n;
};

__device__ __forceinline__ int sn_find(const int* prefix, int n, int b) {
    int gi = 0;
    while (gi + 1 < n && b >= prefix[gi + 1]) ++gi;
    return gi;
}

__device__ __forceinline__ float sn_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// A: partial[chunk][c] = sum_{r in chunk} W[r][c] * u[r]
__global__ __launch_bounds__(256) void sn_wtu_partial_kernel(const SnGroups a, float* __restrict__ ws) {
    const int gi = sn_find(a.blk_a, a.n, blockIdx.x);
    const spk_sn_group& q = a.g[gi];
    const int cblocks = (q.C + 255) / 256;
    const int b = blockIdx.x - a.blk_a[gi];
    const int chunk = b / cblocks, c = (b % cblocks) * 256 + threadIdx.x;
    if (c >= q.C) return;
    const int r0 = chunk * RC, r1 = min(q.R, r0 + RC);
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += q.w[(size_t)r * q.C + c] * q.u[r];
    ws[a.off_p[gi] + (size_t)chunk * q.C + c] = acc;
}

// A2: t[c] = sum_chunk partial[chunk][c] (fixed order); per block the sum of squares of its 256 columns
__global__ __launch_bounds__(256) void sn_wtu_finish_kernel(const SnGroups a, float* __restrict__ ws) {
    __shared__ float red[4];
    const int gi = sn_find(a.blk_c, a.n, blockIdx.x);
    const spk_sn_group& q = a.g[gi];
    const int cb = blockIdx.x - a.blk_c[gi], c = cb * 256 + threadIdx.x;
    const int chunks = (q.R + RC - 1) / RC;
    float t = 0.f;
    if (c < q.C) {
        for (int k = 0; k < chunks; ++k) t += ws[a.off_p[gi] + (size_t)k * q.C + c];
        ws[a.off_t[gi] + c] = t;
    }
    const float ss = sn_block_sum(t * t, red);
    if (threadIdx.x == 0) ws[a.off_q[gi] + cb] = ss;
}

// B: s[r] = sum_c W[r][c] * vhat[c], vhat = t / max(|t|, eps) (power iteration) or the stored v; one wave per row
__global__ __launch_bounds__(256) void sn_wv_kernel(const SnGroups a, float* __restrict__ ws, int power_iter, float eps) {
    const int gi = sn_find(a.blk_r, a.n, blockIdx.x);
    const spk_sn_group& q = a.g[gi];
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x - a.blk_r[gi]) * 4 + (threadIdx.x >> 6);
    float inv = 1.f;
    const float* vec = q.v;
    if (power_iter) {
        const int cblocks = (q.C + 255) / 256;
        float ss = 0.f;
        for (int k = 0; k < cblocks; ++k) ss += ws[a.off_q[gi] + k];       // same order in every wave: identical inv
        inv = 1.f / fmaxf(sqrtf(ss), eps);
        vec = ws + a.off_t[gi];
    }
    if (r >= q.R) return;
    const float* wr = q.w + (size_t)r * q.C;
    float acc = 0.f;
    for (int c = lane; c < q.C; c += 64) acc += wr[c] * (vec[c] * inv);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) ws[a.off_s[gi] + r] = acc;
}

// C: one block per layer: (power iteration) v <- vhat, u <- s / max(|s|, eps); sigma = <u, s>
__global__ __launch_bounds__(256) void sn_finish_kernel(const SnGroups a, float* __restrict__ ws, int power_iter, float eps) {
    __shared__ float red[4];
    const int gi = blockIdx.x;
    const spk_sn_group& q = a.g[gi];
    const float* s = ws + a.off_s[gi];
    if (power_iter) {
        const int cblocks = (q.C + 255) / 256;
        float ss = 0.f;
        for (int k = 0; k < cblocks; ++k) ss += ws[a.off_q[gi] + k];
        const float inv = 1.f / fmaxf(sqrtf(ss), eps);
        for (int c = threadIdx.x; c < q.C; c += 256) q.v[c] = ws[a.off_t[gi] + c] * inv;
        float p = 0.f;
        for (int r = threadIdx.x; r < q.R; r += 256) p += s[r] * s[r];
        const float ns = sn_block_sum(p, red);
        const float invs = 1.f / fmaxf(sqrtf(ns), eps);
        for (int r = threadIdx.x; r < q.R; r += 256) q.u[r] = s[r] * invs;
        __syncthreads();
    }
    float d = 0.f;
    for (int r = threadIdx.x; r < q.R; r += 256) d += q.u[r] * s[r];
    const float sigma = sn_block_sum(d, red);
    if (threadIdx.x == 0) *q.sigma = sigma;
}

// D: W_hat = W / sigma
__global__ __launch_bounds__(256) void sn_scale_kernel(const SnGroups a) {
    const int gi = sn_find(a.blk_e, a.n, blockIdx.x);
    const spk_sn_group& q = a.g[gi];
    const float sigma = *q.sigma;
    const size_t n = (size_t)q.R * q.C, i0 = (size_t)(blockIdx.x - a.blk_e[gi]) * 1024 + threadIdx.x * 4;
    if ((n & 3) == 0 && i0 + 3 < n) {
        const float4 w = *reinterpret_cast<const float4*>(q.w + i0);
        *reinterpret_cast<float4*>(q.w_hat + i0) = make_float4(w.x / sigma, w.y / sigma, w.z / sigma, w.w / sigma);
    } else {
        for (size_t i = i0; i < std::min(n, i0 + 4); ++i) q.w_hat[i] = q.w[i] / sigma;
    }
}

// backward E: per 1024-element block the partial dot <G, W>
__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const SnGroups a, float* __restrict__ ws) {
    __shared__ float red[4];
    const int gi = sn_find(a.blk_e, a.n, blockIdx.x);
    const spk_sn_group& q = a.g[gi];
    const int b = blockIdx.x - a.blk_e[gi];
    const size_t n = (size_t)q.R * q.C, i0 = (size_t)b * 1024 + threadIdx.x * 4;
    float d = 0.f;
    if ((n & 3) == 0 && i0 + 3 < n) {                                               // (16-byte accesses; the same summation order)
        const float4 gq = *reinterpret_cast<const float4*>(q.w_hat + i0), wq = *reinterpret_cast<const float4*>(q.w + i0);
        d += gq.x * wq.x; d += gq.y * wq.y; d += gq.z * wq.z; d += gq.w * wq.w;
    } else {
        for (size_t i = i0; i < std::min(n, i0 + 4); ++i) d += q.w_hat[i] * q.w[i];     // w_hat slot = the incoming gradient G
    }
    const float tot = sn_block_sum(d, red);
    if (threadIdx.x == 0) ws[a.off_q[gi] + b] = tot;
}

// backward F: dW = (G - (<G, W> / sigma) u v^T) / sigma      (<G, W_hat> = <G, W> / sigma)
__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const SnGroups a, const float* __restrict__ ws) {
    __shared__ float red[4];
    const int gi = sn_find(a.blk_e, a.n, blockIdx.x);
    const spk_sn_group& q = a.g[gi];
    const int nb = a.blk_e[gi + 1] - a.blk_e[gi];
    float p = 0.f;
    for (int k = threadIdx.x; k < nb; k += 256) p += ws[a.off_q[gi] + k];          // every block: the same fixed-order total
    const float dot = sn_block_sum(p, red);
    const float sigma = *q.sigma;
    const float coef = dot / sigma;
    const size_t n = (size_t)q.R * q.C, i0 = (size_t)(blockIdx.x - a.blk_e[gi]) * 1024 + threadIdx.x * 4;
    if ((q.C & 3) == 0 && i0 + 3 < n) {              // rows are a multiple of 4 floats: the four elements share a row, 16-byte accesses
        const int r = (int)(i0 / q.C), c = (int)(i0 - (size_t)r * q.C);
        const float4 gq = *reinterpret_cast<const float4*>(q.w_hat + i0), vq = *reinterpret_cast<const float4*>(q.v + c);
        const float cu = coef * q.u[r];
        float4 d4 = make_float4((gq.x - cu * vq.x) / sigma, (gq.y - cu * vq.y) / sigma, (gq.z - cu * vq.z) / sigma, (gq.w - cu * vq.w) / sigma);
        float4* o = reinterpret_cast<float4*>(q.dw + i0);
        if (q.accumulate) { const float4 old = *o; d4.x += old.x; d4.y += old.y; d4.z += old.z; d4.w += old.w; }
        *o = d4;
        return;
    }
    for (size_t i = i0; i < std::min(n, i0 + 4); ++i) {
        const int r = (int)(i / q.C), c = (int)(i - (size_t)r * q.C);
        const float d = (q.w_hat[i] - coef * q.u[r] * q.v[c]) / sigma;
        q.dw[i] = q.accumulate ? q.dw[i] + d : d;
    }
}

int sn_fill(SnGroups& a, const spk_sn_group* groups, int n, int64_t* ws_floats) {
    a.n = n;
    a.blk_a[0] = a.blk_c[0] = a.blk_r[0] = a.blk_e[0] = 0;
    long long off = 0;
    for (int i = 0; i < n; ++i) {
        const spk_sn_group& q = groups[i];
        if (!(q.w && q.u && q.v && q.sigma && q.R > 0 && q.C > 0)) return spk::fail(SPK_EINVAL, "spectral_norm: group %d: bad arguments", i);
        a.g[i] = q;
        const int chunks = spk::ceil_div(q.R, RC), cblocks = spk::ceil_div(q.C, 256);
        const long long elems = (long long)q.R * q.C;
        a.blk_a[i + 1] = a.blk_a[i] + chunks * cblocks;
        a.blk_c[i + 1] = a.blk_c[i] + cblocks;
        a.blk_r[i + 1] = a.blk_r[i] + spk::ceil_div(q.R, 4);
        a.blk_e[i + 1] = a.blk_e[i] + (int)((elems + 1023) / 1024);
        a.off_p[i] = off; off += (long long)chunks * q.C;
        a.off_t[i] = off; off += q.C;
        a.off_s[i] = off; off += q.R;
        a.off_q[i] = off; off += std::max<long long>(cblocks, (elems + 1023) / 1024);
        off = (off + 3) & ~3ll;
    }
    *ws_floats = off;
    return SPK_OK;
}

}  // namespace

extern "C" {

int64_t spk_spectral_norm_workspace_bytes(const spk_sn_group* groups, int n_groups) {
    if (!groups || n_groups <= 0 || n_groups > SPK_SN_MAX_GROUPS) return -1;
    SnGroups a;
    int64_t f = 0;
    if (sn_fill(a, groups, n_groups, &f) != SPK_OK) return -1;
    return f * (int64_t)sizeof(float);
}

int spk_spectral_norm_grouped(const spk_sn_group* groups, int n_groups, int power_iteration, float eps, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    SPK_REQUIRE(groups && n_groups > 0 && n_groups <= SPK_SN_MAX_GROUPS, "spectral_norm: 1..%d groups", SPK_SN_MAX_GROUPS);
    SnGroups a;
    int64_t f = 0;
    int rc = sn_fill(a, groups, n_groups, &f);
    if (rc != SPK_OK) return rc;
    SPK_REQUIRE(workspace && workspace_bytes >= f * (int64_t)sizeof(float), "spectral_norm: needs a %lld-byte workspace", (long long)(f * 4));
    for (int i = 0; i < n_groups; ++i) SPK_REQUIRE(groups[i].w_hat, "spectral_norm: group %d: null w_hat", i);
    float* ws = static_cast<float*>(workspace);
    hipStream_t s = (hipStream_t)stream;
    if (power_iteration) {
        hipLaunchKernelGGL(sn_wtu_partial_kernel, dim3((unsigned)a.blk_a[n_groups]), dim3(256), 0, s, a, ws);
        hipLaunchKernelGGL(sn_wtu_finish_kernel, dim3((unsigned)a.blk_c[n_groups]), dim3(256), 0, s, a, ws);
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3((unsigned)a.blk_r[n_groups]), dim3(256), 0, s, a, ws, power_iteration ? 1 : 0, eps);
    hipLaunchKernelGGL(sn_finish_kernel, dim3((unsigned)n_groups), dim3(256), 0, s, a, ws, power_iteration ? 1 : 0, eps);
    hipLaunchKernelGGL(sn_scale_kernel, dim3((unsigned)a.blk_e[n_groups]), dim3(256), 0, s, a);
    return spk::check_launch("spectral_norm kernels");
}

int spk_spectral_norm_bwd_grouped(const spk_sn_group* groups, int n_groups, void* workspace, int64_t workspace_bytes, void* stream) {
    SPK_REQUIRE(groups && n_groups > 0 && n_groups <= SPK_SN_MAX_GROUPS, "spectral_norm_bwd: 1..%d groups", SPK_SN_MAX_GROUPS);
    SnGroups a;
    int64_t f = 0;
    int rc = sn_fill(a, groups, n_groups, &f);
    if (rc != SPK_OK) return rc;
    SPK_REQUIRE(workspace && workspace_bytes >= f * (int64_t)sizeof(float), "spectral_norm_bwd: needs a %lld-byte workspace", (long long)(f * 4));
    for (int i = 0; i < n_groups; ++i) SPK_REQUIRE(groups[i].w_hat && groups[i].dw, "spectral_norm_bwd: group %d: null gradient pointer", i);
    float* ws = static_cast<float*>(workspace);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3((unsigned)a.blk_e[n_groups]), dim3(256), 0, s, a, ws);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3((unsigned)a.blk_e[n_groups]), dim3(256), 0, s, a, ws);
    return spk::check_launch("spectral_norm backward kernels");
}

}  // extern "C"
