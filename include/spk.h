/*
 * spk.h -- C ABI of libspk_hip.so: the MI355X (gfx950) kernels behind the SPEAK generative hot path.
 *
 * The reference (johndpope/SPEAK-hack) has no FFI layer: its hot path is a chain of stock ATen
 * calls made from Python nn.Modules.  Each entry point below replaces one such call site (or a
 * fused run of them); the citation after "replaces:" is the reference file:line under
 * /root/reference.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller
 *     (fp32, contiguous, NCHW for 4-D tensors) unless the parameter name ends in _host;
 *   - nothing is allocated, freed or synchronised inside; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream) and the call returns at once,
 *     so every entry point may be captured into a hipGraph;
 *   - returns 0 on success, a negative SPK_E* code otherwise; never throws, never aborts.
 *     spk_last_error() returns a thread-local message for the last failure on this thread;
 *   - re-entrant and thread-safe (no mutable global state besides a one-time attribute cache).
 */
#ifndef SPK_H_
#define SPK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPK_OK            0
#define SPK_EINVAL       -1   /* bad argument (shape, null pointer, unsupported geometry) */
#define SPK_ELAUNCH      -2   /* HIP reported a launch error */
#define SPK_EUNSUPPORTED -3

/* epilogue flags of spk_conv3x3_desc.flags */
#define SPK_EPI_BIAS   1u
#define SPK_EPI_NOISE  2u
#define SPK_EPI_LRELU  4u
#define SPK_EPI_STYLE  8u
#define SPK_CONV_UPSAMPLE2X 16u   /* input is [B,Cin,H/2,W/2]; bilinear x2 is applied while staging */
#define SPK_EPI_ACCUM  32u        /* y += result (used by weight-gradient / residual paths) */

const char* spk_version(void);
const char* spk_last_error(void);

/* ---- 3x3 convolution, stride 1, zero pad 1, fp32 on the f32 MFMA pipe ------------------------
 * y[b,co,h,w] = epi( sum_{ci,ky,kx} w[co,ci,ky,kx] * xin[b,ci,h+ky-1,w+kx-1] )
 * epi(v) = style( lrelu( v + bias[co] + noise_w[co]*noise[b,0,h,w] ) ),
 * style(v) = v*(style[b*style_stride + co] + 1) + style[b*style_stride + Cout + co]
 * Each stage of epi is enabled by its SPK_EPI_* flag.  With SPK_CONV_UPSAMPLE2X, xin is the
 * bilinear x2 (align_corners=False) upsampling of x[B,Cin,H/2,W/2], formed on the fly.
 * replaces: styleganv1.py:624-628 (upsample, conv1, noise1, leaky_relu, style_mod1) and
 *           styleganv1.py:630-633 (conv2, noise2, leaky_relu, style_mod2) -- one launch each;
 *           stylegan.py:45-46 (WSConv2d: in_scale folds the x*scale pre-multiply).
 * Weights must first be packed for the chosen tile config with spk_conv3x3_pack_weights. */
typedef struct spk_conv3x3_desc {
    const float* x;          /* [B,Cin,H,W] (or [B,Cin,H/2,W/2] with SPK_CONV_UPSAMPLE2X) */
    const float* w_packed;   /* from spk_conv3x3_pack_weights, same `config` */
    const float* bias;       /* [Cout] or NULL */
    const float* noise_w;    /* [Cout] or NULL */
    const float* noise;      /* [B,1,H,W] or NULL */
    const float* style;      /* row b at style + b*style_stride: [s0(Cout) | s1(Cout)] or NULL */
    float*       y;          /* [B,Cout,H,W] */
    int32_t B, Cin, Cout, H, W;   /* H, W: OUTPUT spatial size */
    int32_t style_stride;    /* in floats */
    uint32_t flags;
    float lrelu_slope;
    float in_scale;          /* multiplies the result of the contraction before bias (1.0 = none) */
    int32_t config;          /* tile config id, 0..spk_conv3x3_num_configs()-1, or -1 = auto */
    int32_t ksplit;          /* slices of the input-channel range (split-K); 0 = auto, 1 = none */
    void*   workspace;       /* device scratch for split-K partial sums (may be NULL if not needed) */
    int64_t workspace_bytes; /* its size; see spk_conv3x3_workspace_bytes */
} spk_conv3x3_desc;

int spk_conv3x3_num_configs(void);
/* tile config chosen by the heuristic for this problem (what `config = -1` resolves to) */
int spk_conv3x3_pick_config(int B, int Cin, int Cout, int H, int W);
/* CO_T / CI_T / PIX_T of a config (any out pointer may be NULL) */
int spk_conv3x3_config_info(int config, int* co_tile, int* ci_tile, int* pix_tile);
/* number of floats of the packed image of a [Cout,Cin,3,3] weight for `config` */
int64_t spk_conv3x3_packed_floats(int config, int Cin, int Cout);
/* bytes of scratch spk_conv3x3_fwd needs for this problem (0 when it will not split K; <0 = bad args).
 * config = -1 / ksplit = 0 ask for the library's own choices. */
int64_t spk_conv3x3_workspace_bytes(int config, int ksplit, int B, int Cin, int Cout, int H, int W);
/* w[Cout,Cin,3,3] -> packed [co_tile][ci_chunk][tap][ci][co] (zero padded).
 * transpose_flip != 0 packs the data-gradient operator instead: w'[ci,co,ky,kx] = w[co,ci,2-ky,2-kx]
 * (then the packed image is that of a [Cin,Cout,3,3] weight).
 * replaces: nothing in the reference (layout change private to this library). */
int spk_conv3x3_pack_weights(const float* w, float* w_packed, int Cin, int Cout, int config,
                             int transpose_flip, void* stream);
int spk_conv3x3_fwd(const spk_conv3x3_desc* desc, void* stream);

/* ---- fully connected + LeakyReLU ---------------------------------------------------------------
 * out[b,o] = act( wmul * sum_i x[b*x_stride + i] * w[o*I + i] + bmul * bias[o] ),
 * act(v) = v > 0 ? v : slope*v  (slope = 1 => identity).
 * replaces: styleganv1.py:489-495 (FC.forward: F.linear with runtime w_lrmul/b_lrmul + leaky_relu),
 *           used by the mapping stack :513-518,:532 and by every ApplyStyle :461,:464;
 *           stylegan.py:20-21 (WSLinear). */
int spk_fc_fwd(const float* x, int64_t x_stride, const float* w, const float* bias, float* out,
               int64_t out_stride, int B, int I, int O, float wmul, float bmul, float slope, void* stream);

/* ---- bias + noise + style (decoder prologue; stand-alone ApplyNoise / ApplyStyle) ------------------
 * y[b,c,p] = (x[b*x_batch_stride + c*HW + p] + bias[c] + noise_w[c]*noise[b,p]) * (s0[b,c]+1) + s1[b,c]
 * x_batch_stride = 0 broadcasts one [C,HW] constant over the batch; bias / noise / style may be NULL.
 * replaces: styleganv1.py:596-599 (const_input.expand + bias, noise_input1, style_mod) in one launch;
 *           styleganv1.py:453-456 (ApplyNoise.forward) and :463-468 (ApplyStyle.forward) when those
 *           modules are called on their own. */
int spk_bias_noise_style_fwd(const float* x, int64_t x_batch_stride, const float* bias, const float* noise_w,
                             const float* noise, const float* style, int64_t style_stride, float* y, int B, int C,
                             int HW, void* stream);

/* ---- 1x1 convolution with few output channels (toRGB) -------------------------------------------
 * y[b,o,p] = sum_c w[o*C + c] * x[b,c,p] * in_scale + bias[o],  O <= 4.  HBM-bound streaming kernel.
 * replaces: styleganv1.py:607 (to_rgb = nn.Conv2d(64,3,1)); stylegan.py:138-140,175-176 (rgb layers). */
int spk_conv1x1_small_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int O,
                          int64_t HW, float in_scale, void* stream);

/* ---- bilinear x2 upsampling (align_corners = False) -----------------------------------------------
 * replaces: styleganv1.py:621,624 (nn.Upsample) / stylegan.py:168 (F.interpolate) when used un-fused. */
int spk_upsample2x_bilinear_fwd(const float* x, float* y, int64_t planes, int Hin, int Win, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPK_H_ */
