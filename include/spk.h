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

/* flags of spk_conv2d_desc.flags */
#define SPK_EPI_BIAS   1u
#define SPK_EPI_NOISE  2u
#define SPK_EPI_LRELU  4u
#define SPK_EPI_STYLE  8u
#define SPK_CONV_UPSAMPLE2X 16u      /* input is bilinearly upsampled x2 while staging (3x3 stride 1 only) */
#define SPK_EPI_ACCUM  32u           /* y += result */
#define SPK_EPI_STATS  64u           /* stats[0:Cout] += sum_bhw y, stats[Cout:2Cout] += sum_bhw y^2 (fp64); on a 3x3 stride-1
                                      * conv with a plain or SPK_CONV_IN_AFFINE_RELU input only (not UPSAMPLE2X / IN_BATCH_SCALE) */
#define SPK_CONV_IN_AFFINE_RELU 128u /* x' = max(x*in_scale[ci] + in_shift[ci], 0) applied while staging */
#define SPK_CONV_IN_BATCH_SCALE 256u /* x' = x * in_scale[b*Cin + ci] applied while staging (weight modulation); may be
                                      * combined with SPK_CONV_UPSAMPLE2X (configs 4-7) */
#define SPK_CONV_DGRAD_S2 1024u  /* kh = kw = 3, stride = 2: run the DATA GRADIENT of that conv instead: x = the output-side
                                  * gradient [B,Cin,Hin,Win], y = the input-side gradient [B,Cout,H,W] (H in {2Hin-1, 2Hin});
                                  * w_packed from spk_conv2d_pack_weights(w[Cin][Cout][3][3], transpose_flip = 2) for a
                                  * the tile config spk_conv2d_dgrad_s2_config(B, Cin, Cout, Hin, Win) names;
                                  * SPK_EPI_ACCUM is the only other flag.  Computed by output parity -- dx[2m+py, 2n+px]
                                  * needs 1/2/2/4 of the 9 taps.  Config 13: one kernel with exactly those taps (a wave owns
                                  * the four classes of its channels and pixels; 9 MFMAs per contraction pair).  Configs 0-3:
                                  * four zero-padded 2x2 kernels in one launch of the general kernel (16 taps executed). */
#define SPK_CONV_TRANSPOSE4X4_S2 2048u /* kh = kw = 4, stride = 2: the FORWARD of nn.ConvTranspose2d(Cin, Cout, 4, stride=2,
                                  * padding=1) -- the fused upscale of the legacy GBlock, replaces: styleganv1.py:231,258 --
                                  * x [B,Cin,Hin,Win] -> y [B,Cout,2Hin,2Win]; w_packed from spk_conv2d_pack_weights(w[Cin][Cout]
                                  * [4][4], kh = kw = 4, transpose_flip = 3), config in 0-3 (spk_conv2d_pick_config(2, 2, 1, B, Cin,
                                  * 4*Cout, Hin+1, Win+1)); bias [Cout]; SPK_EPI_BIAS / SPK_EPI_ACCUM only.  Every output pixel
                                  * (2m+py, 2n+px) takes exactly 2x2 of the 16 taps: four 2x2 kernels, one launch, interleaved. */
#define SPK_CONV_BF16X3 4096u    /* OPT-IN speed path for 3x3 stride-1 forward convs: operands split into bf16 hi + lo halves
                                  * (16 significant bits), three bf16 MFMAs per product into an fp32 accumulator -- 5.3x the
                                  * exact-f32 matrix rate at ~3e-5 rel-L2 through the whole decoder (csrc/conv3x3_bf16x3.hip).
                                  * w_packed from spk_conv2d_pack_weights_bf16x3; flags: BIAS / NOISE / LRELU / STYLE /
                                  * UPSAMPLE2X (+ UP_FIR1331) / IN_BATCH_SCALE (+ out_scale_bc), y_pre; config / ksplit ignored. */
#define SPK_CONV_WINOGRAD 16384u  /* 3x3 stride-1 pad-1 convs as Winograd F(2x2, 3x3) on the f32 MFMA pipe (csrc/conv3x3_wino_f32.hip): fp32
                                  * operands, products and accumulation, 16 multiplies per 2x2 output tile where the direct form
                                  * spends 36 -- the algorithm the reference's own backend (MIOpen under PyTorch-ROCm) runs for an
                                  * fp32 3x3 nn.Conv2d (styleganv1.py:615-616,625,630,662-672).  Differs from the direct form in
                                  * summation order and the +-1 / 0.5 transforms only (1e-6-class rel-L2 per layer).  Plain input
                                  * (no UPSAMPLE2X / IN_AFFINE_RELU / IN_BATCH_SCALE), ungrouped, H % 8 == 0, W % 32 == 0,
                                  * Cin % 8 == 0; flags BIAS / NOISE / LRELU / STYLE / ACCUM, y_pre, out_scale(_dev).
                                  * w_packed from spk_conv2d_pack_weights_wino; config / ksplit ignored. */
#define SPK_EPI_ACCUM_HALF 8192u     /* y += accum_half at the EVEN pixels: accum_half is [B, groups*Cout, ceil(H/2), ceil(W/2)], element
                                      * (h, w) is added to y(2h, 2w) -- the data gradient of a stride-2 1x1 conv (the
                                      * trunk's downsample.0, model.py:60-62 via torchvision Bottleneck) joins the block input's
                                      * gradient without ever being dilated in memory.  1x1 stride-1 GEMM form (configs 14, 15). */
#define SPK_EPI_TORGB 32768u         /* SPK_CONV_WINOGRAD launches with Cout <= 64 only: the 1x1 conv to 3 channels that follows the last
                                      * synthesis block (styleganv1.py:607: self.to_rgb) runs inside the epilogue -- rgb_y[b,o] = rgb_bias[o] +
                                      * sum_co rgb_w[o,co] * y[b,co] on the epilogue's own registers; y may then be NULL (the activation is
                                      * neither stored nor re-read).  No y_pre / ACCUM / modulation, ksplit 1. */
#define SPK_CONV_UP_FIR1331 512u     /* with UPSAMPLE2X: the x2 interpolation is upfirdn2d(up=2, FIR [1,3,3,1], pad (2,1)) --
                                      * the same (.75,.25) taps as bilinear, but neighbours outside the image are zero */

const char* spk_version(void);
const char* spk_last_error(void);

/* ---- 2-D convolution on the f32 MFMA pipe -----------------------------------------------------------
 * Kernel sizes 1x1, 3x3, 7x7; stride 1 or 2; zero padding (k-1)/2; fp32, exact (fmaf-chain) arithmetic.
 *   y[b,co,h,w] = epi( out_scale * sum_{ci,ky,kx} w[co,ci,ky,kx] * xin[b,ci,h*s+ky-p,w*s+kx-p] )
 *   epi(v) = style( lrelu( v + bias[co] + noise_w[co]*noise[b,0,h,w] ) ),
 *   style(v) = v*(style[b*style_stride + co] + 1) + style[b*style_stride + Cout + co]
 * Each stage of epi is enabled by its SPK_EPI_* flag.  xin is x, or (SPK_CONV_UPSAMPLE2X) the bilinear x2
 * (align_corners=False) upsampling of x formed on the fly, or (SPK_CONV_IN_AFFINE_RELU) max(x*a+b, 0)
 * with per-input-channel a, b -- the producer's BatchNorm + ReLU, never materialised.
 * replaces: styleganv1.py:624-628 (upsample, conv1, noise1, leaky_relu, style_mod1) and
 *           styleganv1.py:630-633 (conv2, noise2, leaky_relu, style_mod2) -- one launch each;
 *           stylegan.py:45-46 (WSConv2d: out_scale folds the x*scale pre-multiply);
 *           every nn.Conv2d of the torchvision ResNet-50 trunk built at model.py:60-62 (7x7 s2 stem,
 *           1x1 / 3x3 bottleneck convs, 1x1 s2 downsample) with the following BatchNorm2d statistics
 *           pass (SPK_EPI_STATS) and the preceding BatchNorm2d+ReLU application folded in.
 * Weights must first be packed for the chosen tile config with spk_conv2d_pack_weights. */
typedef struct spk_conv2d_desc {
    const float* x;          /* [B,Cin,Hin,Win] */
    const float* w_packed;   /* from spk_conv2d_pack_weights, same kernel size and `config` */
    const float* bias;       /* [Cout] or NULL */
    const float* noise_w;    /* [Cout] or NULL */
    const float* noise;      /* [B,1,H,W] or NULL */
    const float* style;      /* row b at style + b*style_stride: [s0(Cout) | s1(Cout)] or NULL */
    const float* in_scale;   /* [Cin] (SPK_CONV_IN_AFFINE_RELU) or NULL */
    const float* in_shift;   /* [Cin] */
    double*      stats;      /* [stats_slots][2*Cout] (SPK_EPI_STATS) or NULL; caller zeroes it */
    float*       y;          /* [B,Cout,H,W] */
    float*       y_pre;      /* [B,Cout,H,W] or NULL: the value before the style stage (after LeakyReLU), kept
                              * for the backward pass (sign = LeakyReLU mask, value = d style / d s0) */
    int32_t B, Cin, Cout;
    int32_t H, W;            /* OUTPUT spatial size */
    int32_t Hin, Win;        /* input spatial size (H = 2*Hin with SPK_CONV_UPSAMPLE2X, else (Hin+2p-k)/s+1) */
    int32_t kh, kw, stride;
    int32_t style_stride;    /* in floats */
    uint32_t flags;
    float lrelu_slope;
    float out_scale;         /* multiplies the contraction before bias (1.0 = none) */
    int32_t config;          /* tile config id, or -1 = auto */
    int32_t ksplit;          /* slices of the input-channel range (split-K); 0 = auto, 1 = none */
    void*   workspace;       /* device scratch for split-K partial sums (may be NULL if not needed) */
    int64_t workspace_bytes; /* its size; see spk_conv2d_workspace_bytes */
    /* StyleGAN2-style modulated convolution (build-defined variant): with
     * SPK_CONV_IN_BATCH_SCALE, in_scale is the modulation s[B,Cin]; out_scale_bc[B,Cout] (or NULL) multiplies the
     * contraction before bias/noise (the demodulation d[b,co] = rsqrt(sum (w*s)^2 + eps), spk_modconv_demod);
     * act_gain (0 = 1) multiplies the LeakyReLU output (sqrt 2 of FusedLeakyReLU).  out_scale_bc without
     * SPK_CONV_IN_BATCH_SCALE is rejected (the plain kernels are built without the demodulation read). */
    const float* out_scale_bc;
    float act_gain;
    /* Grouped convolution: `groups` (0 or 1 = ordinary) independent convs of the same shape in one launch -- the three
     * IRFD encoders Ei / Ee / Ep run the same ResNet-50 on the same image (model.py:84-90), so every layer of the three
     * is one launch here.  With groups > 1, Cin / Cout are PER GROUP; x has group_in_stride*(groups-1) + Cin channels,
     * group g reading [g*group_in_stride, +Cin) (group_in_stride = 0: all groups read the same input, the stem);
     * y, bias, stats have groups*Cout channels; w_packed is the groups' packed images one after another.
     * Allowed flags: SPK_EPI_BIAS | LRELU | ACCUM | STATS, SPK_CONV_IN_AFFINE_RELU (in_scale / in_shift per x channel). */
    int32_t groups;
    int32_t group_in_stride;
    /* SPK_EPI_STATS: `stats` holds stats_slots (0 = 1) copies of the [2*groups*Cout] sums; the workgroups of pixel
     * tile i add into copy i % stats_slots and spk_bn_finalize adds the copies up.  fp64 atomics are what this
     * epilogue costs on MI355X: same-address ones serialise at ~0.3 us each (the 512 pixel tiles of the trunk's 128^2
     * stem spent 500 us queueing on one copy) and ~10-20 per ns is all the chip retires.  With stats_slots >=
     * spk_conv2d_stats_slots(...) every pixel tile owns its copy and the sums are plain stores (no atomics at all):
     * the setting for high-resolution layers.  The caller zeroes all copies either way. */
    int32_t stats_slots;
    const float* accum_half; /* SPK_EPI_ACCUM_HALF: [B, groups*Cout, ceil(H/2), ceil(W/2)], added at the even pixels; else NULL */
    const float* out_scale_dev; /* optional DEVICE scalar multiplied into out_scale (NULL = 1): 1 / sigma of a spectrally normalised
                                 * weight (styleganv1.py:644-672 wraps every discriminator layer), so that weight_orig is packed once
                                 * per optimizer step instead of W / sigma once per forward.  Tap kernels and SPK_CONV_DGRAD_S2. */
    /* SPK_EPI_TORGB: rgb_w [rgb_channels = 3][Cout], rgb_bias [3] or NULL, rgb_y [B,3,H,W] (8-byte aligned) */
    const float* rgb_w;
    const float* rgb_bias;
    float* rgb_y;
    int32_t rgb_channels;
    int32_t reserved;
} spk_conv2d_desc;

int spk_conv2d_num_configs(void);
/* pixel tiles of the launch (its gridDim.x): the stats_slots value from which every tile owns its copy; -1 = unsupported */
int spk_conv2d_stats_slots(int config, int kh, int kw, int stride, int B, int Cin, int Cout, int H, int W);
/* whether tile config `config` is built for this kernel size / stride */
int spk_conv2d_config_valid(int config, int kh, int kw, int stride);
/* tile config chosen by the heuristic for this problem (what `config = -1` resolves to); H, W = output size */
int spk_conv2d_pick_config(int kh, int kw, int stride, int B, int Cin, int Cout, int H, int W);
/* tile config for SPK_CONV_DGRAD_S2 (the data gradient of a 3x3 stride-2 conv with Cin -> Cout... seen from the gradient:
 * Cin = channels of the output-side gradient, Cout = channels of the input-side one, Hin x Win = the gradient's size):
 * 13 = the exact-tap kernel, else one of 0-3 */
int spk_conv2d_dgrad_s2_config(int B, int Cin, int Cout, int Hin, int Win);
/* SPK_CONV_DGRAD_S2 with the exact-tap kernel (config 13) and few workgroups (small gradient planes): with desc->ksplit != 1 and a
 * workspace of this many bytes (0: none needed) the contraction runs in slices and the split-K finisher adds them up.  Arguments as
 * the DGRAD_S2 descriptor's: Cin / Hin / Win describe the output-side gradient, Cout / H / W the input-side one. */
int64_t spk_conv2d_dgrad_s2_workspace_bytes(int B, int Cin, int Cout, int Hin, int Win, int H, int W, int groups);
/* CO_T / CI_T / PIX_T of a config (any out pointer may be NULL) */
int spk_conv2d_config_info(int config, int* co_tile, int* ci_tile, int* pix_tile);
/* number of floats of the packed image of a [Cout,Cin,kh,kw] weight for `config` (<0 = bad args) */
int64_t spk_conv2d_packed_floats(int config, int kh, int kw, int Cin, int Cout);
/* bytes of scratch spk_conv2d_fwd needs for this problem (0 when it will not split K; <0 = the config cannot
 * host the shape).  config = -1 / ksplit = 0 ask for the library's own choices. */
int64_t spk_conv2d_workspace_bytes(int config, int ksplit, int kh, int kw, int stride, int B, int Cin, int Cout,
                                   int H, int W);
/* the same for a grouped launch (Cin, Cout per group) */
int64_t spk_conv2d_workspace_bytes_grouped(int config, int ksplit, int kh, int kw, int stride, int B, int Cin, int Cout,
                                           int H, int W, int groups);
/* w[Cout,Cin,kh,kw] -> packed [co_tile][ci_chunk][tap][ci][co] (zero padded).
 * transpose_flip = 1 packs the data-gradient operator instead: w'[ci,co,ky,kx] = w[co,ci,kh-1-ky,kw-1-kx]
 * (then the packed image is that of a [Cin,Cout,kh,kw] weight).
 * transpose_flip = 2 (kh = kw = 3): the four output-parity 2x2 kernels of the STRIDE-2 data gradient
 * (SPK_CONV_DGRAD_S2), the image of a [4*Cin, Cout, 2, 2] weight: spk_conv2d_packed_floats(config, 2, 2, Cout, 4*Cin)
 * floats, config in 0-3; config 13 (the exact-tap kernel) packs the transposed 3x3 operator itself,
 * [co tile 64][chunk 8][tap][8][64].  2x2 is accepted by the size / config queries for that purpose only.
 * transpose_flip = 3 (kh = kw = 4, w is the [Cin,Cout,4,4] weight of a ConvTranspose2d): the four output-parity 2x2 kernels
 * of SPK_CONV_TRANSPOSE4X4_S2: spk_conv2d_packed_floats(config, 2, 2, Cin, 4*Cout) floats, config in 0-3.
 * replaces: nothing in the reference (layout change private to this library). */
int spk_conv2d_pack_weights(const float* w, float* w_packed, int kh, int kw, int Cin, int Cout, int config,
                            int transpose_flip, void* stream);
/* The same for `n` (1..SPK_PACK_LIST_MAX) weight tensors of one shape on ONE launch: their packed images one after another
 * in w_packed (n x spk_conv2d_packed_floats floats) -- what a grouped launch (spk_conv2d_desc.groups) reads.  `ws` is a HOST
 * array of device pointers.  replaces: the per-encoder weight reads of the three torchvision trunks Ei / Ee / Ep
 * (model.py:60-62), whose convs of one position run as one grouped launch. */
#define SPK_PACK_LIST_MAX 8
int spk_conv2d_pack_weights_list(const float* const* ws, int n, float* w_packed, int kh, int kw, int Cin, int Cout, int config,
                                 int transpose_flip, void* stream);
int spk_conv2d_fwd(const spk_conv2d_desc* desc, void* stream);
/* The SPK_CONV_BF16X3 path: packed image size in BYTES / packer (w is the fp32 [Cout,Cin,3,3] parameter; the hi / lo split
 * happens here, once per weight update) / whether a shape is served / the launch itself (spk_conv2d_fwd forwards to it). */
int64_t spk_conv2d_packed_bytes_bf16x3(int Cin, int Cout);
int spk_conv2d_pack_weights_bf16x3(const float* w, void* w_packed, int Cin, int Cout, void* stream);
/* transpose_flip = 1: the data-gradient operator of the conv (w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]): its image has
 * spk_conv2d_packed_bytes_bf16x3(Cout, Cin) bytes and is run with Cin / Cout exchanged -- the opt-in reduced-precision TRAINING
 * path (forward and data gradients on the bf16 pipe, weight gradients exact) */
int spk_conv2d_pack_weights_bf16x3_tf(const float* w, void* w_packed, int Cin, int Cout, int transpose_flip, void* stream);
int spk_conv2d_bf16x3_supported(int B, int Cin, int Cout, int H, int W);
int spk_conv2d_bf16x3_fwd(const spk_conv2d_desc* desc, void* stream);

/* The SPK_CONV_WINOGRAD path: size in BYTES of the transformed weight image U = G g G^T / the packer (w is the fp32
 * [Cout,Cin,3,3] parameter; transpose_flip = 1: the data-gradient operator w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx], whose
 * image has spk_conv2d_packed_bytes_wino(Cout, Cin) bytes and is run with Cin / Cout exchanged) / whether a shape is served /
 * the launch itself (spk_conv2d_fwd forwards to it).  replaces: the F.conv2d of styleganv1.py:625,630 (SynthesisBlock conv1 / conv2
 * after the separate x2 upsampling), styleganv1.py:662 (DiscriminatorBlock conv1) and their data gradients. */
int64_t spk_conv2d_packed_bytes_wino(int Cin, int Cout);
int spk_conv2d_pack_weights_wino(const float* w, float* w_packed, int Cin, int Cout, int transpose_flip, void* stream);
/* n <= SPK_WINO_PACK_MAX weights in one launch (host arrays of length n; the same bits as n single calls): what a training step does
 * after every optimizer step for both images of each decoder layer */
#define SPK_WINO_PACK_MAX 32
int spk_conv2d_pack_weights_wino_list(const float* const* w, float* const* w_packed, const int* Cin, const int* Cout, const int* transpose_flip,
                                      int n, void* stream);
int spk_conv2d_wino_supported(int B, int Cin, int Cout, int H, int W);
/* Regions are 32 x 8 output pixels, or 16 x 16 where the image is narrower than 32.  A problem with too few (region, channel tile)
 * pairs to fill the CUs runs its channel contraction in `ksplit` slices: spk_conv2d_wino_ksplit(want, ...) = the count the launch
 * will use for desc->ksplit = want (0 = automatic; 1 = no split), ..._workspace_bytes the partial-sum workspace it then needs in
 * desc->workspace ([ksplit][B][Cout][H][W]; 0 when not split); the epilogue then runs in the direct kernels' split-K finisher. */
int spk_conv2d_wino_ksplit(int want, int B, int Cin, int Cout, int H, int W);
int64_t spk_conv2d_wino_workspace_bytes(int ksplit, int B, int Cin, int Cout, int H, int W);
int spk_conv2d_wino_fwd(const spk_conv2d_desc* desc, void* stream);

/* ---- backward of the convolution ---------------------------------------------------------------------
 * Data gradient: spk_conv2d_fwd itself on the output gradient with weights packed transpose_flip = 1
 * (stride 1).  Weight gradient (any supported kernel / stride), on the f32 MFMA pipe with the pixel axis
 * as the contraction:
 *   dw[co,ci,ky,kx] (+)= scale * sum_{b,h,w} g[b,co,h,w] * xin[b,ci,h*s+ky-p,w*s+kx-p]
 * xin is x, or max(x*a+b, 0) per input channel (SPK_CONV_IN_AFFINE_RELU), as the forward pass formed it; for a
 * forward that used SPK_CONV_UPSAMPLE2X pass the x2 image itself (spk_upsample2x_bilinear_fwd).
 * Partial sums go to `workspace` and are reduced in a fixed order: bitwise reproducible, no float atomics.
 * replaces: the aten::convolution_backward weight path under loss.backward() (train.py:205) for
 *           styleganv1.py:625,630 and the trunk convs. */
typedef struct spk_wgrad_desc {
    const float* g;          /* [B,Cout,H,W] gradient w.r.t. the conv output */
    const float* x;          /* [B,Cin,Hin,Win] forward input */
    const float* in_scale;   /* [Cin] / NULL, as in the forward call */
    const float* in_shift;
    float*       dw;         /* [Cout,Cin,kh,kw] */
    int32_t B, Cin, Cout, H, W, Hin, Win, kh, kw, stride;
    uint32_t flags;          /* 0, SPK_CONV_IN_AFFINE_RELU, SPK_CONV_UPSAMPLE2X, or SPK_CONV_IN_BATCH_SCALE
                              * [| SPK_CONV_UPSAMPLE2X | SPK_CONV_UP_FIR1331] (the modulated convolution, see g_scale) */
    float scale;
    int32_t accumulate;      /* dw += instead of dw = */
    int32_t splits;          /* pixel-range splits; 0 = auto */
    void*   workspace;
    int64_t workspace_bytes; /* >= spk_conv2d_wgrad_workspace_bytes(...) (with Cout = groups * Cout when grouped) */
    /* grouped form, as in spk_conv2d_desc: Cin / Cout per group, g has groups*Cout channels, x has
     * group_in_stride*(groups-1) + Cin, dw is [groups*Cout, Cin, kh, kw] (the groups' gradients one after another);
     * Cout must be a multiple of 64.  0 / 1 = ordinary. */
    int32_t groups;
    int32_t group_in_stride;
    /* fold (0 / 1 = none; must divide groups): groups q and q + groups/fold are the SAME conv applied to another set of
     * images (IRFD runs each encoder on x_s and on x_t, model.py:84-90), so their weight gradients add: dw is
     * [groups/fold * Cout, Cin, kh, kw], summed in the slab reduce instead of by a separate pass. */
    int32_t fold;
    /* SPK_CONV_IN_BATCH_SCALE (the StyleGAN2 variant's modulated conv, reference/styleganv2.txt:1835): the forward input was
     * x * in_scale[b,ci] (in_scale = the modulation s, [B][Cin]) and the gradient that reaches the conv output is
     * g * g_scale[b,co] (g_scale = demodulation x activation gain, [B][Cout]): both factors are applied while the tiles are
     * staged into LDS, neither rescaled tensor exists.  With SPK_CONV_UPSAMPLE2X | SPK_CONV_UP_FIR1331 x is the low-resolution
     * tensor and the x2 image (upfirdn2d up = 2, [1,3,3,1]: zero border) is interpolated LDS -> LDS.  3x3 stride 1, ungrouped;
     * shapes: spk_conv2d_wgrad_mod_supported. */
    const float* g_scale;
} spk_wgrad_desc;
int64_t spk_conv2d_wgrad_workspace_bytes(int kh, int kw, int stride, int splits, int B, int Cin, int Cout, int H, int W);
int spk_conv2d_wgrad(const spk_wgrad_desc* desc, void* stream);
/* whether spk_conv2d_wgrad takes SPK_CONV_UPSAMPLE2X for a 3x3 stride-1 problem with OUTPUT size H x W (x is then the
 * low-resolution [B,Cin,H/2,W/2] tensor and the x2 image is never materialised); 0: upsample first. */
int spk_conv2d_wgrad_up_supported(int B, int Cin, int Cout, int H, int W);
int spk_conv2d_wgrad_mod_supported(int B, int Cin, int Cout, int H, int W, int upsample);
/* The same gradient as Winograd F(2x2, 3x3) (wgrad3x3_wino_f32.hip): flags = SPK_CONV_WINOGRAD on a 3x3 stride-1 pad-1 problem
 * (x is the conv's actual input: a x2 layer passes the materialised x2 image, spk_upsample2x_fwd); fp32 throughout, 16/36 of the
 * direct form's multiply-adds.  With SPK_CONV_IN_BATCH_SCALE (ungrouped: in_scale = s[B,Cin], g_scale = d'[B,Cout]) the modulated
 * convolution; with SPK_CONV_IN_AFFINE_RELU (in_scale / in_shift per input channel of x) the input was relu(x * scale + shift);
 * groups / group_in_stride / fold as in spk_wgrad_desc (the shape queries then take Cout = groups * Cout).
 * Shapes: Cin, Cout multiples of 64, H even, W a multiple of 16 (..._supported); the workspace holds
 * `splits` slabs [Cout][9][Cin] (..._workspace_bytes; ..._splits returns the split count the kernel will use for `splits` = the
 * wanted count, 0 = auto), reduced in a fixed order by spk_wgrad_reduce_slabs -- bitwise reproducible. */
int spk_conv2d_wgrad_wino_supported(int B, int Cin, int Cout, int H, int W);
int spk_conv2d_wgrad_wino_splits(int splits, int B, int Cin, int Cout, int H, int W);
int64_t spk_conv2d_wgrad_wino_workspace_bytes(int splits, int B, int Cin, int Cout, int H, int W);
int spk_conv2d_wgrad_wino(const spk_wgrad_desc* desc, void* stream);
/* dw[co][ci][tap] (=, or += when accumulate) scale * sum over n_slabs slabs [Cout][taps][Cin], in slab order; fold > 1: rows
 * co and co + Cout/fold add (see spk_wgrad_desc.fold) */
int spk_wgrad_reduce_slabs(const float* slabs, float* dw, int n_slabs, int Cout, int Cin, int taps, float scale, int accumulate, int fold,
                           void* stream);

/* Adjoint of the fused epilogue of spk_conv2d_fwd, one pass.  With y = a*(s0+1)+s1, a = lrelu(t),
 * t = conv + bias + noise_w*noise and dy = dL/dy:
 *   dt[b,c,p] = dy * (s0[b,c]+1) * (a > 0 ? 1 : slope)          (feeds the data / weight gradient)
 *   sums[b,:,c] = { sum_p dy*a, sum_p dy, sum_p dt, sum_p dt*noise[b,p] }          (layout [B][4][C])
 * so d s0 = sums[b,0,:], d s1 = sums[b,1,:] -- rows 0-1 of an image are its style gradient [d s0 | d s1] as ApplyStyle's
 * linear layer wants it, no copy -- and d bias = sum_b sums[b,2,:], d noise_w = sum_b sums[b,3,:].
 * a / noise / style may be NULL (stage absent).  dt may alias dy.
 * replaces: autograd's backward of styleganv1.py:626-628 / :631-633 (noise, leaky_relu, style_mod). */
int spk_epilogue_bwd(const float* dy, const float* a, const float* noise, const float* style, int64_t style_stride,
                     float slope, float* dt, float* sums, int B, int C, int64_t HW, void* stream);
/* adjoint of spk_upsample2x_bilinear_fwd: dy [planes,2Hin,2Win] -> dx [planes,Hin,Win] */
int spk_upsample2x_bilinear_bwd(const float* dy, float* dx, int64_t planes, int Hin, int Win, void* stream);
/* toRGB backward: dx (may be NULL) and per-workgroup partial sums partial[blocks][O*C + O]
 * (d w then d bias; sum over the first axis, blocks = spk_conv1x1_small_bwd_blocks(B, HW)). */
int spk_conv1x1_small_bwd_blocks(int B, int64_t HW);
int spk_conv1x1_small_bwd(const float* x, const float* w, const float* dy, float* dx, float* partial, int B, int C, int O,
                          int64_t HW, float in_scale, void* stream);
/* FC backward: dz = dout * (out > 0 ? 1 : slope); dx = wmul * dz @ w (NULL to skip);
 * dw = wmul * dz^T @ x, db = bmul * sum_b dz (dw NULL to skip both).
 * replaces: autograd's backward of FC.forward (styleganv1.py:489-495). */
int spk_fc_bwd(const float* dout, const float* out, const float* x, int64_t x_stride, const float* w, float* dx,
               int64_t dx_stride, float* dw, float* db, int B, int I, int O, float wmul, float bmul, float slope,
               void* stream);

/* ---- fully connected + LeakyReLU ---------------------------------------------------------------
 * out[b,o] = act( wmul * sum_i x[b*x_stride + i] * w[o*I + i] + bmul * bias[o] ),
 * act(v) = v > 0 ? v : slope*v  (slope = 1 => identity).
 * replaces: styleganv1.py:489-495 (FC.forward: F.linear with runtime w_lrmul/b_lrmul + leaky_relu),
 *           used by the mapping stack :513-518,:532 and by every ApplyStyle :461,:464;
 *           stylegan.py:20-21 (WSLinear); model.py:121-122 (Cm = nn.Linear(2048, 8)). */
int spk_fc_fwd(const float* x, int64_t x_stride, const float* w, const float* bias, float* out,
               int64_t out_stride, int B, int I, int O, float wmul, float bmul, float slope, void* stream);

/* Several independent FCs in one launch -- the 13 ApplyStyle affines of a decoder step (styleganv1.py:466, called
 * from :599,:628,:633) all depend only on the dlatents, so they need not be 13 launches.  `groups` is a HOST array of
 * n_groups <= SPK_FC_MAX_GROUPS descriptors (copied into the kernel arguments); every group computes
 * out[b,o] = lrelu_slope(wmul * <x[b], w[o]> + bmul * bias[o]) for b < B.  Rows must be 16-byte aligned, I % 4 == 0. */
#define SPK_FC_MAX_GROUPS 16
typedef struct spk_fc_group {
    const float* x;      /* [B, I], row stride x_stride */
    int64_t x_stride;
    const float* w;      /* [O, I] */
    const float* bias;   /* [O] or NULL */
    float* out;          /* [B, O], row stride out_stride */
    int64_t out_stride;
    int32_t I, O;
    float wmul, bmul, slope;
    int32_t reserved;
} spk_fc_group;
int spk_fc_grouped_fwd(const spk_fc_group* groups, int n_groups, int B, void* stream);
/* The backward of up to SPK_FC_MAX_GROUPS independent FCs in two launches (spk_fc_bwd's two kernels over the groups): per group
 * dz = dout * act'(out); dx[b, :] (row stride dx_stride; NULL to skip) = wmul * dz @ w; dw = wmul * dz^T @ x, db = bmul * sum_b dz
 * (dw NULL to skip both).  replaces: autograd's backward of the 13 ApplyStyle.linear FCs of a synthesis pass
 * (styleganv1.py:463-468), whose input gradients are the rows of one [B, 13, 512] latent gradient. */
typedef struct spk_fc_bwd_group {
    const float* dout;   /* [B, O], row stride dout_stride (>= O: e.g. rows 0-1 of spk_epilogue_bwd's sums, O = 2C, stride 4C) */
    int64_t dout_stride;
    const float* out;    /* [B, O] the saved forward output */
    const float* x;      /* [B, I], row stride x_stride (for dw) */
    int64_t x_stride;
    const float* w;      /* [O, I] (for dx) */
    float* dx;           /* [B, I], row stride dx_stride, or NULL */
    int64_t dx_stride;
    float* dw;           /* [O, I] or NULL */
    float* db;           /* [O] or NULL */
    int32_t I, O;
    float wmul, bmul, slope;
    int32_t reserved;
} spk_fc_bwd_group;
int spk_fc_grouped_bwd(const spk_fc_bwd_group* groups, int n_groups, int B, void* stream);

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
/* 1x1 conv FROM C <= 4 channels with bias and LeakyReLU (slope 1 = none) fused: y[b,o] = lrelu(bias[o] + scale * sum_c w[o,c] x[b,c]); w is
 * the plain [O][C] matrix, scale_dev an optional DEVICE scalar on it (1 / sigma of a spectrally normalised layer).  A store stream:
 * HW % 4 == 0, 16-byte aligned tensors.  replaces: StyleDiscriminator.fromrgb + leaky_relu (styleganv1.py:675,684), 3 -> 64 at 256^2. */
int spk_conv1x1_expand_fwd(const float* x, const float* w, const float* bias, const float* scale_dev, float* y, int B, int C, int O,
                           int64_t HW, float slope, void* stream);
int spk_conv1x1_small_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int O,
                          int64_t HW, float in_scale, void* stream);

/* ---- bilinear x2 upsampling (align_corners = False) -----------------------------------------------
 * replaces: styleganv1.py:621,624 (nn.Upsample) / stylegan.py:168 (F.interpolate) when used un-fused. */
int spk_upsample2x_bilinear_fwd(const float* x, float* y, int64_t planes, int Hin, int Win, void* stream);
/* the same x2 image with a choice of border: zero_border = 0 bilinear (edge taps clamped, as above); 1 = upfirdn2d(up = 2, FIR
 * [1,3,3,1], pad (2,1)) -- the same (.25, .75) taps with neighbours outside the image counted as zero: the x2 of the StyleGAN2
 * variant's styled convs (reference/styleganv2.txt:1835).  The materialised input of a Winograd x2 layer. */
int spk_upsample2x_fwd(const float* x, float* y, int64_t planes, int Hin, int Win, int zero_border, void* stream);

/* ---- BatchNorm2d pieces (torchvision ResNet-50 trunk, model.py:60-62) ------------------------------
 * spk_bn_finalize: turns the fp64 batch sums a conv epilogue accumulated (SPK_EPI_STATS) into the affine
 *   the consumer applies: mean = S/N, var = SS/N - mean^2 (biased), scale = gamma*rsqrt(var+eps),
 *   shift = beta - mean*scale; and (momentum > 0) updates running_mean / running_var (unbiased var) as
 *   nn.BatchNorm2d does in training mode.  With stats = NULL (eval mode) the running statistics are used.
 *   save_mean / save_invstd ([C], may be NULL) keep the batch statistics for the backward pass.
 *   stats = [stats_slots][2C] (spk_conv2d_desc.stats_slots; 0 = 1): the copies are added here, and with more than one
 *   copy the totals are written back to copy 0 (so a second finalize of the same sums -- the replayed running-statistics
 *   update of a pass that ran twice -- passes stats_slots = 1).
 * replaces: the statistics half of F.batch_norm for every bn1/bn2/bn3/downsample.1 of the trunk. */
int spk_bn_finalize(double* stats, int stats_slots, int64_t count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, float* scale, float* shift, float* save_mean,
                    float* save_invstd, int C, void* stream);
/* The replayed running-statistics update of a pass (the visible side effect of the reference's re-entrant checkpoint,
 * model.py:84-90: every BatchNorm's momentum update happens a second time) for MANY BatchNorms on one launch: item i applies
 * exactly spk_bn_finalize's update -- running = (1 - momentum) running + momentum {mean, unbiased var} -- from the TOTALS
 * stats[i] = [2*C[i]] (copy 0 of a finalized sums buffer).  Host array of up to SPK_BN_LIST_MAX items per call. */
#define SPK_BN_LIST_MAX 64
typedef struct spk_bn_replay_item {
    const double* stats;     /* [2*C] totals: sum, sum of squares */
    float* running_mean;     /* [C] */
    float* running_var;      /* [C] */
    int64_t count;           /* elements per channel the sums run over */
    int32_t C;
    int32_t reserved;
} spk_bn_replay_item;
int spk_bn_replay_list(const spk_bn_replay_item* items_host, int n, float momentum, void* stream);
/* y = [relu]( a*sa[c] + ba[c] + (b ? b*sb[c] + bb[c] : 0) ): BatchNorm apply (+ residual add) (+ ReLU).
 * sb/bb NULL = identity on b.  replaces: bn3 + `out += identity` + relu at the end of every torchvision
 * Bottleneck.forward, bn1+relu of the stem (with b = NULL). */
int spk_bn_add_relu_fwd(const float* a, const float* sa, const float* ba, const float* b, const float* sb,
                        const float* bb, float* y, int B, int C, int64_t HW, int relu, void* stream);
/* 3x3 stride-2 pad-1 max pooling; with in_scale/in_shift != NULL the input is max(x*s[c]+b[c],0) formed on the
 * fly (stem bn1+relu folded).  replaces: resnet50.maxpool (children()[3], model.py:62). */
int spk_maxpool3x3s2_fwd(const float* x, const float* in_scale, const float* in_shift, float* y, int B, int C,
                         int Hin, int Win, void* stream);
/* BatchNorm2d backward (training mode), in the same split form as the forward.  z = r*scale[c]+shift[c] is the BN
 * output of the raw conv output r; g is the gradient w.r.t. relu(z) (mask_mode 1: mask recomputed from r), w.r.t.
 * relu(z + identity) (mask_mode 2: mask = mask_src > 0, mask_src = the block output) or w.r.t. z (mask_mode 0).
 * g_per_plane != 0: g holds one value per (b,c) plane (the global-average-pool gradient), scaled by g_scale.
 *   spk_bn_bwd_reduce: sums[b,:,c] = { sum dz, sum dz*rhat }, layout [B][2][C]: csum = sum_b sums is [2][C], its rows
 *                      d beta and d gamma as they are (no strided copies)
 *   spk_bn_bwd_apply : dr = gamma*invstd*(dz - csum[0,c]/count - rhat*csum[1,c]/count); dz_out (may be NULL) = dz
 * replaces: autograd's native_batch_norm_backward + threshold_backward for every BatchNorm2d/ReLU of the trunk. */
int spk_bn_bwd_reduce(const float* g, const float* r, const float* mask_src, int mask_mode, const float* scale,
                      const float* shift, const float* mean, const float* invstd, float g_scale, int g_per_plane,
                      float* sums, int B, int C, int64_t HW, void* stream);
int spk_bn_bwd_apply(const float* g, const float* r, const float* mask_src, int mask_mode, const float* scale,
                     const float* shift, const float* mean, const float* invstd, const float* csum, int64_t count,
                     float g_scale, int g_per_plane, float* dr, float* dz_out, int B, int C, int64_t HW, void* stream);
/* spk_bn_bwd_apply taking the reduce pass's per-image sums [B][2][C] as they are: every workgroup adds up its channel's B pairs
 * itself (no batch-reduction launch in between) and the totals -- rows d beta, d gamma -- are written to csum_out [2][C].
 * batch_stats = 0: an eval-mode BatchNorm (a fixed affine: no mean / variance terms in dr; csum_out is still the totals). */
int spk_bn_bwd_apply_sums(const float* g, const float* r, const float* mask_src, int mask_mode, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const float* sums, float* csum_out,
                          int batch_stats, int64_t count, float g_scale, int g_per_plane, float* dr, float* dz_out, int B, int C,
                          int64_t HW, void* stream);
/* zero-insertion x2 ([planes,H,W] -> [planes,Ho,Wo], Ho in {2H-1,2H}): the data gradient of a stride-2 conv is the
 * stride-1 transpose_flip conv of the dilated output gradient. */
int spk_dilate2x(const float* x, float* y, int64_t planes, int H, int W, int Ho, int Wo, void* stream);
/* adjoint of spk_maxpool3x3s2_fwd (same optional folded affine+relu on x; torch's first-maximum tie rule) */
int spk_maxpool3x3s2_bwd(const float* x, const float* in_scale, const float* in_shift, const float* dy, float* dx, int B,
                         int C, int Hin, int Win, void* stream);
/* y[b,c] = mean_hw x[b,c,:,:].  replaces: resnet50.avgpool = AdaptiveAvgPool2d(1) (children()[8]). */
int spk_global_avgpool_fwd(const float* x, float* y, int64_t planes, int64_t HW, void* stream);

/* ---- StyleGAN2 pieces (build-defined variant; the reference only describes them in prose,
 *      reference/styleganv2.txt:1835,1912 -- parity unpinned by the reference) --------------------------------
 * spk_modconv_demod: d[b,co] = rsqrt( scale^2 * sum_{ci,k} (w[co,ci,k] * s[b,ci])^2 + eps )
 * spk_upfirdn2d_fwd: zero-insert upsample by `up`, pad (pad0 before / pad1 after, negative = crop), correlate
 *   with the FLIPPED k x k filter (given on the host), keep every `down`-th sample -- upfirdn2d of the StyleGAN2
 *   reference implementation; out size = (H*up + pad0 + pad1 - k)/down + 1.
 * spk_conv1x1_small_mod_fwd: y[b,o,p] = sum_c w[o,c]*mod[b,c]*x[b,c,p]*in_scale + bias[o] (toRGB: modulated, no
 *   demodulation), O <= 4. */
int spk_modconv_demod(const float* w, const float* s, float* d, int B, int Cin, int Cout, int taps, float scale, float eps,
                      void* stream);
/* ---- backward of the modulated convolution (the parts that are not spk_conv2d_fwd / spk_conv2d_wgrad themselves) ----
 * With y = gain * lrelu(d[b,co] * scale * conv3x3(up?(x) * s[b,ci], w) + ...), dz the gradient at the conv output
 * (spk_epilogue_bwd) and dx~ = spk_conv2d_fwd(dz, w transpose-flipped, SPK_CONV_IN_BATCH_SCALE with in_scale = d * gain):
 *   spk_modconv_dx_finish:  dx = s[b,ci] * up^T(dx~)   (up^T = the adjoint of upfirdn2d(up = 2, [1,3,3,1]) when `upsample`,
 *                           identity otherwise; dx may be NULL, or == dxt when !upsample) and
 *                           ds[b,ci] = <up^T(dx~), x>_plane -- the modulation gradient through the conv, evaluated at the
 *                           LOW resolution, so up(x) is never formed.  dxt [B,C,(2)Hs,(2)Ws], x / dx [B,C,Hs,Ws], s / ds [B,C].
 *   spk_modconv_demod_bwd:  the adjoint of spk_modconv_demod: with e = -dd * d^3 * scale^2,
 *                           ds[b,ci] += s[b,ci] * sum_co e[b,co] * sum_k w[co,ci,k]^2,
 *                           dw[co,ci,k] += w[co,ci,k] * sum_b e[b,co] * s[b,ci]^2      (either output may be NULL).
 *   spk_torgb_mod_bwd_data: dx[b,c,p] = in_scale * mod[b,c] * sum_o w[o,c] * dy[b,o,p]  (modulated toRGB, O <= 4); its weight /
 *                           modulation gradients come from spk_conv1x1_small_bwd's per-image partial sums on the UNmodulated x.
 * replaces: what autograd would run for the published formulas (reference/styleganv2.txt:1835,1912). */
/* spk_modconv_epi_finish: from spk_epilogue_bwd's plane sums [B][4][C] (a = y, no style) of a modulated conv, in one launch:
 *   dprime = d * gain (d NULL: 1), dd = (sum dy*y - gain*noise_w*sum dt*noise - gain*bias*sum dt) / d  (NULL: skipped),
 *   dbias = gain * sum_b sum dt, dnw = gain * sum_b sum dt*noise  (NULL: skipped). */
int spk_modconv_epi_finish(const float* sums, const float* d, const float* bias, const float* noise_w, float gain, float* dd,
                           float* dprime, float* dbias, float* dnw, int B, int C, void* stream);
int spk_modconv_dx_finish(const float* dxt, const float* x, const float* s, float* dx, float* ds, int B, int C, int Hs, int Ws,
                          int upsample, void* stream);
int64_t spk_modconv_demod_bwd_workspace_bytes(int B, int Cin, int Cout);      /* scratch for the ds half */
int spk_modconv_demod_bwd(const float* w, const float* s, const float* d, const float* dd, float* ds, float* dw, void* workspace,
                          int64_t workspace_bytes, int B, int Cin, int Cout, int taps, float scale, void* stream);
int spk_torgb_mod_bwd_data(const float* w, const float* mod, const float* dy, float* dx, int B, int C, int O, int64_t HW,
                           float in_scale, void* stream);
/* the same for several layers in one launch (a decoder step's 13 demodulation vectors depend only on its modulations);
 * `groups` is a HOST array of n_groups <= SPK_DEMOD_MAX_GROUPS descriptors. */
#define SPK_DEMOD_MAX_GROUPS 16
typedef struct spk_demod_group {
    const float* w;   /* [Cout, Cin, taps] */
    const float* s;   /* [B, Cin] */
    float* d;         /* [B, Cout] */
    int32_t Cin, Cout, taps;
    float scale;
} spk_demod_group;
int spk_modconv_demod_grouped(const spk_demod_group* groups, int n_groups, int B, float eps, void* stream);
int spk_upfirdn2d_fwd(const float* x, float* y, const float* filter_host, int k, int64_t planes, int H, int W, int up, int down,
                      int pad0, int pad1, float gain, void* stream);
int spk_conv1x1_small_mod_fwd(const float* x, const float* w, const float* mod, const float* bias, float* y, int B, int C, int O,
                              int64_t HW, float in_scale, void* stream);
/* The whole skip-generator toRGB step of StyleGAN2 in one launch: y = modulated 1x1 conv (as above) + bias +
 * upfirdn2d(skip, up = 2, FIR [1,3,3,1], pad (2,1)); x [B,C,H,W], skip [B,O,H/2,W/2] or NULL, y [B,O,H,W].  The 3-channel
 * upsample + add ride in the epilogue of the kernel that streams the C-channel activations (two HBM passes fewer). */
int spk_torgb_mod_skip_fwd(const float* x, const float* w, const float* mod, const float* bias, const float* skip, float* y, int B,
                           int C, int O, int H, int W, float in_scale, void* stream);

/* ---- stand-alone StyleGAN1 / ProGAN ops (the reference's only definitions of the PixelNorm / FIR-blur family) ----
 * spk_pixelnorm_fwd: y = x * rsqrt(mean_c x^2 + eps) over dim 1 of [B,C,HW] (HW = 1 for latents).
 *   replaces: styleganv1.py:132-136 (PixelNorm, sqrt_form = 0); stylegan.py:28-29 (x / sqrt(...), sqrt_form = 1).
 * spk_instance_norm_affine_fwd: per (b,c) plane y = (x-mean)*rsqrt(var_biased+eps) * scale[b*sb_stride+c] + bias[...]
 *   (scale / bias NULL = 1 / 0).  replaces: styleganv1.py:148-152 (InstanceNorm, eps 1e-8);
 *   stylegan.py:91-95 (AdaIN = nn.InstanceNorm2d (eps 1e-5) then style_scale * x + style_bias).
 * spk_blur2d_fwd: depthwise k x k FIR (filter given on the HOST, k <= 7), zero pad (k-1)/2, stride 1 or 2.
 *   replaces: styleganv1.py:52-63 (Blur2d.forward: F.conv2d with groups = C).
 * spk_upscale2d_nearest_fwd: y[h,w] = gain * x[h/f, w/f].  replaces: styleganv1.py:113-120 (Upscale2d).
 * spk_fade_in_tanh_fwd: y = tanh(alpha*a + (1-alpha)*b).  replaces: stylegan.py:155-157 (Generator.fade_in). */
int spk_pixelnorm_fwd(const float* x, float* y, int B, int C, int64_t HW, float eps, int sqrt_form, void* stream);
int spk_instance_norm_affine_fwd(const float* x, float* y, const float* scale, const float* bias, int64_t sb_stride, int B, int C,
                                 int64_t HW, float eps, void* stream);
/* its adjoint (autograd of stylegan.py:84-95): dx [B,C,HW] (or NULL), dscale / dbias [B,C] contiguous (or NULL);
 * `scale` as in the forward (NULL = 1). */
int spk_instance_norm_affine_bwd(const float* x, const float* dy, const float* scale, int64_t sb_stride, float* dx, float* dscale,
                                 float* dbias, int B, int C, int64_t HW, float eps, void* stream);
int spk_blur2d_fwd(const float* x, float* y, const float* filter_host, int k, int64_t planes, int H, int W, int stride, void* stream);
int spk_upscale2d_nearest_fwd(const float* x, float* y, int64_t planes, int H, int W, int factor, float gain, void* stream);
int spk_fade_in_tanh_fwd(const float* a, const float* b, float* y, float alpha, int64_t n, void* stream);
/* Adjoints of the three stand-alone ops above (what autograd derives from styleganv1.py:52-63, :113-120, :132-136 and
 * stylegan.py:28-29 when the legacy modules are trained):
 * spk_pixelnorm_bwd: dx = r*dy - x * r^3 * mean_c(x*dy), r = rsqrt(mean_c x^2 + eps) (both forward spellings);
 * spk_blur2d_bwd: dx [planes,H,W] from dy [planes,Ho,Wo] (Ho = (H+2p-k)/stride+1), the same HOST filter as the forward;
 * spk_upscale2d_nearest_bwd: dx[h,w] = gain * sum of the factor x factor block of dy. */
int spk_pixelnorm_bwd(const float* x, const float* dy, float* dx, int B, int C, int64_t HW, float eps, void* stream);
int spk_blur2d_bwd(const float* dy, float* dx, const float* filter_host, int k, int64_t planes, int H, int W, int stride, void* stream);
int spk_upscale2d_nearest_bwd(const float* dy, float* dx, int64_t planes, int H, int W, int factor, float gain, void* stream);

/* ---- spectral normalisation of many layers at once (StyleDiscriminator) ------------------------------------------------
 * One power iteration + division for up to SPK_SN_MAX_GROUPS weight matrices W[R, C] (R = Cout, C = Cin*kh*kw, row major):
 *   power_iteration != 0:  v <- normalize(W^T u), u <- normalize(W v)   (in place, eps inside max(|.|, eps))
 *   sigma = u^T W v  (written to *sigma, a device float);  w_hat = W / sigma.
 * Five launches for all layers together, every sum in a fixed order (no atomics).
 * replaces: torch.nn.utils.spectral_norm's pre-forward hook (SpectralNorm.compute_weight) on the 16 layers wrapped at
 *   styleganv1.py:644-657,662-672 -- about a dozen ATen launches per layer and forward, six forwards per iteration.
 * spk_spectral_norm_bwd_grouped: with the group's `w_hat` slot holding the gradient G w.r.t. w_hat,
 *   dw = (G - <G, W> / sigma * u v^T) / sigma   -- autograd of W / sigma with u, v held constant, as in the hook.
 * `workspace`: device scratch of spk_spectral_norm_workspace_bytes(groups, n) bytes (same for both calls). */
#define SPK_SN_MAX_GROUPS 24
typedef struct spk_sn_group {
    const float* w;       /* [R, C] weight_orig */
    float* u;             /* [R] weight_u (updated in place when power_iteration) */
    float* v;             /* [C] weight_v */
    float* w_hat;         /* [R, C] out: W / sigma        (backward: in: the gradient G w.r.t. w_hat) */
    float* sigma;         /* [1]    out: the spectral norm estimate (backward: in) */
    float* dw;            /* backward only: [R, C] out */
    int32_t R, C;
    int32_t accumulate;   /* backward only: 1 = dw += (a discriminator runs several forwards before one backward, train.py:160-182:
                           * the passes' gradients of one weight_orig add up inside the kernel instead of as separate tensors) */
    int32_t reserved;
} spk_sn_group;
int64_t spk_spectral_norm_workspace_bytes(const spk_sn_group* groups, int n_groups);
int spk_spectral_norm_grouped(const spk_sn_group* groups, int n_groups, int power_iteration, float eps, void* workspace,
                              int64_t workspace_bytes, void* stream);
int spk_spectral_norm_bwd_grouped(const spk_sn_group* groups, int n_groups, void* workspace, int64_t workspace_bytes, void* stream);
/* out[c] (+)= sum_b sums[b][row][c] over a [B][rows][C] array of per-plane sums (spk_epilogue_bwd's output): the bias gradient of a
 * conv + bias + LeakyReLU layer (styleganv1.py:662-672), accumulated in place across the passes of one backward. */
int spk_plane_sums_reduce(const float* sums, int B, int rows, int C, int row, float* out, int accumulate, void* stream);

/* ---- launch lists: a whole module forward per C call ---------------------------------------------------------------
 * The reference's callers run a decoder pass as one Python call (model.py:113-114 `self.Gd(gen_input)`,
 * styleganv1.py:593-610 SynthesisNetwork.forward); behind it sit ~25 kernel launches whose descriptors depend only on
 * (module, batch size, device).  spk_launch_list enqueues a HOST array of pre-built ops in order on `stream` -- exactly
 * the entry points above with exactly their arguments -- so the per-call host cost is one crossing plus ~4 us per
 * launch instead of a descriptor build, a config query and a crossing per launch.  `kind_mask`: bit k set = ops of kind
 * k are launched, the others skipped (~0u = everything; measurement harnesses time one kernel family of a step this
 * way, e.g. 1u << SPK_OP_CONV2D, without any switch inside the library).  Stops at the first failing op and returns
 * its code.  replaces: nothing in the reference's arithmetic -- the host-side loop over ATen calls. */
enum {
    SPK_OP_CONV2D = 1,            /* desc: spk_conv2d_desc          -> spk_conv2d_fwd */
    SPK_OP_FC = 2,                /* desc: spk_fc_args              -> spk_fc_fwd */
    SPK_OP_FC_GROUPED = 3,        /* desc: spk_fc_grouped_args      -> spk_fc_grouped_fwd */
    SPK_OP_BIAS_NOISE_STYLE = 4,  /* desc: spk_bias_noise_style_args-> spk_bias_noise_style_fwd */
    SPK_OP_TORGB = 5,             /* desc: spk_torgb_args           -> spk_conv1x1_small_fwd / spk_torgb_mod_skip_fwd */
    SPK_OP_DEMOD_GROUPED = 6,     /* desc: spk_demod_grouped_args   -> spk_modconv_demod_grouped */
    SPK_OP_PIXELNORM = 7,         /* desc: spk_pixelnorm_args       -> spk_pixelnorm_fwd */
    SPK_OP_UPSAMPLE2X = 8         /* desc: spk_upsample2x_args      -> spk_upsample2x_fwd (the x2 image of a block whose conv1
                                   * runs as Winograd, styleganv1.py:621,624) */
};
typedef struct spk_op { int32_t kind; int32_t reserved; const void* desc; } spk_op;
typedef struct spk_fc_args {
    const float* x; int64_t x_stride; const float* w; const float* bias; float* out; int64_t out_stride;
    int32_t B, I, O; float wmul, bmul, slope;
} spk_fc_args;
typedef struct spk_fc_grouped_args { const spk_fc_group* groups; int32_t n_groups, B; } spk_fc_grouped_args;
typedef struct spk_bias_noise_style_args {
    const float* x; int64_t x_batch_stride; const float* bias; const float* noise_w; const float* noise; const float* style;
    int64_t style_stride; float* y; int32_t B, C, HW, reserved;
} spk_bias_noise_style_args;
typedef struct spk_torgb_args {   /* mod NULL: plain 1x1 (styleganv1.py:607); else modulated (+ optional skip [B,O,H/2,W/2]) */
    const float* x; const float* w; const float* mod; const float* bias; const float* skip; float* y;
    int32_t B, C, O, H, W; float in_scale;
} spk_torgb_args;
typedef struct spk_demod_grouped_args { const spk_demod_group* groups; int32_t n_groups, B; float eps; int32_t reserved; } spk_demod_grouped_args;
typedef struct spk_pixelnorm_args { const float* x; float* y; int32_t B, C; int64_t HW; float eps; int32_t sqrt_form; } spk_pixelnorm_args;
typedef struct spk_upsample2x_args { const float* x; float* y; int64_t planes; int32_t Hin, Win; int32_t zero_border, reserved; } spk_upsample2x_args;
int spk_launch_list(const spk_op* ops, int n_ops, uint32_t kind_mask, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPK_H_ */
