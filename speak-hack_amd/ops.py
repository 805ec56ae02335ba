"""Thin Python launchers over the C ABI (include/spk.h).  Forward primitives; every output buffer is
a torch allocation (caching allocator, so no hipMalloc in steady state) and every launch goes on
torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import contextlib
import weakref

import os

import torch

from . import _lib as L



def _launch_conv2d(desc):
    """The one place a conv launch leaves Python (measurement harnesses wrap THIS function from outside the package;
    the product path carries no switches)."""
    L.check(L.lib().spk_conv2d_fwd(C.byref(desc), L.stream_ptr()), "spk_conv2d_fwd")


# --------------------------------------------------------------------------------------------------
class PackedConvWeight:
    """A [Cout,Cin,k,k] weight re-laid for one tile config of the MFMA conv kernel
    ([co_tile][ci_chunk][tap][ci][co], zero padded).  Re-packed when the source tensor changes: the cache is keyed
    on the tensor OBJECT (a weak reference -- a fresh temporary such as ``w * scale`` can be handed the address, shape
    and version 0 of last step's temporary by the caching allocator, so pointer + version alone would serve stale
    weights after an optimizer step) plus its autograd version counter and storage pointer."""

    def __init__(self):
        self._cache = {}

    def get(self, weight: torch.Tensor, config: int, transpose_flip: bool = False) -> torch.Tensor:
        key = (config, transpose_flip)
        stamp = (weight.data_ptr(), weight._version, tuple(weight.shape))
        hit = self._cache.get(key)
        if hit is not None and hit[0] == stamp and hit[2]() is weight:
            return hit[1]
        packed = pack_conv_weight(weight.detach(), config, transpose_flip)
        self._cache[key] = (stamp, packed, weakref.ref(weight))
        return packed

    def get_bf16x3(self, weight: torch.Tensor, transpose_flip: bool = False) -> torch.Tensor:
        """The bf16 hi / lo image of the opt-in split-precision conv (``conv3x3_bf16x3``), cached the same way."""
        key = ("bf16x3", bool(transpose_flip))
        stamp = (weight.data_ptr(), weight._version, tuple(weight.shape))
        hit = self._cache.get(key)
        if hit is not None and hit[0] == stamp and hit[2]() is weight:
            return hit[1]
        packed = pack_conv_weight_bf16x3(weight.detach(), transpose_flip=transpose_flip)
        self._cache[key] = (stamp, packed, weakref.ref(weight))
        return packed

    def get_wino(self, weight: torch.Tensor, transpose_flip: bool = False) -> torch.Tensor:
        """The transformed image U = G g G^T of the fp32 Winograd conv (``conv3x3_wino``), cached the same way."""
        key = ("wino", bool(transpose_flip))
        stamp = (weight.data_ptr(), weight._version, tuple(weight.shape))
        hit = self._cache.get(key)
        if hit is not None and hit[0] == stamp and hit[2]() is weight:
            return hit[1]
        packed = pack_conv_weight_wino(weight.detach(), transpose_flip=transpose_flip)
        self._cache[key] = (stamp, packed, weakref.ref(weight))
        return packed

    def clear(self):
        self._cache.clear()


def conv2d_pick_config(k, stride, B, Cin, Cout, H, W) -> int:
    """Tile config the library's heuristic picks; H, W are the OUTPUT size."""
    cfg = L.lib().spk_conv2d_pick_config(k, k, stride, B, Cin, Cout, H, W)
    if cfg < 0:
        raise L.SpkError(f"spk_conv2d_pick_config: {L.lib().spk_last_error().decode()}")
    return cfg


def conv2d_config_fits(config: int, k, stride, B, Cin, Cout, H, W) -> bool:
    """Whether tile config ``config`` is built for this kernel and can host this problem."""
    return L.lib().spk_conv2d_workspace_bytes(int(config), 1, k, k, stride, B, Cin, Cout, H, W) >= 0


def conv2d_config_info(config: int):
    co, ci, px = C.c_int(), C.c_int(), C.c_int()
    L.check(L.lib().spk_conv2d_config_info(config, C.byref(co), C.byref(ci), C.byref(px)), "spk_conv2d_config_info")
    return co.value, ci.value, px.value


def pack_conv_weight(weight: torch.Tensor, config: int, transpose_flip=False, out=None) -> torch.Tensor:
    """``transpose_flip``: False = the forward operator; True (1) = the data-gradient operator of a stride-1 conv (run
    by the forward kernel); 2 = the four output-parity 2x2 kernels of a 3x3 STRIDE-2 conv's data gradient; 3 = those of a
    ConvTranspose2d(4, stride 2, pad 1) forward (``weight`` is then [Cin,Cout,4,4])."""
    Cout, Cin, kh, kw = weight.shape
    tf = int(transpose_flip)
    if tf == 2:
        n = L.lib().spk_conv2d_packed_floats(config, 2, 2, Cout, 4 * Cin)
    elif tf == 3:        # ConvTranspose2d weight [Cin,Cout,4,4] -> the four parity 2x2 kernels (SPK_CONV_TRANSPOSE4X4_S2)
        Cin, Cout = weight.shape[:2]
        n = L.lib().spk_conv2d_packed_floats(config, 2, 2, Cin, 4 * Cout)
    else:
        n = L.lib().spk_conv2d_packed_floats(config, kh, kw, Cout if tf else Cin, Cin if tf else Cout)
    if n <= 0:
        raise L.SpkError("spk_conv2d_packed_floats: bad arguments")
    if out is None:
        out = torch.empty(n, device=weight.device, dtype=torch.float32)
    elif out.numel() != n or not out.is_contiguous() or out.device != weight.device:
        raise L.SpkError(f"pack_conv_weight: out must be a contiguous buffer of {n} floats on {weight.device}")
    L.check(L.lib().spk_conv2d_pack_weights(L.dptr(weight.contiguous(), "weight"), L.dptr(out), kh, kw, Cin, Cout,
                                            config, tf, L.stream_ptr()),
            "spk_conv2d_pack_weights")
    return out


PACK_LIST_MAX = 8


def pack_conv_weights_list(weights, config: int, transpose_flip=False, out=None) -> torch.Tensor:
    """The packed images of several same-shape weights one after another (what a grouped launch reads), ONE launch per
    ``PACK_LIST_MAX`` tensors (``spk_conv2d_pack_weights_list``) instead of a pack per tensor and a concatenation."""
    w0 = weights[0]
    Cout, Cin, kh, kw = w0.shape
    tf = int(transpose_flip)
    if tf == 2:          # the stride-2 data-gradient form (see pack_conv_weight)
        n1 = L.lib().spk_conv2d_packed_floats(config, 2, 2, Cout, 4 * Cin)
    elif tf == 3:
        raise L.SpkError("pack_conv_weights_list: transpose_flip 0, 1 or 2")
    else:
        n1 = L.lib().spk_conv2d_packed_floats(config, kh, kw, Cout if tf else Cin, Cin if tf else Cout)
    if n1 <= 0:
        raise L.SpkError("spk_conv2d_packed_floats: bad arguments")
    if out is None:
        out = torch.empty(n1 * len(weights), device=w0.device, dtype=torch.float32)
    elif out.numel() != n1 * len(weights) or not out.is_contiguous() or out.device != w0.device:
        raise L.SpkError(f"pack_conv_weights_list: out must be a contiguous buffer of {n1 * len(weights)} floats on {w0.device}")
    ws = [w.contiguous() for w in weights]
    for w in ws:
        if w.shape != w0.shape or w.dtype != torch.float32 or w.device != w0.device:
            raise L.SpkError("pack_conv_weights_list: the weights must share shape, dtype and device")
    for i in range(0, len(ws), PACK_LIST_MAX):
        part = ws[i:i + PACK_LIST_MAX]
        arr = (C.c_void_p * len(part))(*[w.data_ptr() for w in part])
        L.check(L.lib().spk_conv2d_pack_weights_list(arr, len(part), out.data_ptr() + 4 * n1 * i, kh, kw, Cin, Cout, config, tf,
                                                     L.stream_ptr()),
                "spk_conv2d_pack_weights_list")
    return out


_workspaces = {}
_retired = []            # outgrown scratch buffers: never freed (a captured hipGraph may have their address baked in)


def _workspace(device, nbytes: int):
    """Scratch for split-K partial sums, one buffer per (device, stream), grown on demand.  Reuse is stream-ordered:
    every consumer of the scratch is enqueued on the same stream right behind its producer, and two streams never
    share a buffer.  An outgrown buffer is retired, not freed, so launches already queued -- or captured into a
    hipGraph -- keep writing into memory nobody else owns.  During stream capture a larger request is served by a
    tensor of the capture's own memory pool (it lives and dies with that graph) and the cache is left alone; warm
    the shapes up before capturing, as bench.py does, to share one buffer."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        fresh = torch.empty((max(nbytes, 1 << 20) + 3) // 4, device=device, dtype=torch.float32)
        if torch.cuda.is_current_stream_capturing():
            return fresh
        if ws is not None:
            _retired.append(ws)
        ws = _workspaces[key] = fresh
    return ws


_side_streams = {}


def side_stream(device):
    """The device's second HIP stream: a trunk backward queues its weight gradients there, behind the producer of their
    operands, so that a weight gradient's fixed costs (first-tile latency, slab store, slab reduce: 25-35 us of a 70-180 us
    launch) and the data-gradient chain's own tails fill each other's idle CUs (tools/lab_wgrad_overlap.py: 9-19 % on the
    trunk's layers).  The backward joins the stream before it returns.  ``SPK_WGRAD_STREAM=0``: None (everything in order on
    the current stream)."""
    if os.environ.get("SPK_WGRAD_STREAM", "1") == "0":
        return None
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device, priority=int(os.environ.get("SPK_WGRAD_STREAM_PRIORITY", "0")))
    return st


def side_stream_delay():
    """Test hook (``SPK_WGRAD_STREAM_DELAY`` = spin cycles): stall the CURRENT stream before a weight gradient is launched on
    it.  With the second stream held back by milliseconds, any consumer that does not wait for it reads garbage -- the
    second-stream equality tests run once this way (tests/test_second_stream_delay_gpu.py)."""
    n = int(os.environ.get("SPK_WGRAD_STREAM_DELAY", "0"))
    if n > 0:
        torch.cuda._sleep(n)


def conv_out_size(n, k, stride):
    return (n + 2 * ((k - 1) // 2) - k) // stride + 1


def conv2d_fused(x, w_packed, Cout: int, k: int = 3, stride: int = 1, *, bias=None, noise_w=None, noise=None,
                 style=None, style_stride=None, upsample=False, lrelu_slope=None, out_scale=1.0, in_affine=None,
                 stats=None, config=-1, ksplit=0, out=None, accumulate=False, out_pre=None, batch_scale=None, demod=None,
                 act_gain=1.0, up_fir=False, groups=1, shared_input=False, accum_half=None, out_scale_dev=None):
    """y = style(lrelu(conv_kxk(in(x)) * out_scale + bias + noise_w*noise)) -- one launch.

    ``accum_half`` [B, groups*Cout, ceil(H/2), ceil(W/2)]: added to y at the even pixels (``SPK_EPI_ACCUM_HALF``; the
    GEMM form of a stride-1 1x1, configs 14 / 15).  ``out_scale_dev``: a one-element device tensor multiplied into
    ``out_scale`` (1 / sigma of a spectrally normalised weight whose packed image is that of ``weight_orig``).

    ``groups`` > 1: that many independent convs of the same shape in one launch (``Cout`` per group; ``w_packed`` = the
    groups' packed images concatenated; x carries the groups' input channels side by side, or -- ``shared_input`` -- one
    set of channels every group reads); y has ``groups * Cout`` channels.

    ``in``: identity; or bilinear x2 (``upsample``; x is [B,Cin,H/2,W/2]); or ``max(x*a+b, 0)`` per
    input channel (``in_affine=(a, b)``: the producer's BatchNorm+ReLU, folded into staging).
    ``style``: rows [s0(Cout) | s1(Cout)] with row stride ``style_stride``.  ``stats``: fp64 [2*Cout],
    accumulates sum / sum of squares of y over (b,h,w) (BatchNorm batch statistics).
    """
    B, Cin, Hs, Ws = x.shape
    G = int(groups)
    if G > 1 and not shared_input:
        if Cin % G:
            raise L.SpkError(f"conv2d_fused: {Cin} input channels do not split into {G} groups")
        Cin //= G
    if upsample:
        H, W = 2 * Hs, 2 * Ws
    else:
        H, W = conv_out_size(Hs, k, stride), conv_out_size(Ws, k, stride)
    if out is None:
        out = torch.empty((B, G * Cout, H, W), device=x.device, dtype=torch.float32)
    flags = 0
    if bias is not None:
        flags |= L.EPI_BIAS
    if noise is not None:
        if noise_w is None or noise.numel() != B * H * W:
            raise L.SpkError(f"conv2d_fused: noise must be [B,1,H,W]={B, 1, H, W}, got {tuple(noise.shape)}")
        flags |= L.EPI_NOISE
    if lrelu_slope is not None:
        flags |= L.EPI_LRELU
    if style is not None:
        if style_stride is None:
            style_stride = style.stride(0) if style.dim() == 2 else 2 * Cout
        flags |= L.EPI_STYLE
    if upsample:
        flags |= L.CONV_UPSAMPLE2X | (L.CONV_UP_FIR1331 if up_fir else 0)
    if accumulate:
        flags |= L.EPI_ACCUM
    if in_affine is not None:
        flags |= L.CONV_IN_AFFINE_RELU
    if accum_half is not None:
        if tuple(accum_half.shape) != (B, G * Cout, (H + 1) // 2, (W + 1) // 2) or not accum_half.is_contiguous():
            raise L.SpkError(f"conv2d_fused: accum_half must be a contiguous {(B, G * Cout, (H + 1) // 2, (W + 1) // 2)} tensor")
        flags |= L.EPI_ACCUM_HALF
    if batch_scale is not None:          # modulated convolution: s[B,Cin] applied to the input while staging
        if tuple(batch_scale.shape) != (B, Cin) or in_affine is not None:
            raise L.SpkError("conv2d_fused: batch_scale must be [B,Cin] and excludes in_affine")
        flags |= L.CONV_IN_BATCH_SCALE
    if demod is not None and tuple(demod.shape) != (B, Cout):
        raise L.SpkError("conv2d_fused: demod must be [B,Cout]")
    slots = 0
    if stats is not None:
        slots = stats.numel() // (2 * G * Cout)
        if (stats.dtype != torch.float64 or not stats.is_cuda or not stats.is_contiguous() or slots < 1
                or stats.numel() != slots * 2 * G * Cout):
            raise L.SpkError("conv2d_fused: stats must be a contiguous float64 HIP tensor of slots*2*groups*Cout elements "
                             "(see stats_slots)")
        flags |= L.EPI_STATS
    if config < 0:
        config = conv2d_pick_config(k, stride, B, Cin, Cout, H, W)
        if batch_scale is not None and config < 4:
            config += 4                  # the modulated variant is built for the half-depth-chunk configs
    ws_bytes = L.lib().spk_conv2d_workspace_bytes_grouped(int(config), int(ksplit), k, k, stride, B, Cin, Cout, H, W, G)
    if ws_bytes < 0:
        raise L.SpkError(f"conv2d_fused: config {config} cannot host k={k} s={stride} shape {(B, Cin, Cout, H, W)}")
    ws = _workspace(x.device, ws_bytes) if ws_bytes > 0 else None
    d = L.Conv2dDesc(x=L.dptr(x, "x"), w_packed=L.dptr(w_packed, "w_packed"), bias=L.dptr(bias, "bias"),
                     noise_w=L.dptr(noise_w, "noise_w") if noise is not None else None,
                     noise=L.dptr(noise, "noise"), style=_style_ptr(style),
                     in_scale=(L.dptr(in_affine[0], "in_scale") if in_affine is not None
                               else L.dptr(batch_scale, "batch_scale")),
                     in_shift=L.dptr(in_affine[1], "in_shift") if in_affine is not None else None,
                     out_scale_bc=L.dptr(demod, "demod"), act_gain=float(act_gain),
                     stats=stats.data_ptr() if stats is not None else None, y=L.dptr(out, "out"),
                     y_pre=L.dptr(out_pre, "out_pre"), B=B, Cin=Cin, Cout=Cout, H=H, W=W, Hin=Hs, Win=Ws, kh=k, kw=k, stride=stride,
                     style_stride=int(style_stride or 0), flags=flags,
                     lrelu_slope=float(lrelu_slope if lrelu_slope is not None else 1.0), out_scale=float(out_scale),
                     config=int(config), ksplit=int(ksplit), workspace=ws.data_ptr() if ws is not None else None,
                     workspace_bytes=ws.numel() * 4 if ws is not None else 0, groups=G,
                     group_in_stride=0 if (shared_input or G == 1) else Cin, stats_slots=slots,
                     accum_half=L.dptr(accum_half, "accum_half"), out_scale_dev=L.dptr(out_scale_dev, "out_scale_dev"))
    _launch_conv2d(d)
    return out


# ---- the opt-in split-precision speed path (SPK_CONV_BF16X3, csrc/conv3x3_bf16x3.hip) ------------------------------
def bf16x3_supported(B, Cin, Cout, H, W) -> bool:
    return bool(L.lib().spk_conv2d_bf16x3_supported(B, Cin, Cout, H, W))


def pack_conv_weight_bf16x3(weight: torch.Tensor, out=None, transpose_flip=False) -> torch.Tensor:
    """[Cout,Cin,3,3] fp32 -> the bf16 hi / lo image of the BF16X3 conv (a byte tensor).  ``transpose_flip``: the image of
    the conv's data-gradient operator (run it with Cin / Cout exchanged)."""
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise L.SpkError("pack_conv_weight_bf16x3: 3x3 kernels only")
    n = L.lib().spk_conv2d_packed_bytes_bf16x3(Cout, Cin) if transpose_flip else L.lib().spk_conv2d_packed_bytes_bf16x3(Cin, Cout)
    if out is None:
        out = torch.empty(n, device=weight.device, dtype=torch.uint8)
    elif out.numel() * out.element_size() != n or not out.is_contiguous():
        raise L.SpkError(f"pack_conv_weight_bf16x3: out must hold {n} bytes")
    L.check(L.lib().spk_conv2d_pack_weights_bf16x3_tf(L.dptr(weight.contiguous(), "weight"), out.data_ptr(), Cin, Cout,
                                                      1 if transpose_flip else 0, L.stream_ptr()),
            "spk_conv2d_pack_weights_bf16x3_tf")
    return out


# ---- opt-in reduced-precision TRAINING: forward convs and data gradients of the 3x3 stride-1 layers on the bf16 pipe (operands
# split hi + lo, fp32 accumulation: ~2e-5 per layer), weight gradients and everything else exact.  The reference's own training
# config runs IRFD.forward under fp16 autocast (config.yaml:28, train.py:334); the default here stays exact fp32.
TRAIN_CONV_PRECISION = "f32"
TRAIN_BF16X3_MIN_PIXELS = 2048


@contextlib.contextmanager
def train_conv_precision(precision: str):
    """``with ops.train_conv_precision("bf16x3"):`` around a training step (forward AND backward)."""
    global TRAIN_CONV_PRECISION
    if precision not in ("f32", "bf16x3"):
        raise ValueError("precision must be 'f32' or 'bf16x3'")
    prev, TRAIN_CONV_PRECISION = TRAIN_CONV_PRECISION, precision
    try:
        yield
    finally:
        TRAIN_CONV_PRECISION = prev


def train_bf16x3(B, Cin, Cout, H, W) -> bool:
    """Whether a 3x3 stride-1 conv with this OUTPUT shape takes the split-precision kernel under the training switch."""
    return (TRAIN_CONV_PRECISION == "bf16x3" and B * H * W >= TRAIN_BF16X3_MIN_PIXELS and bf16x3_supported(B, Cin, Cout, H, W)
            and not (H * W <= 256 and use_wino(B, Cin, Cout, H, W)))      # (a <= 16^2 layer: the sliced fp32 Winograd launch is faster, and exact)


def conv3x3_bf16x3(x, w_packed, Cout, *, bias=None, noise_w=None, noise=None, style=None, upsample=False, up_fir=False,
                   lrelu_slope=None, out_scale=1.0, batch_scale=None, demod=None, act_gain=1.0, out=None, out_pre=None):
    """Forward 3x3 stride-1 conv with the fused decoder epilogue on the bf16 matrix pipe, operands split hi + lo (three MFMAs
    per product, fp32 accumulation): ~3e-5 rel-L2 through the decoder, 5.3x the exact-f32 matrix rate.  ``out_pre``: also
    keep the value before the style stage (a training forward)."""
    B, Cin, Hs, Ws = x.shape
    H, W = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
    if out is None:
        out = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32)
    flags = L.CONV_BF16X3 | (L.EPI_BIAS if bias is not None else 0) | (L.EPI_NOISE if noise is not None else 0) | \
        (L.EPI_LRELU if lrelu_slope is not None else 0) | (L.EPI_STYLE if style is not None else 0) | \
        (L.CONV_UPSAMPLE2X if upsample else 0) | (L.CONV_UP_FIR1331 if (upsample and up_fir) else 0) | \
        (L.CONV_IN_BATCH_SCALE if batch_scale is not None else 0)
    d = L.Conv2dDesc(x=L.dptr(x, "x"), w_packed=w_packed.data_ptr(), bias=L.dptr(bias, "bias"),
                     noise_w=L.dptr(noise_w, "noise_w") if noise is not None else None, noise=L.dptr(noise, "noise"),
                     style=_style_ptr(style), in_scale=L.dptr(batch_scale, "batch_scale"), in_shift=None,
                     out_scale_bc=L.dptr(demod, "demod"), act_gain=float(act_gain), stats=None, y=L.dptr(out, "out"), y_pre=L.dptr(out_pre, "out_pre"),
                     B=B, Cin=Cin, Cout=Cout, H=H, W=W, Hin=Hs, Win=Ws, kh=3, kw=3, stride=1,
                     style_stride=int(style.stride(0)) if style is not None else 0, flags=flags,
                     lrelu_slope=float(lrelu_slope if lrelu_slope is not None else 1.0), out_scale=float(out_scale), config=-1,
                     ksplit=1, workspace=None, workspace_bytes=0, groups=1, group_in_stride=0, stats_slots=0)
    _launch_conv2d(d)
    return out


# ---- Winograd F(2x2, 3x3) on the f32 MFMA pipe (SPK_CONV_WINOGRAD, csrc/conv3x3_wino_f32.hip) ------------------------------------
# Which fp32 algorithm a 3x3 stride-1 conv runs: "auto" -- Winograd F(2x2, 3x3) where the kernel serves the shape and the launch
# fills the chip, the direct (fmaf-chain) kernel elsewhere; "direct" -- the direct kernel everywhere (SPK_CONV3X3_ALGO=direct: the
# previous rounds' arithmetic, bit for bit).  Both are fp32 operands, fp32 products, fp32 accumulation.
CONV3X3_ALGO = os.environ.get("SPK_CONV3X3_ALGO", "auto")
WINO_MIN_WORKGROUPS = 192          # one workgroup per CU: below ~3/4 of a round the direct kernel's split-K wins


@contextlib.contextmanager
def conv3x3_algo(algo: str):
    """``with ops.conv3x3_algo("direct"):`` -- scoped override of ``CONV3X3_ALGO`` (plans are keyed on it)."""
    global CONV3X3_ALGO
    if algo not in ("auto", "direct"):
        raise ValueError("conv3x3 algo must be 'auto' or 'direct'")
    prev, CONV3X3_ALGO = CONV3X3_ALGO, algo
    try:
        yield
    finally:
        CONV3X3_ALGO = prev


def use_wino(B, Cin, Cout, H, W, groups=1) -> bool:
    """Whether a 3x3 stride-1 conv with this OUTPUT shape (plain input) goes to the Winograd kernel under ``CONV3X3_ALGO``: the
    kernel serves the shape and -- with the contraction split into the slices ``wino_ksplit`` picks -- fills the chip.
    ``groups``: Cin / Cout per group (the launch has groups x the channel tiles)."""
    if CONV3X3_ALGO != "auto" or not wino_supported(B, Cin, Cout, H, W):
        return False
    rw, rh = (32, 8) if (W % 32 == 0 and H % 8 == 0) else (16, 16)
    tiles = groups * ((Cout + 63) // 64)
    return B * (H // rh) * (W // rw) * tiles * wino_ksplit(B, Cin, groups * Cout, H, W) >= WINO_MIN_WORKGROUPS


def wino_ksplit(B, Cin, Cout, H, W, want=0) -> int:
    """Slices of the channel contraction the Winograd launch will use (1: none; > 1: partial sums through the split-K workspace)."""
    return int(L.lib().spk_conv2d_wino_ksplit(int(want), B, Cin, Cout, H, W))


def wino_supported(B, Cin, Cout, H, W) -> bool:
    """Whether the fp32 Winograd kernel serves a 3x3 stride-1 conv with this OUTPUT shape (whole 32 x 8 regions, Cin % 16 == 0)."""
    return bool(L.lib().spk_conv2d_wino_supported(B, Cin, Cout, H, W))


def pack_conv_weight_wino(weight: torch.Tensor, out=None, transpose_flip=False) -> torch.Tensor:
    """[Cout,Cin,3,3] fp32 -> the transformed image U = G g G^T of the Winograd conv.  ``transpose_flip``: the image of the
    conv's data-gradient operator (run it with Cin / Cout exchanged)."""
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise L.SpkError("pack_conv_weight_wino: 3x3 kernels only")
    n = (L.lib().spk_conv2d_packed_bytes_wino(Cout, Cin) if transpose_flip else L.lib().spk_conv2d_packed_bytes_wino(Cin, Cout)) // 4
    if out is None:
        out = torch.empty(n, device=weight.device, dtype=torch.float32)
    elif out.numel() != n or out.dtype != torch.float32 or not out.is_contiguous():
        raise L.SpkError(f"pack_conv_weight_wino: out must hold {n} floats")
    L.check(L.lib().spk_conv2d_pack_weights_wino(L.dptr(weight.contiguous(), "weight"), L.dptr(out), Cin, Cout,
                                                 1 if transpose_flip else 0, L.stream_ptr()), "spk_conv2d_pack_weights_wino")
    return out


WINO_PACK_MAX = 32


def pack_conv_weights_wino_into(weights, outs, transpose_flip=False):
    """Winograd images of several [Cout,Cin,3,3] weights into the given buffers (views of one tensor for a grouped launch): one
    spk_conv2d_pack_weights_wino_list launch per 32."""
    for i in range(0, len(weights), WINO_PACK_MAX):
        ws, os_ = weights[i:i + WINO_PACK_MAX], outs[i:i + WINO_PACK_MAX]
        n = len(ws)
        keep = [w.contiguous() for w in ws]
        L.check(L.lib().spk_conv2d_pack_weights_wino_list((C.c_void_p * n)(*[L.dptr(w, "weight") for w in keep]),
                                                          (C.c_void_p * n)(*[o.data_ptr() for o in os_]),
                                                          (C.c_int * n)(*[w.shape[1] for w in ws]), (C.c_int * n)(*[w.shape[0] for w in ws]),
                                                          (C.c_int * n)(*[1 if transpose_flip else 0] * n), n, L.stream_ptr()),
                "spk_conv2d_pack_weights_wino_list")



def prepack_wino(items):
    """``items`` = [(PackedConvWeight cache, weight, transpose_flip), ...]: fill every STALE Winograd image in ONE launch
    (spk_conv2d_pack_weights_wino_list) -- the same bits ``get_wino`` would produce one launch at a time."""
    todo = []
    for pk, w, tf in items:
        key = ("wino", bool(tf))
        stamp = (w.data_ptr(), w._version, tuple(w.shape))
        hit = pk._cache.get(key)
        if hit is not None and hit[0] == stamp and hit[2]() is w:
            continue
        Cout, Cin = w.shape[:2]
        n = (L.lib().spk_conv2d_packed_bytes_wino(Cout, Cin) if tf else L.lib().spk_conv2d_packed_bytes_wino(Cin, Cout)) // 4
        todo.append((pk, key, stamp, w, bool(tf), torch.empty(n, device=w.device, dtype=torch.float32)))
    for i in range(0, len(todo), WINO_PACK_MAX):
        part = todo[i:i + WINO_PACK_MAX]
        n = len(part)
        keep = [t[3].detach().contiguous() for t in part]          # (alive until the launch is queued)
        ws = (C.c_void_p * n)(*[L.dptr(w_, "weight") for w_ in keep])
        outs = (C.c_void_p * n)(*[t[5].data_ptr() for t in part])
        cin = (C.c_int * n)(*[t[3].shape[1] for t in part])
        cout = (C.c_int * n)(*[t[3].shape[0] for t in part])
        tfs = (C.c_int * n)(*[1 if t[4] else 0 for t in part])
        L.check(L.lib().spk_conv2d_pack_weights_wino_list(ws, outs, cin, cout, tfs, n, L.stream_ptr()), "spk_conv2d_pack_weights_wino_list")
        for pk, key, stamp, w, tf, out in part:
            pk._cache[key] = (stamp, out, weakref.ref(w))


def upsample2x(x, zero_border=False):
    """The x2 image: bilinear (edge taps clamped), or -- ``zero_border`` -- upfirdn2d(up=2, [1,3,3,1], pad (2,1)) (the same taps,
    neighbours outside the image zero).  What a Winograd x2 layer reads."""
    B, Cc, H, W = x.shape
    y = torch.empty((B, Cc, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_upsample2x_fwd(L.dptr(x.contiguous(), "x"), L.dptr(y), B * Cc, H, W, 1 if zero_border else 0, L.stream_ptr()),
            "spk_upsample2x_fwd")
    return y


def conv3x3_wino(x, w_packed, Cout, *, bias=None, noise_w=None, noise=None, style=None, style_stride=None, lrelu_slope=None,
                 out_scale=1.0, act_gain=1.0, out=None, out_pre=None, accumulate=False, out_scale_dev=None, batch_scale=None, demod=None,
                 ksplit=0, rgb=None, rgb_out=None, store_out=True, groups=1):
    """3x3 stride-1 pad-1 conv with the fused decoder epilogue as Winograd F(2x2, 3x3): fp32 throughout, 2.25x fewer matrix
    instructions than the direct form, 1e-6-class rel-L2 against it.  ``w_packed`` from ``pack_conv_weight_wino``.
    ``rgb`` = (weight [3,Cout,1,1], bias [3] | None): the 1x1 conv of styleganv1.py:607 inside the epilogue (Cout <= 64, unsliced);
    returns (out, rgb image), and with ``store_out=False`` (None, rgb image) -- the activation is then never written.
    ``groups`` > 1: that many independent convs in one launch (``Cout`` per group, x carries the groups' input channels side by side,
    ``w_packed`` = the groups' images one after another); plain or ``accumulate`` only -- the encoders' data gradients."""
    B, Cx, H, W = x.shape
    G = int(groups)
    if Cx % G:
        raise L.SpkError(f"conv3x3_wino: {Cx} input channels do not split into {G} groups")
    Cin = Cx // G
    if G > 1 and any(t is not None for t in (bias, noise, style, out_pre, batch_scale, demod, rgb, lrelu_slope)):
        raise L.SpkError("conv3x3_wino: a grouped launch is plain (accumulate allowed)")
    if rgb is not None:
        if Cout > 64 or out_pre is not None or accumulate or batch_scale is not None or tuple(rgb[0].shape[:2]) != (3, Cout):
            raise L.SpkError("conv3x3_wino: a fused toRGB needs Cout <= 64, weight [3,Cout,1,1], no out_pre / accumulate / modulation")
        if rgb_out is None:
            rgb_out = torch.empty((B, 3, H, W), device=x.device, dtype=torch.float32)
        ksplit = 1
        rgb_w2d = rgb[0].detach().reshape(3, Cout).contiguous()      # (a local: alive until the launch is queued)
    if out is None and (rgb is None or store_out):
        out = torch.empty((B, G * Cout, H, W), device=x.device, dtype=torch.float32)
    if noise is not None and (noise_w is None or noise.numel() != B * H * W):
        raise L.SpkError(f"conv3x3_wino: noise must be [B,1,H,W]={B, 1, H, W}, got {tuple(noise.shape)}")
    if style is not None and style_stride is None:
        style_stride = style.stride(0) if style.dim() == 2 else 2 * Cout
    if batch_scale is not None and tuple(batch_scale.shape) != (B, Cin):
        raise L.SpkError("conv3x3_wino: batch_scale must be [B,Cin]")
    if demod is not None and (batch_scale is None or tuple(demod.shape) != (B, Cout)):
        raise L.SpkError("conv3x3_wino: demod must be [B,Cout] and goes with batch_scale")
    flags = L.CONV_WINOGRAD | (L.EPI_BIAS if bias is not None else 0) | (L.EPI_NOISE if noise is not None else 0) | \
        (L.EPI_LRELU if lrelu_slope is not None else 0) | (L.EPI_STYLE if style is not None else 0) | (L.EPI_ACCUM if accumulate else 0) | \
        (L.CONV_IN_BATCH_SCALE if batch_scale is not None else 0) | (L.EPI_TORGB if rgb is not None else 0)
    d = L.Conv2dDesc(x=L.dptr(x, "x"), w_packed=L.dptr(w_packed, "w_packed"), bias=L.dptr(bias, "bias"),
                     noise_w=L.dptr(noise_w, "noise_w") if noise is not None else None, noise=L.dptr(noise, "noise"),
                     style=_style_ptr(style), in_scale=L.dptr(batch_scale, "batch_scale"), in_shift=None, out_scale_bc=L.dptr(demod, "demod"),
                     act_gain=float(act_gain), stats=None,
                     y=L.dptr(out, "out"), y_pre=L.dptr(out_pre, "out_pre"), B=B, Cin=Cin, Cout=Cout, H=H, W=W, Hin=H, Win=W,
                     kh=3, kw=3, stride=1, style_stride=int(style_stride or 0), flags=flags,
                     lrelu_slope=float(lrelu_slope if lrelu_slope is not None else 1.0), out_scale=float(out_scale), config=-1,
                     ksplit=int(ksplit), workspace=None, workspace_bytes=0, groups=G, group_in_stride=Cin if G > 1 else 0, stats_slots=0,
                     accum_half=None, out_scale_dev=L.dptr(out_scale_dev, "out_scale_dev"),
                     rgb_w=L.dptr(rgb_w2d, "rgb weight") if rgb is not None else None,
                     rgb_bias=L.dptr(rgb[1], "rgb bias") if rgb is not None and rgb[1] is not None else None,
                     rgb_y=L.dptr(rgb_out, "rgb_out") if rgb is not None else None, rgb_channels=3 if rgb is not None else 0)
    ws_bytes = L.lib().spk_conv2d_wino_workspace_bytes(int(ksplit), B, Cin, G * Cout, H, W)
    if ws_bytes > 0:                        # few regions: the contraction runs in slices, partial sums through the split-K workspace
        ws = _workspace(x.device, ws_bytes)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    _launch_conv2d(d)
    return (out, rgb_out) if rgb is not None else out


# 3x3 stride-1 spellings used by the decoder
def conv3x3_pick_config(B, Cin, Cout, H, W) -> int:
    return conv2d_pick_config(3, 1, B, Cin, Cout, H, W)


def conv3x3_config_fits(config, B, Cin, Cout, H, W) -> bool:
    return conv2d_config_fits(config, 3, 1, B, Cin, Cout, H, W)


conv3x3_config_info = conv2d_config_info
pack_conv3x3_weight = pack_conv_weight


def conv3x3_fused(x, w_packed, Cout, **kw):
    return conv2d_fused(x, w_packed, Cout, 3, 1, **kw)


def _style_ptr(style):
    """Style rows may be a strided view (a column block of the batched style buffer)."""
    if style is None:
        return None
    if not style.is_cuda or style.dtype != torch.float32 or style.stride(-1) != 1:
        raise L.SpkError("style: expected a float32 HIP tensor with unit inner stride")
    return style.data_ptr()


def fc(x, weight, bias=None, wmul=1.0, bmul=1.0, slope=1.0, out=None):
    """out[b,o] = act(wmul * <x[b], weight[o]> + bmul*bias[o]); x may be a row-strided 2-D view."""
    if x.dim() != 2 or x.stride(1) != 1:
        raise L.SpkError("fc: x must be 2-D with unit inner stride")
    B, I = x.shape
    O = weight.shape[0]
    if weight.shape[1] != I:
        raise L.SpkError(f"fc: weight {tuple(weight.shape)} does not match input width {I}")
    if out is None:
        out = torch.empty((B, O), device=x.device, dtype=torch.float32)
    if out.stride(1) != 1:
        raise L.SpkError("fc: out must have unit inner stride")
    if not x.is_cuda or x.dtype != torch.float32:
        raise L.SpkError(f"fc: expected a float32 HIP tensor, got {x.dtype} on {x.device} (no CPU path)")
    L.check(L.lib().spk_fc_fwd(x.data_ptr(), x.stride(0), L.dptr(weight, "weight"), L.dptr(bias, "bias"),
                               out.data_ptr(), out.stride(0), B, I, O, float(wmul), float(bmul), float(slope),
                               L.stream_ptr()), "spk_fc_fwd")
    return out


def fc_grouped(items):
    """One launch for up to 16 independent FCs.  ``items``: iterable of (x [B,I] row-strided view, weight [O,I], bias | None,
    wmul, bmul, slope); returns the list of outputs [B,O].  All x share the batch size."""
    items = list(items)
    outs, groups = [], (L.FcGroup * len(items))()
    B = items[0][0].shape[0]
    for g, (x, weight, bias, wmul, bmul, slope) in zip(groups, items):
        if x.dim() != 2 or x.stride(1) != 1 or x.shape[0] != B or not x.is_cuda or x.dtype != torch.float32:
            raise L.SpkError("fc_grouped: every x must be a float32 HIP [B,I] view with unit inner stride")
        O, I = weight.shape
        out = torch.empty((B, O), device=x.device, dtype=torch.float32)
        outs.append(out)
        g.x, g.x_stride, g.w, g.bias = x.data_ptr(), x.stride(0), L.dptr(weight, "weight"), L.dptr(bias, "bias")
        g.out, g.out_stride, g.I, g.O = out.data_ptr(), O, I, O
        g.wmul, g.bmul, g.slope = float(wmul), float(bmul), float(slope)
    L.check(L.lib().spk_fc_grouped_fwd(C.cast(groups, C.c_void_p), len(items), B, L.stream_ptr()), "spk_fc_grouped_fwd")
    return outs


def fc_grouped_bwd(items, B):
    """Backward of up to 16 independent FCs in two launches.  ``items``: iterable of (dout [B,O], out [B,O] saved output,
    x [B,I] row-strided view, weight [O,I], dx [B,I] row-strided view | None, need_dw, has_bias, wmul, bmul, slope);
    returns the list of (dw | None, db | None)."""
    items = list(items)
    groups, res, keep = (L.FcBwdGroup * len(items))(), [], []
    for g, (dout, out, x, weight, dx, need_dw, has_bias, wmul, bmul, slope) in zip(groups, items):
        O, I = weight.shape
        if dout.dim() != 2 or dout.stride(1) != 1 or dout.stride(0) < O:      # a row-strided view is read in place
            dout = dout.contiguous()
        keep.append(dout)
        if x.stride(1) != 1 or (dx is not None and dx.stride(1) != 1):
            raise L.SpkError("fc_grouped_bwd: x / dx must have unit inner stride")
        dw = torch.empty((O, I), device=weight.device, dtype=torch.float32) if need_dw else None
        db = torch.empty(O, device=weight.device, dtype=torch.float32) if (need_dw and has_bias) else None
        res.append((dw, db))
        g.dout, g.dout_stride, g.out, g.x, g.x_stride = dout.data_ptr(), dout.stride(0), L.dptr(out, "out"), x.data_ptr(), x.stride(0)
        g.w, g.dx, g.dx_stride = L.dptr(weight, "weight"), (dx.data_ptr() if dx is not None else None), (dx.stride(0) if dx is not None else 0)
        g.dw, g.db, g.I, g.O = L.dptr(dw), L.dptr(db), I, O
        g.wmul, g.bmul, g.slope = float(wmul), float(bmul), float(slope)
    L.check(L.lib().spk_fc_grouped_bwd(C.cast(groups, C.c_void_p), len(items), B, L.stream_ptr()), "spk_fc_grouped_bwd")
    return res


def bias_noise_style(x, B: int, bias=None, noise_w=None, noise=None, style=None):
    """y = (x + bias + noise_w*noise) * (s0+1) + s1 -> [B,C,H,W]; x is [B,C,H,W] or a [1,C,H,W]
    constant broadcast over the batch.  Any of bias / noise / style may be None."""
    xb, Cc, H, W = x.shape
    if xb not in (1, B):
        raise L.SpkError(f"bias_noise_style: batch {xb} does not broadcast to {B}")
    if noise is not None and noise.numel() != B * H * W:
        raise L.SpkError(f"bias_noise_style: noise must be [B,1,H,W], got {tuple(noise.shape)}")
    out = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_bias_noise_style_fwd(L.dptr(x, "x"), 0 if xb == 1 and B > 1 else Cc * H * W,
                                             L.dptr(bias, "bias"),
                                             L.dptr(noise_w, "noise_w") if noise is not None else None,
                                             L.dptr(noise, "noise"), _style_ptr(style),
                                             style.stride(0) if style is not None else 0, L.dptr(out), B, Cc, H * W,
                                             L.stream_ptr()), "spk_bias_noise_style_fwd")
    return out


def const_prologue(const_in, bias, noise_w, noise, style, B: int):
    """styleganv1.py:596-599 in one launch."""
    return bias_noise_style(const_in, B, bias, noise_w, noise, style)


def conv1x1_small(x, weight, bias=None, in_scale=1.0):
    """1x1 conv to <= 4 channels (toRGB)."""
    B, Cc, H, W = x.shape
    O = weight.shape[0]
    out = torch.empty((B, O, H, W), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_conv1x1_small_fwd(L.dptr(x, "x"), L.dptr(weight.reshape(O, Cc), "weight"), L.dptr(bias, "bias"),
                                          L.dptr(out), B, Cc, O, H * W, float(in_scale), L.stream_ptr()),
            "spk_conv1x1_small_fwd")
    return out


def conv1x1_expand(x, weight, bias=None, scale_dev=None, slope=None):
    """1x1 conv from <= 4 channels (fromRGB) + bias + LeakyReLU as one store stream (csrc/pointwise.hip conv1x1_expand_kernel)."""
    B, Cc, H, W = x.shape
    O = weight.shape[0]
    out = torch.empty((B, O, H, W), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_conv1x1_expand_fwd(L.dptr(x, "x"), L.dptr(weight.reshape(O, Cc), "weight"), L.dptr(bias, "bias"),
                                           L.dptr(scale_dev, "scale_dev"), L.dptr(out), B, Cc, O, H * W,
                                           float(slope if slope is not None else 1.0), L.stream_ptr()), "spk_conv1x1_expand_fwd")
    return out


def conv1x1_expand_ok(x, Cin, k, stride) -> bool:
    return k == 1 and stride == 1 and Cin <= 4 and (x.shape[-1] * x.shape[-2]) % 4 == 0 and x.data_ptr() % 16 == 0 and x.shape[0] < 65536


def upsample2x_bilinear(x):
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_upsample2x_bilinear_fwd(L.dptr(x, "x"), L.dptr(out), B * Cc, H, W, L.stream_ptr()),
            "spk_upsample2x_bilinear_fwd")
    return out


# ---- backward launchers ------------------------------------------------------------------------------
def wgrad_mod_supported(B, Cin, Cout, H, W, upsample) -> bool:
    """Whether ``conv2d_wgrad(..., batch_scale=, g_scale=)`` runs fused for this OUTPUT shape (H, W)."""
    return bool(L.lib().spk_conv2d_wgrad_mod_supported(B, Cin, Cout, H, W, 1 if upsample else 0))


def conv2d_wgrad(g, x, Cout, Cin, k=3, stride=1, *, upsample=False, in_affine=None, scale=1.0, out=None,
                 accumulate=False, splits=0, groups=1, shared_input=False, fold=1, batch_scale=None, g_scale=None, up_fir=False):
    """dW[Cout,Cin,k,k] = scale * sum_{b,h,w} g[b,co,h,w] * in(x)[b,ci,h*s+ky-p,w*s+kx-p].
    ``batch_scale`` [B,Cin] + ``g_scale`` [B,Cout]: the modulated convolution (StyleGAN2 variant) -- in(x) = x * batch_scale
    and g is multiplied by g_scale, both while staging (no rescaled tensor); with ``upsample`` (and ``up_fir``) x is the
    low-resolution tensor and the x2 image is upfirdn2d(up=2, [1,3,3,1]).  Shapes: ``wgrad_mod_supported``.
    ``groups`` > 1: Cout / Cin per group, g has groups*Cout channels, the result is [groups*Cout, Cin, k, k].
    ``fold`` > 1: groups q and q + groups/fold share their weights (the same conv on another image set): their
    gradients are summed in the slab reduce and the result is [groups/fold*Cout, Cin, k, k]."""
    B, _, H, W = g.shape
    if batch_scale is not None:
        if g_scale is None or in_affine is not None or int(groups) > 1 or (upsample and not up_fir):
            raise L.SpkError("conv2d_wgrad: batch_scale goes with g_scale, ungrouped, no in_affine, and up_fir when upsampling")
        if tuple(batch_scale.shape) != (B, Cin) or tuple(g_scale.shape) != (B, Cout):
            raise L.SpkError("conv2d_wgrad: batch_scale must be [B,Cin] and g_scale [B,Cout]")
    elif upsample and (int(groups) > 1 or not L.lib().spk_conv2d_wgrad_up_supported(B, Cin, int(groups) * Cout, H, W)
                       or x.data_ptr() % 16 or g.data_ptr() % 16):
        # small or odd planes: materialise the x2 image once and run the plain kernel.  Everything else (W % 8 == 0, at least
        # 16 x 4) interpolates the plane LDS -> LDS from a low-resolution source patch inside the kernel: no x2 tensor in HBM.
        x, upsample = upsample2x_bilinear(x), False
    Hs, Ws = x.shape[-2:]
    G = int(groups)
    fold = int(fold)
    if fold < 1 or G % fold:
        raise L.SpkError(f"conv2d_wgrad: fold {fold} must divide groups {G}")
    if (k == 3 and stride == 1 and int(splits) == 0 and Cout % 64 == 0 and (G == 1 or (batch_scale is None and not upsample))
            and use_wgrad_wino(B, Cin, G * Cout, H, W) and x.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0):
        # fp32 Winograd (ops.CONV3X3_ALGO): 16/36 of the multiply-adds; a x2 layer reads the materialised x2 image
        if upsample:
            x = upsample2x(x, zero_border=True) if up_fir else upsample2x_bilinear(x)
        return conv2d_wgrad_wino(g, x, Cout, Cin, scale=scale, out=out, accumulate=accumulate, batch_scale=batch_scale, g_scale=g_scale,
                                 in_affine=in_affine, groups=G, shared_input=shared_input, fold=fold)
    if out is None:
        out = torch.empty((G // fold * Cout, Cin, k, k), device=g.device, dtype=torch.float32)
    ws_bytes = L.lib().spk_conv2d_wgrad_workspace_bytes(k, k, stride, int(splits), B, Cin, G * Cout, H, W)
    if ws_bytes < 0:
        raise L.SpkError("conv2d_wgrad: unsupported problem")
    ws = _workspace(g.device, ws_bytes)
    flags = (L.CONV_UPSAMPLE2X if upsample else 0) | (L.CONV_IN_AFFINE_RELU if in_affine is not None else 0) | \
        (L.CONV_IN_BATCH_SCALE if batch_scale is not None else 0) | (L.CONV_UP_FIR1331 if (upsample and up_fir) else 0)
    d = L.WgradDesc(g=L.dptr(g, "g"), x=L.dptr(x, "x"),
                    in_scale=(L.dptr(in_affine[0], "in_scale") if in_affine is not None else L.dptr(batch_scale, "batch_scale")),
                    in_shift=L.dptr(in_affine[1], "in_shift") if in_affine is not None else None,
                    dw=L.dptr(out, "dw"), B=B, Cin=Cin, Cout=Cout, H=H, W=W, Hin=Hs, Win=Ws, kh=k, kw=k, stride=stride,
                    flags=flags, scale=float(scale), accumulate=1 if accumulate else 0, splits=int(splits),
                    workspace=ws.data_ptr(), workspace_bytes=ws.numel() * 4, groups=G,
                    group_in_stride=0 if (shared_input or G == 1) else Cin, fold=fold, g_scale=L.dptr(g_scale, "g_scale"))
    L.check(L.lib().spk_conv2d_wgrad(C.byref(d), L.stream_ptr()), "spk_conv2d_wgrad")
    return out


def wgrad_wino_supported(B, Cin, Cout, H, W) -> bool:
    return bool(L.lib().spk_conv2d_wgrad_wino_supported(B, Cin, Cout, H, W))


def use_wgrad_wino(B, Cin, Cout, H, W) -> bool:
    """Whether a plain 3x3 stride-1 weight gradient with OUTPUT size H x W goes to the Winograd kernel: the switch is on, the shape is
    served, and there are enough 16 x 2 pixel chunks for every workgroup to amortise its in-register G^T dU G epilogue."""
    if CONV3X3_ALGO == "direct" or not wgrad_wino_supported(B, Cin, Cout, H, W):
        return False
    return B * (H // 2) * (W // 16) >= WGRAD_WINO_MIN_CHUNKS * max(1, 256 // ((Cin // 64) * (Cout // 64)))


WGRAD_WINO_MIN_CHUNKS = 8           # per workgroup of the one-round grid (256 workgroups)


def conv2d_wgrad_wino(g, x, Cout, Cin, *, scale=1.0, out=None, accumulate=False, splits=0, batch_scale=None, g_scale=None,
                      in_affine=None, groups=1, shared_input=False, fold=1):
    """``conv2d_wgrad`` of a 3x3 stride-1 pad-1 conv as Winograd F(2x2, 3x3) (csrc/wgrad3x3_wino_f32.hip); x is the conv's
    actual input (a x2 layer passes the materialised x2 image).  ``batch_scale`` [B,Cin] + ``g_scale`` [B,Cout]: the modulated
    convolution -- x * batch_scale and g * g_scale are formed in registers on the way into the transforms.  ``in_affine`` =
    (scale, shift) per input channel: the conv's input was relu(x * scale + shift) (a folded BatchNorm).  ``groups`` / ``fold`` /
    ``shared_input`` as in ``conv2d_wgrad``."""
    B, _, H, W = g.shape
    G, fold = int(groups), int(fold)
    if fold < 1 or G % fold:
        raise L.SpkError(f"conv2d_wgrad_wino: fold {fold} must divide groups {G}")
    if (batch_scale is None) != (g_scale is None):
        raise L.SpkError("conv2d_wgrad_wino: batch_scale goes with g_scale")
    if batch_scale is not None and (tuple(batch_scale.shape) != (B, Cin) or tuple(g_scale.shape) != (B, Cout) or G > 1 or in_affine is not None):
        raise L.SpkError("conv2d_wgrad_wino: batch_scale must be [B,Cin] and g_scale [B,Cout], ungrouped, without in_affine")
    gin = 0 if (shared_input or G == 1) else Cin
    Cx = gin * (G - 1) + Cin
    if tuple(x.shape) != (B, Cx, H, W) or g.shape[1] != G * Cout:
        raise L.SpkError(f"conv2d_wgrad_wino: g {tuple(g.shape)} / x {tuple(x.shape)} do not fit {G} groups of Cout {Cout}, Cin {Cin}")
    ws_bytes = L.lib().spk_conv2d_wgrad_wino_workspace_bytes(int(splits), B, Cin, G * Cout, H, W) if Cout % 64 == 0 else -1
    if ws_bytes < 0:
        raise L.SpkError("conv2d_wgrad_wino: shape not served (Cin, Cout multiples of 64, H even, W a multiple of 16)")
    if out is None:
        out = torch.empty((G // fold * Cout, Cin, 3, 3), device=g.device, dtype=torch.float32)
    ws = _workspace(g.device, ws_bytes)
    flags = L.CONV_WINOGRAD | (L.CONV_IN_BATCH_SCALE if batch_scale is not None else 0) | (L.CONV_IN_AFFINE_RELU if in_affine is not None else 0)
    d = L.WgradDesc(g=L.dptr(g, "g"), x=L.dptr(x, "x"),
                    in_scale=(L.dptr(in_affine[0], "in_scale") if in_affine is not None else L.dptr(batch_scale, "batch_scale")),
                    in_shift=L.dptr(in_affine[1], "in_shift") if in_affine is not None else None, dw=L.dptr(out, "dw"),
                    B=B, Cin=Cin, Cout=Cout, H=H, W=W, Hin=H, Win=W, kh=3, kw=3, stride=1, flags=flags, scale=float(scale),
                    accumulate=1 if accumulate else 0, splits=int(splits), workspace=ws.data_ptr(), workspace_bytes=ws.numel() * 4,
                    groups=G, group_in_stride=gin, fold=fold, g_scale=L.dptr(g_scale, "g_scale"))
    L.check(L.lib().spk_conv2d_wgrad(C.byref(d), L.stream_ptr()), "spk_conv2d_wgrad")
    return out


def epilogue_bwd(dy, a=None, noise=None, style=None, slope=1.0, inplace=False):
    """Adjoint of the fused conv epilogue: returns (dt, sums[B,4,C]) with
    sums[:, k] = {sum dy*a, sum dy, sum dt, sum dt*noise}[k] per (b,c) plane -- ``sums[:, :2].reshape(B, 2C)`` is the style
    gradient [d s0 | d s1] (a view), ``sums[:, 2:].sum(0)`` the bias and noise-weight gradients."""
    B, Cc, H, W = dy.shape
    dt = dy if inplace else torch.empty_like(dy)
    sums = torch.empty((B, 4, Cc), device=dy.device, dtype=torch.float32)
    L.check(L.lib().spk_epilogue_bwd(L.dptr(dy, "dy"), L.dptr(a, "a"), L.dptr(noise, "noise"), _style_ptr(style),
                                     style.stride(0) if style is not None else 0, float(slope), L.dptr(dt), L.dptr(sums),
                                     B, Cc, H * W, L.stream_ptr()), "spk_epilogue_bwd")
    return dt, sums


def upsample2x_bilinear_bwd(dy):
    B, Cc, H2, W2 = dy.shape
    dx = torch.empty((B, Cc, H2 // 2, W2 // 2), device=dy.device, dtype=torch.float32)
    L.check(L.lib().spk_upsample2x_bilinear_bwd(L.dptr(dy, "dy"), L.dptr(dx), B * Cc, H2 // 2, W2 // 2, L.stream_ptr()),
            "spk_upsample2x_bilinear_bwd")
    return dx


def conv1x1_small_bwd(x, weight, dy, need_dx=True, in_scale=1.0):
    """toRGB backward -> (dx | None, dw[O,C,1,1], db[O])."""
    B, Cc, H, W = x.shape
    O = weight.shape[0]
    nblk = L.lib().spk_conv1x1_small_bwd_blocks(B, H * W)
    partial = torch.empty((nblk, O * Cc + O), device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x) if need_dx else None
    L.check(L.lib().spk_conv1x1_small_bwd(L.dptr(x, "x"), L.dptr(weight.reshape(O, Cc), "weight"), L.dptr(dy, "dy"),
                                          L.dptr(dx), L.dptr(partial), B, Cc, O, H * W, float(in_scale), L.stream_ptr()),
            "spk_conv1x1_small_bwd")
    tot = partial.sum(0)                      # [O*C + O]: a few hundred numbers
    return dx, tot[:O * Cc].view(O, Cc, 1, 1), tot[O * Cc:]


def fc_bwd(dout, out, x, weight, wmul=1.0, bmul=1.0, slope=1.0, need_dx=True, need_dw=True, has_bias=True):
    """FC backward -> (dx | None, dw | None, db | None); ``out`` is the saved forward output."""
    B, I = x.shape
    O = weight.shape[0]
    dev = x.device
    dx = torch.empty((B, I), device=dev, dtype=torch.float32) if need_dx else None
    dw = torch.empty((O, I), device=dev, dtype=torch.float32) if need_dw else None
    db = torch.empty(O, device=dev, dtype=torch.float32) if (need_dw and has_bias) else None
    L.check(L.lib().spk_fc_bwd(L.dptr(dout, "dout"), L.dptr(out, "out"), x.data_ptr(), x.stride(0), L.dptr(weight, "weight"),
                               L.dptr(dx), I, L.dptr(dw), L.dptr(db), B, I, O, float(wmul), float(bmul), float(slope),
                               L.stream_ptr()), "spk_fc_bwd")
    return dx, dw, db


# ---- spectral normalisation, all layers of a module in one call ---------------------------------------------
def _sn_groups(items):
    arr = (L.SnGroup * len(items))()
    for q, (w, u, v, w_hat, sigma, dw) in zip(arr, items):
        R = w.shape[0]
        q.w, q.u, q.v = L.dptr(w, "weight_orig"), L.dptr(u, "weight_u"), L.dptr(v, "weight_v")
        q.w_hat, q.sigma, q.dw = L.dptr(w_hat, "w_hat"), sigma.data_ptr(), L.dptr(dw, "dw")
        q.R, q.C = R, w.numel() // R
        if u.numel() != q.R or v.numel() != q.C:
            raise L.SpkError(f"spectral_norm: u / v sizes {u.numel()} / {v.numel()} do not match the [{q.R},{q.C}] matrix")
    return arr


def spectral_norm_grouped(weights, us, vs, power_iteration, eps=1e-12):
    """-> (list of W / sigma, sigma [n]); ``us`` / ``vs`` are updated in place when ``power_iteration``."""
    n = len(weights)
    if n > L.SN_MAX_GROUPS:
        raise L.SpkError(f"spectral_norm_grouped: at most {L.SN_MAX_GROUPS} layers per call")
    dev = weights[0].device
    sigma = torch.empty(n, device=dev, dtype=torch.float32)
    hats = [torch.empty_like(w, memory_format=torch.contiguous_format) for w in weights]
    arr = _sn_groups([(w, u, v, h, sigma[i:i + 1], None) for i, (w, u, v, h) in enumerate(zip(weights, us, vs, hats))])
    nbytes = L.lib().spk_spectral_norm_workspace_bytes(C.cast(arr, C.c_void_p), n)
    ws = _workspace(dev, nbytes)
    L.check(L.lib().spk_spectral_norm_grouped(C.cast(arr, C.c_void_p), n, 1 if power_iteration else 0, float(eps), ws.data_ptr(),
                                              ws.numel() * 4, L.stream_ptr()), "spk_spectral_norm_grouped")
    return hats, sigma


def spectral_norm_grouped_bwd(grads, weights, us, vs, sigma, into=None):
    """dW = (G - <G, W>/sigma u v^T) / sigma for every layer with a gradient (``grads[i]`` None -> None).  ``into[i]`` (a tensor or
    None): ADD layer i's result into that tensor instead of returning a new one (the entry of the result is then None)."""
    idx = [i for i, g in enumerate(grads) if g is not None]
    out = [None] * len(grads)
    if not idx:
        return out
    acc = {i: (into[i] if into is not None else None) for i in idx}
    dws = {i: (acc[i] if acc[i] is not None else torch.empty_like(weights[i], memory_format=torch.contiguous_format)) for i in idx}
    arr = _sn_groups([(weights[i], us[i], vs[i], grads[i], sigma[i:i + 1], dws[i]) for i in idx])
    for g, i in zip(arr, idx):
        g.accumulate = 1 if acc[i] is not None else 0
    nbytes = L.lib().spk_spectral_norm_workspace_bytes(C.cast(arr, C.c_void_p), len(idx))
    ws = _workspace(weights[0].device, nbytes)
    L.check(L.lib().spk_spectral_norm_bwd_grouped(C.cast(arr, C.c_void_p), len(idx), ws.data_ptr(), ws.numel() * 4, L.stream_ptr()),
            "spk_spectral_norm_bwd_grouped")
    for i in idx:
        out[i] = dws[i] if acc[i] is None else None
    return out


def plane_sums_reduce(sums, row, out=None):
    """out[c] (+)= sum_b sums[b, row, c]  (``out`` given: accumulate into it; else a new tensor)."""
    B, rows, Cc = sums.shape
    acc = out is not None
    if out is None:
        out = torch.empty(Cc, device=sums.device, dtype=torch.float32)
    L.check(L.lib().spk_plane_sums_reduce(L.dptr(sums, "sums"), B, rows, Cc, int(row), L.dptr(out), 1 if acc else 0, L.stream_ptr()),
            "spk_plane_sums_reduce")
    return out


# ---- StyleGAN2 pieces (build-defined variant) ---------------------------------------------------------------
def modconv_demod(weight, s, scale, eps=1e-8):
    """d[b,co] = rsqrt(scale^2 * sum_{ci,k} (w[co,ci,k]*s[b,ci])^2 + eps)."""
    Cout, Cin, kh, kw = weight.shape
    B = s.shape[0]
    d = torch.empty((B, Cout), device=s.device, dtype=torch.float32)
    L.check(L.lib().spk_modconv_demod(L.dptr(weight, "weight"), L.dptr(s, "s"), L.dptr(d), B, Cin, Cout, kh * kw, float(scale),
                                      float(eps), L.stream_ptr()), "spk_modconv_demod")
    return d


def modconv_demod_grouped(items, eps=1e-8):
    """One launch for up to 16 layers' demodulation vectors.  ``items``: (weight [Cout,Cin,k,k], s [B,Cin], scale)."""
    items = list(items)
    groups, outs = (L.DemodGroup * len(items))(), []
    B = items[0][1].shape[0]
    for q, (weight, s, scale) in zip(groups, items):
        Cout, Cin, kh, kw = weight.shape
        d = torch.empty((B, Cout), device=s.device, dtype=torch.float32)
        outs.append(d)
        q.w, q.s, q.d = L.dptr(weight, "weight"), L.dptr(s, "s"), d.data_ptr()
        q.Cin, q.Cout, q.taps, q.scale = Cin, Cout, kh * kw, float(scale)
    L.check(L.lib().spk_modconv_demod_grouped(C.cast(groups, C.c_void_p), len(items), B, float(eps), L.stream_ptr()),
            "spk_modconv_demod_grouped")
    return outs


def modconv_epi_finish(sums, d, bias, noise_w, gain):
    """-> (dd [B,C] | None, dprime [B,C], dbias [C] | None, dnw [C] | None) from the epilogue adjoint's plane sums."""
    B, _, Cc = sums.shape
    dev = sums.device
    dd = torch.empty((B, Cc), device=dev, dtype=torch.float32) if d is not None else None
    dprime = torch.empty((B, Cc), device=dev, dtype=torch.float32)
    dbias = torch.empty(Cc, device=dev, dtype=torch.float32) if bias is not None else None
    dnw = torch.empty(Cc, device=dev, dtype=torch.float32) if noise_w is not None else None
    L.check(L.lib().spk_modconv_epi_finish(L.dptr(sums, "sums"), L.dptr(d, "d"), L.dptr(bias, "bias"), L.dptr(noise_w, "noise_w"),
                                           float(gain), L.dptr(dd), L.dptr(dprime), L.dptr(dbias), L.dptr(dnw), B, Cc, L.stream_ptr()),
            "spk_modconv_epi_finish")
    return dd, dprime, dbias, dnw


def modconv_dx_finish(dxt, x, s, upsample, need_dx=True):
    """The tail of the modulated conv's data path: (dx = s * up^T(dxt) | None, ds[b,ci] = <up^T(dxt), x>) -- the adjoint of
    upfirdn2d(up=2, [1,3,3,1]) when ``upsample`` (dxt is then at twice x's resolution), identity otherwise."""
    B, Cc, Hs, Ws = x.shape
    if tuple(dxt.shape) != ((B, Cc, 2 * Hs, 2 * Ws) if upsample else (B, Cc, Hs, Ws)):
        raise L.SpkError(f"modconv_dx_finish: gradient {tuple(dxt.shape)} does not match input {tuple(x.shape)} (upsample={upsample})")
    ds = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    dx = (torch.empty_like(x) if upsample else dxt) if need_dx else None      # same resolution: scaled in place
    L.check(L.lib().spk_modconv_dx_finish(L.dptr(dxt, "dxt"), L.dptr(x, "x"), L.dptr(s, "s"), L.dptr(dx), L.dptr(ds), B, Cc, Hs, Ws,
                                          1 if upsample else 0, L.stream_ptr()), "spk_modconv_dx_finish")
    return dx, ds


def modconv_demod_bwd(weight, s, d, dd, scale, ds=None, dw=None):
    """Adjoint of ``modconv_demod``: accumulates into ``ds`` [B,Cin] and / or ``dw`` [Cout,Cin,k,k] (in place)."""
    Cout, Cin, kh, kw = weight.shape
    B = s.shape[0]
    ws = _workspace(s.device, L.lib().spk_modconv_demod_bwd_workspace_bytes(B, Cin, Cout)) if ds is not None else None
    L.check(L.lib().spk_modconv_demod_bwd(L.dptr(weight, "weight"), L.dptr(s, "s"), L.dptr(d, "d"), L.dptr(dd, "dd"), L.dptr(ds),
                                          L.dptr(dw), ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0,
                                          B, Cin, Cout, kh * kw, float(scale), L.stream_ptr()), "spk_modconv_demod_bwd")


def torgb_mod_bwd(x, weight, mod, dy, in_scale=1.0, need_dx=True):
    """Backward of the modulated toRGB (``conv1x1_small_mod``) -> (dx | None, P [B,O,C] = in_scale * sum_p dy[b,o,p] x[b,c,p],
    db [O]): the weight gradient is sum_b mod[b,c] P[b,o,c], the modulation gradient sum_o w[o,c] P[b,o,c]."""
    B, Cc, H, W = x.shape
    O = weight.shape[0]
    w2 = weight.reshape(O, Cc)
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        L.check(L.lib().spk_torgb_mod_bwd_data(L.dptr(w2, "weight"), L.dptr(mod, "mod"), L.dptr(dy, "dy"), L.dptr(dx), B, Cc, O, H * W,
                                               float(in_scale), L.stream_ptr()), "spk_torgb_mod_bwd_data")
    nblk = L.lib().spk_conv1x1_small_bwd_blocks(B, H * W)
    partial = torch.empty((nblk, O * Cc + O), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_conv1x1_small_bwd(L.dptr(x, "x"), L.dptr(w2, "weight"), L.dptr(dy, "dy"), None, L.dptr(partial), B, Cc, O,
                                          H * W, float(in_scale), L.stream_ptr()), "spk_conv1x1_small_bwd")
    per = partial.view(B, nblk // B, O * Cc + O).sum(1)            # blocks are laid out image by image
    return dx, per[:, :O * Cc].reshape(B, O, Cc), per[:, O * Cc:].sum(0)


def upfirdn2d(x, filt2d, up=1, down=1, pad=(0, 0), gain=1.0):
    """upfirdn2d of the StyleGAN2 reference implementation (zero-insert, pad/crop, flipped-FIR, decimate)."""
    B, Cc, H, W = x.shape
    f = torch.as_tensor(filt2d, dtype=torch.float32).contiguous().cpu()
    k = f.shape[0]
    Ho, Wo = (H * up + pad[0] + pad[1] - k) // down + 1, (W * up + pad[0] + pad[1] - k) // down + 1
    y = torch.empty((B, Cc, Ho, Wo), device=x.device, dtype=torch.float32)
    arr = (C.c_float * (k * k))(*f.flatten().tolist())
    L.check(L.lib().spk_upfirdn2d_fwd(L.dptr(x, "x"), L.dptr(y), arr, k, B * Cc, H, W, int(up), int(down), int(pad[0]), int(pad[1]),
                                      float(gain), L.stream_ptr()), "spk_upfirdn2d_fwd")
    return y


def conv1x1_small_mod(x, weight, mod, bias=None, in_scale=1.0, skip=None):
    """Modulated (not demodulated) 1x1 conv to <= 4 channels: the StyleGAN2 toRGB.  ``skip`` [B,O,H/2,W/2]: the previous
    resolution's image, upsampled (upfirdn2d up=2, [1,3,3,1]) and added in the same launch."""
    B, Cc, H, W = x.shape
    O = weight.shape[0]
    y = torch.empty((B, O, H, W), device=x.device, dtype=torch.float32)
    if skip is not None:
        if tuple(skip.shape) != (B, O, H // 2, W // 2) or H % 2 or W % 2:
            raise L.SpkError(f"conv1x1_small_mod: skip {tuple(skip.shape)} is not [B,O,H/2,W/2] of {(B, O, H, W)}")
        L.check(L.lib().spk_torgb_mod_skip_fwd(L.dptr(x, "x"), L.dptr(weight.reshape(O, Cc), "weight"), L.dptr(mod, "mod"),
                                               L.dptr(bias, "bias"), L.dptr(skip, "skip"), L.dptr(y), B, Cc, O, H, W, float(in_scale),
                                               L.stream_ptr()), "spk_torgb_mod_skip_fwd")
        return y
    L.check(L.lib().spk_conv1x1_small_mod_fwd(L.dptr(x, "x"), L.dptr(weight.reshape(O, Cc), "weight"), L.dptr(mod, "mod"),
                                              L.dptr(bias, "bias"), L.dptr(y), B, Cc, O, H * W, float(in_scale), L.stream_ptr()),
            "spk_conv1x1_small_mod_fwd")
    return y


# ---- stand-alone StyleGAN1 / ProGAN ops ------------------------------------------------------------------
def pixelnorm(x, eps=1e-8, sqrt_form=False):
    """x * rsqrt(mean over dim 1 of x^2 + eps); x is [B,C] or [B,C,H,W]."""
    B, Cc = x.shape[:2]
    HW = x.numel() // (B * Cc)
    y = torch.empty_like(x)
    L.check(L.lib().spk_pixelnorm_fwd(L.dptr(x, "x"), L.dptr(y), B, Cc, HW, float(eps), 1 if sqrt_form else 0, L.stream_ptr()),
            "spk_pixelnorm_fwd")
    return y


def instance_norm_affine(x, scale=None, bias=None, eps=1e-5):
    """Per-(b,c)-plane normalisation then y*scale[b,c] + bias[b,c] (AdaIN); scale/bias are [B,C] (row-strided ok)."""
    B, Cc, H, W = x.shape
    y = torch.empty_like(x)
    stride = scale.stride(0) if scale is not None else (bias.stride(0) if bias is not None else 0)
    if scale is not None and bias is not None and scale.stride(0) != bias.stride(0):
        raise L.SpkError("instance_norm_affine: scale and bias must share their row stride")
    L.check(L.lib().spk_instance_norm_affine_fwd(L.dptr(x, "x"), L.dptr(y), _style_ptr(scale), _style_ptr(bias), stride, B, Cc,
                                                 H * W, float(eps), L.stream_ptr()), "spk_instance_norm_affine_fwd")
    return y


def instance_norm_affine_bwd(x, dy, scale=None, eps=1e-5, need_dx=True):
    """Adjoint of ``instance_norm_affine`` -> (dx | None, dscale [B,C], dbias [B,C])."""
    B, Cc, H, W = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dscale = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    dbias = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_instance_norm_affine_bwd(L.dptr(x, "x"), L.dptr(dy, "dy"), _style_ptr(scale),
                                                 scale.stride(0) if scale is not None else 0, L.dptr(dx), L.dptr(dscale),
                                                 L.dptr(dbias), B, Cc, H * W, float(eps), L.stream_ptr()),
            "spk_instance_norm_affine_bwd")
    return dx, dscale, dbias


def blur2d(x, filt2d, stride=1):
    """Depthwise FIR with the k x k filter ``filt2d`` (a CPU tensor / nested list), zero pad (k-1)/2."""
    B, Cc, H, W = x.shape
    f = torch.as_tensor(filt2d, dtype=torch.float32).contiguous().cpu()
    k = f.shape[0]
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((B, Cc, Ho, Wo), device=x.device, dtype=torch.float32)
    arr = (C.c_float * (k * k))(*f.flatten().tolist())
    L.check(L.lib().spk_blur2d_fwd(L.dptr(x, "x"), L.dptr(y), arr, k, B * Cc, H, W, int(stride), L.stream_ptr()), "spk_blur2d_fwd")
    return y


def upscale2d_nearest(x, factor=2, gain=1.0):
    B, Cc, H, W = x.shape
    y = torch.empty((B, Cc, H * factor, W * factor), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_upscale2d_nearest_fwd(L.dptr(x, "x"), L.dptr(y), B * Cc, H, W, int(factor), float(gain), L.stream_ptr()),
            "spk_upscale2d_nearest_fwd")
    return y


def pixelnorm_bwd(x, dy, eps=1e-8):
    B, Cc = x.shape[:2]
    HW = x.numel() // (B * Cc)
    dx = torch.empty_like(x)
    L.check(L.lib().spk_pixelnorm_bwd(L.dptr(x, "x"), L.dptr(dy, "dy"), L.dptr(dx), B, Cc, HW, float(eps), L.stream_ptr()),
            "spk_pixelnorm_bwd")
    return dx


def blur2d_bwd(dy, filt2d, stride, in_hw):
    """Adjoint of ``blur2d``: dy [B,C,Ho,Wo] -> dx [B,C,H,W] with (H, W) = ``in_hw``."""
    B, Cc = dy.shape[:2]
    H, W = in_hw
    f = torch.as_tensor(filt2d, dtype=torch.float32).contiguous().cpu()
    k = f.shape[0]
    pad = (k - 1) // 2
    if tuple(dy.shape[-2:]) != ((H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1):
        raise L.SpkError(f"blur2d_bwd: gradient {tuple(dy.shape)} is not the output of a {k}x{k} stride-{stride} blur of {H}x{W}")
    dx = torch.empty((B, Cc, H, W), device=dy.device, dtype=torch.float32)
    arr = (C.c_float * (k * k))(*f.flatten().tolist())
    L.check(L.lib().spk_blur2d_bwd(L.dptr(dy, "dy"), L.dptr(dx), arr, k, B * Cc, H, W, int(stride), L.stream_ptr()), "spk_blur2d_bwd")
    return dx


def upscale2d_nearest_bwd(dy, factor=2, gain=1.0):
    B, Cc, Ho, Wo = dy.shape
    if Ho % factor or Wo % factor:
        raise L.SpkError(f"upscale2d_nearest_bwd: {Ho}x{Wo} is not a multiple of the factor {factor}")
    dx = torch.empty((B, Cc, Ho // factor, Wo // factor), device=dy.device, dtype=torch.float32)
    L.check(L.lib().spk_upscale2d_nearest_bwd(L.dptr(dy, "dy"), L.dptr(dx), B * Cc, Ho // factor, Wo // factor, int(factor),
                                              float(gain), L.stream_ptr()), "spk_upscale2d_nearest_bwd")
    return dx


def conv_transpose4x4_s2(x, weight, bias=None, packed=None):
    """nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1) forward (styleganv1.py:231) as four output-parity 2x2 MFMA
    kernels in one launch.  ``weight`` is the module's [Cin,Cout,4,4] parameter; ``packed``: an optional PackedConvWeight."""
    B, Cin, H, W = x.shape
    if tuple(weight.shape[0:1]) != (Cin,) or tuple(weight.shape[2:]) != (4, 4):
        raise L.SpkError(f"conv_transpose4x4_s2: weight {tuple(weight.shape)} does not match [Cin={Cin},Cout,4,4]")
    Cout = weight.shape[1]
    cfg = conv2d_pick_config(2, 1, B, Cin, 4 * Cout, H + 1, W + 1)
    wp = packed.get(weight, cfg, 3) if packed is not None else pack_conv_weight(weight, cfg, 3)
    out = torch.empty((B, Cout, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    d = L.Conv2dDesc(x=L.dptr(x, "x"), w_packed=L.dptr(wp, "w_packed"), bias=L.dptr(bias, "bias"), y=L.dptr(out, "out"), B=B, Cin=Cin,
                     Cout=Cout, H=2 * H, W=2 * W, Hin=H, Win=W, kh=4, kw=4, stride=2,
                     flags=L.CONV_TRANSPOSE4X4_S2 | (L.EPI_BIAS if bias is not None else 0), lrelu_slope=1.0, out_scale=1.0,
                     config=int(cfg), ksplit=1, groups=1, group_in_stride=0)
    _launch_conv2d(d)
    return out


def fade_in_tanh(a, b, alpha):
    y = torch.empty_like(a)
    L.check(L.lib().spk_fade_in_tanh_fwd(L.dptr(a, "a"), L.dptr(b, "b"), L.dptr(y), float(alpha), a.numel(), L.stream_ptr()),
            "spk_fade_in_tanh_fwd")
    return y


MASK_NONE, MASK_RECOMPUTE, MASK_TENSOR = 0, 1, 2


def bn_backward(g, r, affine, mean, invstd, mask_mode, mask_src=None, g_scale=1.0, g_per_plane=False, want_dz=False,
                batch_stats=True):
    """Training-mode BatchNorm (+ReLU) backward in the folded form: returns (dr, dgamma, dbeta[, dz]).

    ``r`` raw conv output, ``affine`` = (scale, shift) of its BatchNorm, ``g`` the gradient w.r.t. what the
    forward consumer saw (see spk_bn_bwd_reduce in include/spk.h for ``mask_mode``).  ``g_per_plane``: g is
    [B,C] (one value per plane, e.g. the global-average-pool gradient)."""
    B, Cc, H, W = r.shape
    HW = H * W
    sums = torch.empty((B, 2, Cc), device=r.device, dtype=torch.float32)
    args = (L.dptr(g, "g"), L.dptr(r, "r"), L.dptr(mask_src, "mask_src"), int(mask_mode), L.dptr(affine[0], "scale"),
            L.dptr(affine[1], "shift"), L.dptr(mean, "mean"), L.dptr(invstd, "invstd"))
    L.check(L.lib().spk_bn_bwd_reduce(*args, float(g_scale), 1 if g_per_plane else 0, L.dptr(sums), B, Cc, HW,
                                      L.stream_ptr()), "spk_bn_bwd_reduce")
    # the reduce pass summed over pixels; the apply pass adds up the B per-image pairs of its channel itself and leaves the totals
    # in csum [2,C] (no reduction launch in between).  Eval-mode BatchNorm is a fixed affine: no mean / variance terms in dr.
    csum = torch.empty((2, Cc), device=r.device, dtype=torch.float32)
    dr = torch.empty_like(r)
    dz = torch.empty_like(r) if want_dz else None
    L.check(L.lib().spk_bn_bwd_apply_sums(*args, L.dptr(sums), L.dptr(csum), 1 if batch_stats else 0, B * HW, float(g_scale),
                                          1 if g_per_plane else 0, L.dptr(dr), L.dptr(dz), B, Cc, HW, L.stream_ptr()),
            "spk_bn_bwd_apply_sums")
    out = (dr, csum[1], csum[0])                    # rows of csum: d gamma, d beta (contiguous views)
    return out + (dz,) if want_dz else out


def dilate2x(x, Ho, Wo):
    B, Cc, H, W = x.shape
    y = torch.empty((B, Cc, Ho, Wo), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_dilate2x(L.dptr(x, "x"), L.dptr(y), B * Cc, H, W, Ho, Wo, L.stream_ptr()), "spk_dilate2x")
    return y


def maxpool3x3s2_bwd(x, dy, in_scale=None, in_shift=None):
    B, Cc, H, W = x.shape
    dx = torch.empty_like(x)
    L.check(L.lib().spk_maxpool3x3s2_bwd(L.dptr(x, "x"), L.dptr(in_scale, "in_scale"), L.dptr(in_shift, "in_shift"),
                                         L.dptr(dy, "dy"), L.dptr(dx), B, Cc, H, W, L.stream_ptr()), "spk_maxpool3x3s2_bwd")
    return dx


def dgrad_at_output_size(k, stride, out=None, accumulate=False):
    """Whether ``conv2d_dgrad`` contracts at the conv's OUTPUT size (then ``config`` must be picked for that size)."""
    return stride == 2 and (k == 3 or (k == 1 and out is None and not accumulate))


def dgrad_plan(k, stride, B, Cout, Cin, in_hw, g_hw, out=None, accumulate=False):
    """-> (tile config, ``transpose_flip`` mode of ``pack_conv_weight``) for ``conv2d_dgrad`` of this conv."""
    if k == 3 and stride == 2:       # by output parity over the gradient's own pixels: the exact-tap kernel (config 13)
        cfg = L.lib().spk_conv2d_dgrad_s2_config(B, Cout, Cin, g_hw[0], g_hw[1])
        if cfg < 0:
            raise L.SpkError(f"spk_conv2d_dgrad_s2_config: {L.lib().spk_last_error().decode()}")
        return cfg, 2
    hw = g_hw if dgrad_at_output_size(k, stride, out, accumulate) else in_hw
    return conv2d_pick_config(k, 1, B, Cout, Cin, hw[0], hw[1]), 1


def _dgrad_s2_parity(g, weight_packed, Cin, in_hw, config, out, accumulate, groups, out_scale_dev=None):
    B, Cg, Hg, Wg = g.shape
    G = int(groups)
    if Cg % G:
        raise L.SpkError(f"conv2d_dgrad: {Cg} gradient channels do not split into {G} groups")
    H, W = in_hw
    if out is None:
        out = torch.empty((B, G * Cin, H, W), device=g.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, G * Cin, H, W) or not out.is_contiguous():
        raise L.SpkError("conv2d_dgrad: out must be a contiguous [B, groups*Cin, H, W] tensor")
    d = L.Conv2dDesc(x=L.dptr(g, "g"), w_packed=L.dptr(weight_packed, "w_packed"), y=L.dptr(out, "out"), B=B, Cin=Cg // G,
                     Cout=Cin, H=H, W=W, Hin=Hg, Win=Wg, kh=3, kw=3, stride=2,
                     flags=L.CONV_DGRAD_S2 | (L.EPI_ACCUM if accumulate else 0), lrelu_slope=1.0, out_scale=1.0,
                     config=int(config), ksplit=0, groups=G, group_in_stride=0 if G == 1 else Cg // G,
                     out_scale_dev=L.dptr(out_scale_dev, "out_scale_dev"))
    ws_bytes = L.lib().spk_conv2d_dgrad_s2_workspace_bytes(B, Cg // G, Cin, Hg, Wg, H, W, G)
    if ws_bytes > 0:                # a small gradient plane: the exact-tap kernel runs its contraction in slices
        ws = _workspace(g.device, ws_bytes)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    _launch_conv2d(d)
    return out


GEMM2_CONFIGS = (14, 15)        # the lean GEMM form of a stride-1 1x1 (csrc/conv1x1_gemm2.hip): the configs that take accum_half


def conv2d_dgrad(g, weight_packed_tf, Cin, k, stride, in_hw, config, out=None, accumulate=False, groups=1, dilate=True,
                 accum_half=None, out_scale_dev=None):
    """Data gradient of a k x k conv.  Stride 1: the forward MFMA kernel on ``g`` with transpose-flipped weights.
    3x3 stride 2: by output parity (``SPK_CONV_DGRAD_S2``: dx[2m+py, 2n+px] needs 1/2/2/4 of the 9 taps; four 2x2
    kernels in one launch over the gradient's own pixels, stored interleaved).  1x1 stride 2 without a destination: at
    the output size, dilated afterwards; into a destination: on the zero-dilated ``g``.  ``in_hw`` = (H, W) of the
    conv's input; ``config`` / the packing of ``weight_packed_tf`` from ``dgrad_plan``.  ``dilate=False`` (strided 1x1 without a
    destination): the gradient stays at the output size; ``accum_half``: such a tensor, added at the even pixels by a stride-1
    1x1 data gradient running in the lean GEMM form (``GEMM2_CONFIGS``)."""
    if k == 3 and stride == 2:
        return _dgrad_s2_parity(g, weight_packed_tf, Cin, in_hw, config, out, accumulate, groups, out_scale_dev)
    if dgrad_at_output_size(k, stride, out, accumulate):
        # a strided 1x1 reads only the even input pixels: dx = dilate(W^T g), the contraction at the OUTPUT size
        t = conv2d_fused(g, weight_packed_tf, Cin, 1, 1, config=config, groups=groups, out_scale_dev=out_scale_dev)
        return dilate2x(t, in_hw[0], in_hw[1]) if dilate else t      # not dilated: the caller adds it through ``accum_half``
    if stride == 2:
        # dx[i] = sum_k gd[i + k' - p] * w[k-1-k'] with gd[2o] = g[o], zeros elsewhere, extended to the input size
        # (an even-sized input has a last row/column no window's stride lattice reaches: it stays zero)
        g = dilate2x(g, in_hw[0], in_hw[1])
    return conv2d_fused(g, weight_packed_tf, Cin, k, 1, config=config, out=out, accumulate=accumulate, groups=groups,
                        accum_half=accum_half, out_scale_dev=out_scale_dev)


# ---- BatchNorm / pooling pieces of the ResNet-50 trunk ----------------------------------------------
def stats_slots(config, k, stride, B, Cin, Cout, H, W):
    """Copies of the BatchNorm sums for a conv launch (``spk_conv2d_desc.stats_slots``): one per pixel tile, so that the
    epilogue stores its sums instead of queueing fp64 atomics (H, W = output size)."""
    n = L.lib().spk_conv2d_stats_slots(int(config), k, k, stride, B, Cin, Cout, H, W)
    if n < 1:
        raise L.SpkError(f"stats_slots: config {config} cannot host k={k} s={stride} shape {(B, Cin, Cout, H, W)}")
    # very large batches: bound the copies (2048 x 2C doubles); beyond that a few tiles share a copy through atomics
    return min(n, 2048)


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, save=False):
    """Batch sums (fp64 [slots][2C], from a conv epilogue) -> the per-channel affine (scale, shift) its consumer
    applies; updates running statistics when momentum > 0.  ``stats=None``: eval mode (running statistics).
    With ``save`` also returns (mean, invstd) for the backward pass."""
    Cc = gamma.numel()
    slots = 0
    if stats is not None:
        slots = stats.numel() // (2 * Cc)
        if slots < 1 or stats.numel() != slots * 2 * Cc:
            raise L.SpkError("bn_finalize: stats must hold slots*2*C sums")
    dev = gamma.device
    scale = torch.empty(Cc, device=dev, dtype=torch.float32)
    shift = torch.empty(Cc, device=dev, dtype=torch.float32)
    mean = torch.empty(Cc, device=dev, dtype=torch.float32) if save else None
    invstd = torch.empty(Cc, device=dev, dtype=torch.float32) if save else None
    L.check(L.lib().spk_bn_finalize(stats.data_ptr() if stats is not None else None, slots, int(count), L.dptr(gamma, "gamma"),
                                    L.dptr(beta, "beta"), L.dptr(running_mean, "running_mean"),
                                    L.dptr(running_var, "running_var"), float(momentum), float(eps), L.dptr(scale),
                                    L.dptr(shift), L.dptr(mean), L.dptr(invstd), Cc, L.stream_ptr()), "spk_bn_finalize")
    return (scale, shift, mean, invstd) if save else (scale, shift)


def bn_replay_running(items, momentum):
    """The second running-statistics update of a pass for many BatchNorms on one launch per 64 (``spk_bn_replay_list``):
    ``items`` = [(stats totals fp64 [2C], count, running_mean [C], running_var [C])]."""
    items = list(items)
    for i in range(0, len(items), L.BN_LIST_MAX):
        part = items[i:i + L.BN_LIST_MAX]
        arr = (L.BnReplayItem * len(part))()
        for j, (stats, count, rm, rv) in enumerate(part):
            Cc = rm.numel()
            if stats.dtype != torch.float64 or stats.numel() < 2 * Cc or rv.numel() != Cc or not stats.is_contiguous():
                raise L.SpkError("bn_replay_running: stats must be a contiguous float64 [2C] next to running buffers of C")
            arr[j] = L.BnReplayItem(stats=stats.data_ptr(), running_mean=L.dptr(rm, "running_mean"),
                                    running_var=L.dptr(rv, "running_var"), count=int(count), C=Cc)
        L.check(L.lib().spk_bn_replay_list(arr, len(part), float(momentum), L.stream_ptr()), "spk_bn_replay_list")


def bn_add_relu(a, sa, ba, b=None, sb=None, bb=None, relu=True):
    """y = [relu](a*sa[c] + ba[c] + (b*sb[c] + bb[c])) -- BatchNorm apply + residual add + ReLU, one pass."""
    B, Cc, H, W = a.shape
    out = torch.empty_like(a)
    L.check(L.lib().spk_bn_add_relu_fwd(L.dptr(a, "a"), L.dptr(sa, "sa"), L.dptr(ba, "ba"), L.dptr(b, "b"),
                                        L.dptr(sb, "sb"), L.dptr(bb, "bb"), L.dptr(out), B, Cc, H * W,
                                        1 if relu else 0, L.stream_ptr()), "spk_bn_add_relu_fwd")
    return out


def maxpool3x3s2(x, in_scale=None, in_shift=None):
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_maxpool3x3s2_fwd(L.dptr(x, "x"), L.dptr(in_scale, "in_scale"), L.dptr(in_shift, "in_shift"),
                                         L.dptr(out), B, Cc, H, W, L.stream_ptr()), "spk_maxpool3x3s2_fwd")
    return out


def global_avgpool(x):
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, 1, 1), device=x.device, dtype=torch.float32)
    L.check(L.lib().spk_global_avgpool_fwd(L.dptr(x, "x"), L.dptr(out), B * Cc, H * W, L.stream_ptr()),
            "spk_global_avgpool_fwd")
    return out
