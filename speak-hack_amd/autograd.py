"""``torch.autograd.Function`` wrappers: forward and backward of every decoder op run on the HIP
kernels (include/spk.h); autograd only sequences them.  No torch math on activations -- the only
torch arithmetic in a backward is on per-(batch,channel) scalars the kernels already reduced
(e.g. summing ``[B,C]`` partials over ``B``).

Double backward: the discriminator's Functions (``ConvBiasLReLUFn``, ``ConvDgradFn``, ``_LReluMaskFn``, ``FCFn``,
``GlobalAvgPoolFn``) are built to be differentiated twice -- the R1 penalty of train.py:246-255 needs that.  Every other
backward here runs raw kernels and is marked ``once_differentiable``: a ``create_graph=True`` pass through it raises
instead of silently dropping the second-order term.
"""
from __future__ import annotations

import contextlib
import weakref

import torch
from torch.utils.weak import WeakTensorKeyDictionary

from . import ops


_INPUT_GRAD_ONLY = False


@contextlib.contextmanager
def input_grad_only():
    """Inside this context a *recorded* backward (``create_graph=True``) of the discriminator layers computes the data
    gradient only.  ``torch.autograd.grad(D(x).sum(), x, create_graph=True)`` (the R1 penalty, train.py:246-255) asks
    for dD/dx alone, but a custom Function cannot see that and would also run one weight-gradient kernel per layer
    whose results autograd then throws away; the penalty's own gradient w.r.t. the weights does not use them (it flows
    through ``ConvDgradFn``).  Leave the context before calling ``backward`` on the penalty."""
    global _INPUT_GRAD_ONLY
    prev, _INPUT_GRAD_ONLY = _INPUT_GRAD_ONLY, True
    try:
        yield
    finally:
        _INPUT_GRAD_ONLY = prev


# ---- weight gradients whose only consumer is SpectralNormAllFn.backward: on the second stream ------------------------------------
# A discriminator conv's weight is W / sigma, an output of SpectralNormAllFn; its gradient sits untouched in that node's input
# buffer until the node runs -- after the LAST conv's backward -- as long as ONE tensor is handed to the engine per weight.  An
# R1 pass adds ConvDgradFn as a second user of the weight and the engine would ADD the two gradients on the current stream, so
# every gradient of a normalised weight goes through _wgrad_shared: the first user of a backward pass allocates it, later
# users accumulate into that tensor inside the kernel.  The weight-gradient launches go to ops.side_stream behind the producer
# of their operands, the data-gradient chain never waits for them, and SpectralNormAllFn.backward joins the stream before it
# reads the gradients (a callback at the end of the backward pass joins the launching stream as well, whatever else happens
# to the graph).  tools/lab_wgrad_overlap.py: 1-5 % of the discriminator's layers.
_side_pending = set()                     # (device index, graph task id) with an end-of-backward join queued


def _side_join(device, stream=None):
    """``stream`` (default: the current one) waits for every weight gradient queued on the device's second stream so far.
    Unconditional -- waiting on a stream with nothing queued costs an event -- so that two backward passes on one device
    (other threads, other streams) cannot cancel each other's join."""
    st = ops.side_stream(device)
    if st is not None:
        (stream if stream is not None else torch.cuda.current_stream(device)).wait_stream(st)


def _wgrad_aside(dt, x, launch, extra=()):
    """``launch()`` (a weight gradient reading the temporaries ``dt`` and ``x``, and those in ``extra``) on the second stream behind
    everything queued on the current stream so far; in order on the current stream when there is no second stream, under a
    stream capture, or outside a backward pass (no place to hang the final join)."""
    dev = dt.device
    side = ops.side_stream(dev)
    if side is None or torch.cuda.is_current_stream_capturing():
        return launch()
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, torch._C._current_graph_task_id())
    if key not in _side_pending:
        def done(dev=dev, cur=cur, key=key):         # holds the device, the launching stream and the key -- never a tensor
            _side_pending.discard(key)
            _side_join(dev, cur)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(done)
        except RuntimeError:
            return launch()
        _side_pending.add(key)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ops.side_stream_delay()
        dw = launch()
    dt.record_stream(side)
    x.record_stream(side)
    for t in extra:
        t.record_stream(side)
    return dw


def _wgrad_shared(weight, a, b, make):
    """The gradient of a spectrally normalised weight (an output of SpectralNormAllFn), however many users it has in the graph
    -- one in a first-order pass; two in an R1 pass (the conv itself and ConvDgradFn); more if a caller reuses a layer.  The
    first user to run in a backward pass allocates the gradient and hands it to autograd; every later one ADDS into that
    tensor inside the weight-gradient kernel (``make(out, accumulate=True)``) and hands autograd nothing.  So exactly one
    tensor reaches SpectralNormAllFn.backward, the engine never adds two contributions on the current stream, and all the
    launches may sit on the second stream, in order among themselves.  SpectralNormAllFn.backward drops the note
    (``_spk_dw``) when it has consumed the gradient.  ``make(out, accumulate)`` launches the kernel; ``a``, ``b`` are its
    temporaries (see _wgrad_aside)."""
    task = torch._C._current_graph_task_id()
    held = getattr(weight, "_spk_dw", None)
    if held is None or held[0] != task:
        dw = _wgrad_aside(a, b, lambda: make(None, False))
        weight._spk_dw = (task, dw)
        return dw
    _wgrad_aside(a, b, lambda: make(held[1], True))
    return None


def _first_or_add(param, key, make):
    """One gradient tensor per parameter and backward pass, whatever the number of graph nodes that contribute to it (a
    discriminator runs six forwards before one backward, train.py:160-182): ``make(None)`` by the first contributor -- that tensor
    goes to autograd --, ``make(held)`` (accumulate into it, on the current stream) by every later one, which hands autograd nothing.
    The engine would otherwise add the contributions pairwise: 156 ``add`` launches per discriminator step."""
    task = torch._C._current_graph_task_id()
    held = getattr(param, key, None)
    if held is None or held[0] != task:
        t = make(None)
        # (the note keeps an ALIAS: a second reference to the tensor object itself would stop AccumulateGrad from adopting it --
        # it clones any gradient whose use count is above one)
        setattr(param, key, (task, t.detach()))
        return t
    make(held[1])
    return None


class WeightGateFn(torch.autograd.Function):
    """Identity on a module's conv weights, applied ONCE before the first conv of a forward pass (SynthesisNetwork.forward).  It is
    created first, so in the backward pass it runs after every conv's backward: the one place where the current stream has to
    wait for the weight gradients that those backwards queued on the second stream, before autograd hands them to the
    parameters (AccumulateGrad, gradient hooks, the data-parallel reducer).  What SpectralNormAllFn is for the discriminator."""

    @staticmethod
    def forward(ctx, *weights):
        ctx.dev = weights[0].device
        return tuple(w.view_as(w) for w in weights)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        _side_join(ctx.dev)
        return grads


def gate_weights(weights):
    """-> the weights behind a WeightGateFn node, each tagged with the parameter it stands for (``_spk_gate_of``: the packed-image
    caches key on the parameter; FusedConvFn.backward puts the weight gradient of a tagged weight on the second stream)."""
    outs = WeightGateFn.apply(*weights)
    for o, w in zip(outs, weights):
        o._spk_gate_of = w
    return outs


def _needs(ctx, grad_mode):
    """Whether to keep tensors for backward.  Grad mode is always off *inside* Function.forward and
    ``needs_input_grad`` ignores ``torch.no_grad()``, so the caller samples the mode and passes it in."""
    return bool(grad_mode) and any(ctx.needs_input_grad)


def _epilogue_param_grads(sums, has_style, has_bias, has_noise):
    """(d style [B,2C] = [d s0 | d s1], d bias [C], d noise_w [C]) from ``ops.epilogue_bwd``'s sums [B,4,C]: the style
    gradient is a view of rows 0-1, bias and noise weight come out of ONE reduction over the batch."""
    B, _, Cc = sums.shape
    dstyle = sums[:, :2].reshape(B, 2 * Cc) if has_style else None
    dbias = dnw = None
    if has_bias and has_noise:
        red = sums[:, 2:].sum(0)
        dbias, dnw = red[0], red[1]
    elif has_bias:
        dbias = sums[:, 2].sum(0)
    elif has_noise:
        dnw = sums[:, 3].sum(0)
    return dstyle, dbias, dnw


class FusedConvFn(torch.autograd.Function):
    """y = style(lrelu(conv3x3(up?(x)) + bias + noise_w*noise)) -- styleganv1.py:624-628 / :630-633.

    Backward: one epilogue-adjoint pass (dt + the four per-plane sums that give d style, d bias,
    d noise_w), the data gradient as the same MFMA conv on transpose-flipped weights (+ the bilinear
    adjoint), the weight gradient on the MFMA wgrad kernel (upsampling re-formed on the fly)."""

    @staticmethod
    def forward(ctx, x, weight, bias, noise_w, noise, style, upsample, slope, packed, grad_mode, w_scale=1.0):
        """``w_scale``: the conv runs on ``weight * w_scale`` (equalised learning rate, stylegan.py:31-46) without that
        product ever existing: the factor rides on the accumulator (``out_scale``) forward and on dx / dw backward."""
        B, Cin, Hs, Ws = x.shape
        Cout = weight.shape[0]
        H, W = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
        keep = _needs(ctx, grad_mode)
        a = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32) if keep else None
        pw = getattr(weight, "_spk_gate_of", weight)      # (behind a WeightGateFn: the parameter itself keys the packed images)
        if ops.train_bf16x3(B, Cin, Cout, H, W):     # opt-in split-precision training (ops.train_conv_precision)
            y = ops.conv3x3_bf16x3(x, packed.get_bf16x3(pw), Cout, bias=bias, noise_w=noise_w, noise=noise, style=style,
                                   upsample=upsample, lrelu_slope=slope, out_pre=a, out_scale=w_scale)
        elif ops.use_wino(B, Cin, Cout, H, W):       # fp32 Winograd (ops.CONV3X3_ALGO); a x2 layer reads the materialised x2 image
            xin = ops.upsample2x_bilinear(x) if upsample else x
            y = ops.conv3x3_wino(xin, packed.get_wino(pw), Cout, bias=bias, noise_w=noise_w, noise=noise, style=style,
                                 lrelu_slope=slope, out_pre=a, out_scale=w_scale)
            if upsample and keep and ops.use_wgrad_wino(B, Cin, Cout, H, W):
                x = xin                               # the Winograd weight gradient reads the x2 image too: keep IT, not the source
                ctx.x_was_up = True
        else:
            cfg = ops.conv2d_pick_config(3, 1, B, Cin, Cout, H, W)
            y = ops.conv2d_fused(x, packed.get(pw, cfg), Cout, 3, 1, bias=bias, noise_w=noise_w, noise=noise,
                                 style=style, upsample=upsample, lrelu_slope=slope, config=cfg, out_pre=a, out_scale=w_scale)
        if keep:
            ctx.save_for_backward(x, weight, a, noise, style)
            ctx.conf = (upsample, slope, packed, bias is not None, noise_w is not None, float(w_scale))
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, a, noise, style = ctx.saved_tensors
        upsample, slope, packed, has_bias, has_noise, w_scale = ctx.conf
        x_is_up = upsample and getattr(ctx, "x_was_up", False)       # (the saved x is the x2 image already)
        B, Cin = x.shape[:2]
        Cout, H, W = a.shape[1:]
        dt, sums = ops.epilogue_bwd(dy.contiguous(), a, noise if has_noise else None, style,
                                    slope if slope is not None else 1.0)
        dstyle, dbias, dnw = _epilogue_param_grads(sums, style is not None, has_bias, has_noise)
        dx = dw = None
        pw = getattr(weight, "_spk_gate_of", None)
        if ctx.needs_input_grad[1]:
            launch = lambda: ops.conv2d_wgrad(dt, x, Cout, Cin, 3, 1, upsample=upsample and not x_is_up, scale=w_scale)
            # behind a WeightGateFn (which joins the second stream before the parameters see their gradients): beside the data gradient
            dw = _wgrad_aside(dt, x, launch) if pw is not None else launch()
        if pw is None:
            pw = weight
        if ctx.needs_input_grad[0]:
            if ops.train_bf16x3(B, Cout, Cin, H, W):
                dx = ops.conv3x3_bf16x3(dt, packed.get_bf16x3(pw, transpose_flip=True), Cin, out_scale=w_scale)
            elif ops.use_wino(B, Cout, Cin, H, W):
                dx = ops.conv3x3_wino(dt, packed.get_wino(pw, transpose_flip=True), Cin, out_scale=w_scale)
            else:
                cfg = ops.conv2d_pick_config(3, 1, B, Cout, Cin, H, W)
                dx = ops.conv2d_fused(dt, packed.get(pw, cfg, transpose_flip=True), Cin, 3, 1, config=cfg, out_scale=w_scale)
            if upsample:
                dx = ops.upsample2x_bilinear_bwd(dx)
        return dx, dw, dbias, dnw, None, dstyle, None, None, None, None, None


class FCFn(torch.autograd.Function):
    """out = lrelu_slope(wmul * x @ W^T + bmul * bias) -- FC.forward, styleganv1.py:489-495."""

    @staticmethod
    def forward(ctx, x, weight, bias, wmul, bmul, slope, grad_mode):
        out = ops.fc(x, weight, bias, wmul, bmul, slope)
        if _needs(ctx, grad_mode):
            ctx.save_for_backward(x, weight, out)
            ctx.conf = (wmul, bmul, slope, bias is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        wmul, bmul, slope, has_bias = ctx.conf
        if torch.is_grad_enabled():        # double backward (R1): the input gradient stays differentiable, on the same HIP kernels
            dx = FCDgradFn.apply(dout, out, weight, wmul, slope) if ctx.needs_input_grad[0] else None
            dw = db = None
            if not _INPUT_GRAD_ONLY and (ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])):
                with torch.no_grad():       # parameter gradients of a recorded pass: plain kernel outputs (as ConvBiasLReLUFn's)
                    _, dw, db = ops.fc_bwd(dout.detach().contiguous(), out, x, weight, wmul, bmul, slope, need_dx=False,
                                           need_dw=True, has_bias=has_bias)
            return dx, dw, db, None, None, None, None
        dx, dw, db = ops.fc_bwd(dout.contiguous(), out, x, weight, wmul, bmul, slope,
                                need_dx=ctx.needs_input_grad[0], need_dw=ctx.needs_input_grad[1] or
                                (has_bias and ctx.needs_input_grad[2]), has_bias=has_bias)
        return dx, dw, db, None, None, None, None


class FCDgradFn(torch.autograd.Function):
    """dx = wmul * (dout * lrelu'(out)) @ W as a differentiable function of (dout, W): the input gradient of ``FCFn`` inside a recorded
    backward (the R1 penalty of train.py:246-255 differentiates dD/dx once more).  Its adjoints are kernels that exist already -- with
    g the gradient w.r.t. dx: d dout = lrelu'(out) * wmul * g @ W^T (``spk_fc_fwd``), d W = wmul * (dout * lrelu'(out))^T @ g (the weight
    half of ``spk_fc_bwd`` with g in x's place) -- so no GEMM library call sits on the discriminator's path (round 3: 14 Tensile
    launches per step here)."""

    @staticmethod
    def forward(ctx, dout, out, weight, wmul, slope):
        dout = dout.contiguous()
        ctx.save_for_backward(dout, out, weight)
        ctx.conf = (wmul, slope)
        dx, _, _ = ops.fc_bwd(dout, out, dout.new_empty((dout.shape[0], weight.shape[1])), weight, wmul, 0.0, slope, need_dx=True,
                              need_dw=False, has_bias=False)
        return dx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        dout, out, weight = ctx.saved_tensors
        wmul, slope = ctx.conf
        g = g.contiguous()
        ddout = dweight = None
        if ctx.needs_input_grad[0]:
            ddout = ops.fc(g, weight, None, wmul, 0.0, 1.0)                  # wmul * g @ W^T ...
            if slope != 1.0:
                ddout = ddout * torch.where(out > 0, 1.0, float(slope))     # ... through the same mask
        if ctx.needs_input_grad[2]:
            _, dweight, _ = ops.fc_bwd(dout, out, g, weight, wmul, 0.0, slope, need_dx=False, need_dw=True, has_bias=False)
        return ddout, None, dweight, None, None


class StyleFCGroupFn(torch.autograd.Function):
    """The style affines of a synthesis pass -- ``ApplyStyle.linear`` of every layer (styleganv1.py:463-468) applied to its row
    of the latent ``w`` [B, L, 512] -- as ONE grouped launch, and their backward as two: the input gradients are written
    straight into the rows of one [B, L, 512] latent gradient (13 FCFn nodes cost 13 + 26 launches and 13 row scatters)."""

    @staticmethod
    def forward(ctx, w, confs, grad_mode, *params):
        # confs: per layer (wmul, bmul, slope, has_bias); params: weight_0, bias_0 | None, weight_1, ...
        n = len(confs)
        ws, bs = params[0::2], params[1::2]
        outs = ops.fc_grouped((w[:, j], ws[j], bs[j], confs[j][0], confs[j][1], confs[j][2]) for j in range(n))
        if _needs(ctx, grad_mode):
            ctx.save_for_backward(w, *ws, *outs)
            ctx.confs = confs
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *douts):
        confs = ctx.confs
        n = len(confs)
        saved = ctx.saved_tensors
        w, ws, outs = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n]
        B = w.size(0)
        need_dx = ctx.needs_input_grad[0]
        # rows of w that feed no layer (or whose layer got no gradient) keep a zero gradient
        live = [j for j in range(n) if douts[j] is not None]
        dwl = (torch.zeros_like(w) if (w.size(1) > n or len(live) < n) else torch.empty_like(w)) if need_dx else None
        res = ops.fc_grouped_bwd(((douts[j], outs[j], w[:, j], ws[j], dwl[:, j] if need_dx else None,
                                   ctx.needs_input_grad[3 + 2 * j] or (confs[j][3] and ctx.needs_input_grad[4 + 2 * j]),
                                   confs[j][3], confs[j][0], confs[j][1], confs[j][2]) for j in live), B) if live else []
        grads = [None] * (2 * n)
        for j, (dw, db) in zip(live, res):
            grads[2 * j], grads[2 * j + 1] = dw, db
        return (dwl, None, None, *grads)


def style_fc_group(w, linears, slope):
    """``linears``: the FC modules (decoder.FC) of the style affines, layer j reading ``w[:, j]`` -> tuple of [B, 2C] styles."""
    confs = tuple((float(m.w_lrmul), float(m.b_lrmul), float(slope), m.bias is not None) for m in linears)
    params = [t for m in linears for t in (m.weight, m.bias)]
    return StyleFCGroupFn.apply(w, confs, torch.is_grad_enabled(), *params)


class ToRGBFn(torch.autograd.Function):
    """1x1 conv to <= 4 channels -- styleganv1.py:607."""

    @staticmethod
    def forward(ctx, x, weight, bias, grad_mode):
        y = ops.conv1x1_small(x, weight, bias)
        if _needs(ctx, grad_mode):
            ctx.save_for_backward(x, weight)
            ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = ops.conv1x1_small_bwd(x, weight, dy.contiguous(), need_dx=ctx.needs_input_grad[0])
        return dx, dw, (db if ctx.has_bias else None), None


class BiasNoiseStyleFn(torch.autograd.Function):
    """y = (x + bias + noise_w*noise) * (s0+1) + s1 -- the decoder prologue (styleganv1.py:596-599) and the
    stand-alone ApplyNoise / ApplyStyle.  ``x`` may be a [1,C,H,W] constant broadcast over the batch."""

    @staticmethod
    def forward(ctx, x, bias, noise_w, noise, style, B, grad_mode):
        y = ops.bias_noise_style(x, B, bias, noise_w, noise, style)
        if _needs(ctx, grad_mode):
            # value before the style stage: needed for d s0 = sum dy * pre
            pre = None
            if style is not None:
                pre = ops.bias_noise_style(x, B, bias, noise_w, noise, None) if (bias is not None or noise is not None
                                                                                   or x.size(0) != B) else x
            ctx.save_for_backward(pre, noise, style)
            ctx.conf = (x.size(0) == 1 and B > 1, bias is not None, noise_w is not None and noise is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        pre, noise, style = ctx.saved_tensors
        broadcast, has_bias, has_noise = ctx.conf
        dt, sums = ops.epilogue_bwd(dy.contiguous(), pre, noise if has_noise else None, style, 1.0)
        dstyle, dbias, dnw = _epilogue_param_grads(sums, style is not None, has_bias, has_noise)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dt.sum(0, keepdim=True) if broadcast else dt      # [1,C,4,4] constant: B tiny planes
        return dx, dbias, dnw, None, dstyle, None, None


class _LReluMaskFn(torch.autograd.Function):
    """dt = dy * lrelu'(y) (y = the saved activation: same sign as the pre-activation).  Linear in dy and its own
    adjoint, so it is differentiable to any order -- the first-order backward of a conv+LeakyReLU layer is built from
    this and ``ConvDgradFn`` when a double backward is requested (R1, train.py:246-255)."""

    @staticmethod
    def forward(ctx, dy, y, slope):
        ctx.save_for_backward(y)
        ctx.slope = slope
        return ops.epilogue_bwd(dy.contiguous(), y, None, None, slope)[0]

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _LReluMaskFn.apply(g, y, ctx.slope), None, None


def _packed(weight, cfg, tf=False):
    """-> (packed image, device scalar for ``out_scale_dev`` or None).  A spectrally normalised weight that carries
    ``_spk_sn = (weight_orig, 1 / sigma)`` (``SpectralNormAllFn``) is served by the image of ``weight_orig`` -- cached on the
    Parameter, so it is packed once per optimizer step instead of once per forward (the discriminator runs six forwards per step:
    164 -> 34 pack launches) -- and the scalar goes into the conv epilogue.  The bf16x3 path keeps W / sigma's own image."""
    sn = getattr(weight, "_spk_sn", None)
    if sn is not None and cfg not in ("bf16x3", 12) + ops.GEMM2_CONFIGS:     # (out_scale_dev lives in the tap kernels)
        return _packed_of(sn[0], cfg, tf), sn[1]
    return _packed_of(weight, cfg, tf), None


_PACKS = WeakTensorKeyDictionary()


def _packed_of(weight, cfg, tf=False):
    """Packed image of a per-forward weight tensor (the discriminator's W / sigma: a fresh tensor every forward, so the
    per-parameter cache of ``ops.PackedConvWeight`` cannot hold it).  The images are held per tensor object (weakly) and die with
    it: the forward pack is reused by the R1 double backward, the data-gradient pack by both backward passes (64 of the 220
    pack launches of a discriminator step)."""
    cache = _PACKS.get(weight)             # weakly keyed by the tensor OBJECT: nothing rides in Parameter.__dict__, which
    if cache is None:                      # ``torch.save(model)`` / any pickling of Parameters would carry along
        cache = _PACKS[weight] = {}
    key = (cfg, tf, weight._version, weight.data_ptr())
    hit = cache.get(key)
    if hit is None:
        if cache and next(iter(cache))[2:] != key[2:]:
            cache.clear()                 # the tensor was updated in place (a Parameter after an optimizer step): drop old images
        if cfg == "wino":
            hit = ops.pack_conv_weight_wino(weight.detach(), transpose_flip=bool(tf))
        elif cfg == "bf16x3":
            hit = ops.pack_conv_weight_bf16x3(weight.detach(), transpose_flip=bool(tf))
        else:
            hit = ops.pack_conv_weight(weight.detach(), cfg, transpose_flip=tf)
        cache[key] = hit
    return hit


def _sn_parts(weight):
    """(weight_orig, 1 / sigma as a device scalar) of a spectrally normalised weight, or (the weight, None)."""
    sn = getattr(weight, "_spk_sn", None)
    return (sn[0].detach(), sn[1]) if sn is not None else (weight.detach(), None)


def _conv_plain(x, weight, k, stride):
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    Ho, Wo = ops.conv_out_size(H, k, stride), ops.conv_out_size(W, k, stride)
    if ops.conv1x1_expand_ok(x, Cin, k, stride):                 # fromRGB: a store stream, not a contraction
        w0, sd = _sn_parts(weight)
        return ops.conv1x1_expand(x.contiguous(), w0, None, sd)
    if k == 3 and stride == 1 and ops.train_bf16x3(B, Cin, Cout, Ho, Wo):
        return ops.conv3x3_bf16x3(x.contiguous(), _packed(weight, "bf16x3")[0], Cout)
    if k == 3 and stride == 1 and ops.use_wino(B, Cin, Cout, Ho, Wo):
        wp, sd = _packed(weight, "wino")
        return ops.conv3x3_wino(x.contiguous(), wp, Cout, out_scale_dev=sd)
    cfg = ops.conv2d_pick_config(k, stride, B, Cin, Cout, Ho, Wo)
    wp, sd = _packed(weight, cfg)
    return ops.conv2d_fused(x, wp, Cout, k, stride, config=cfg, out_scale_dev=sd)


def _conv_dgrad(dt, weight, k, stride, in_hw):
    B, Cout = dt.shape[:2]
    Cin = weight.shape[1]
    if k == 1 and stride == 1 and Cin <= 4:                      # fromRGB's data gradient: a 1x1 conv TO <= 4 channels (the toRGB kernel)
        return ops.conv1x1_small(dt.contiguous(), weight.detach().reshape(Cout, Cin).t().contiguous().view(Cin, Cout, 1, 1))
    if k == 3 and stride == 1 and ops.train_bf16x3(B, Cout, Cin, in_hw[0], in_hw[1]):
        return ops.conv3x3_bf16x3(dt, _packed(weight, "bf16x3", True)[0], Cin)
    if k == 3 and stride == 1 and ops.use_wino(B, Cout, Cin, in_hw[0], in_hw[1]):
        wp, sd = _packed(weight, "wino", True)
        return ops.conv3x3_wino(dt.contiguous(), wp, Cin, out_scale_dev=sd)
    cfg, tf = ops.dgrad_plan(k, stride, B, Cout, Cin, in_hw, dt.shape[-2:])
    wp, sd = _packed(weight, cfg, tf)
    return ops.conv2d_dgrad(dt, wp, Cin, k, stride, in_hw, cfg, out_scale_dev=sd)


class ConvDgradFn(torch.autograd.Function):
    """dx = conv_transpose(dt, w) as a differentiable function of (dt, w): its adjoints are the forward conv
    (w.r.t. dt) and the weight-gradient kernel (w.r.t. w) -- the same MFMA kernels, so the R1 double backward of the
    discriminator (d/dw of |dD/dx|^2) never leaves the HIP path."""

    @staticmethod
    def forward(ctx, dt, weight, k, stride, in_hw):
        ctx.save_for_backward(dt, weight)
        ctx.conf = (k, stride)
        return _conv_dgrad(dt, weight, k, stride, in_hw)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        dt, weight = ctx.saved_tensors
        k, stride = ctx.conf
        g = g.contiguous()
        ddt = dw = None
        if ctx.needs_input_grad[1]:
            Co, Ci = weight.shape[:2]
            if getattr(weight, "_spk_sn", None) is not None:
                dw = _wgrad_shared(weight, dt, g, lambda out, acc: ops.conv2d_wgrad(dt, g, Co, Ci, k, stride, out=out, accumulate=acc))
            else:
                dw = ops.conv2d_wgrad(dt, g, Co, Ci, k, stride)
        if ctx.needs_input_grad[0]:
            ddt = _conv_plain(g, weight, k, stride)
        return ddt, dw, None, None, None


class ConvBiasLReLUFn(torch.autograd.Function):
    """y = lrelu_slope(conv_kxk(x, w, stride) + bias) -- the discriminator's layers (styleganv1.py:662-695): one fused
    launch forward; backward on the epilogue-adjoint / dgrad / wgrad kernels.  When the backward itself is recorded
    (``create_graph=True``) the data path is rebuilt from ``_LReluMaskFn`` and ``ConvDgradFn`` so that it can be
    differentiated once more; the parameter gradients of that pass are plain (non-differentiable) kernel outputs."""

    @staticmethod
    def forward(ctx, x, weight, bias, k, stride, slope, grad_mode):
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        Ho, Wo = ops.conv_out_size(H, k, stride), ops.conv_out_size(W, k, stride)
        if ops.conv1x1_expand_ok(x, Cin, k, stride):             # fromRGB (3 -> 64 at 256^2): a store stream, not a contraction
            w0, sd = _sn_parts(weight)
            y = ops.conv1x1_expand(x.contiguous(), w0, bias, sd, slope)
        elif k == 3 and stride == 1 and ops.train_bf16x3(B, Cin, Cout, Ho, Wo):
            y = ops.conv3x3_bf16x3(x.contiguous(), _packed(weight, "bf16x3")[0], Cout, bias=bias, lrelu_slope=slope)
        elif k == 3 and stride == 1 and ops.use_wino(B, Cin, Cout, Ho, Wo):
            wp, sd = _packed(weight, "wino")
            y = ops.conv3x3_wino(x.contiguous(), wp, Cout, bias=bias, lrelu_slope=slope, out_scale_dev=sd)
        else:
            cfg = ops.conv2d_pick_config(k, stride, B, Cin, Cout, Ho, Wo)
            wp, sd = _packed(weight, cfg)
            y = ops.conv2d_fused(x, wp, Cout, k, stride, bias=bias, lrelu_slope=slope, config=cfg, out_scale_dev=sd)
        if _needs(ctx, grad_mode):
            ctx.save_for_backward(x, weight, y)
            ctx.conf = (k, stride, slope, bias is not None)
            ctx.bias_ref = bias               # (the Parameter: its gradient is shared by the passes of one backward, _first_or_add)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        k, stride, slope, has_bias = ctx.conf
        Cout, Cin = weight.shape[:2]
        dx = dw = db = None
        if torch.is_grad_enabled():        # double backward requested: differentiable data path
            dt = _LReluMaskFn.apply(dy, y, slope) if slope is not None else dy.contiguous()
            if ctx.needs_input_grad[0]:
                dx = ConvDgradFn.apply(dt, weight, k, stride, tuple(x.shape[-2:]))     # a second user of this weight: see _wgrad_shared
            if not _INPUT_GRAD_ONLY:
                with torch.no_grad():
                    dtd = dt.detach()
                    if ctx.needs_input_grad[1]:
                        dw = ops.conv2d_wgrad(dtd, x, Cout, Cin, k, stride)
                    if has_bias and ctx.needs_input_grad[2]:
                        db = dtd.sum((0, 2, 3))
            return dx, dw, db, None, None, None, None
        dt, sums = ops.epilogue_bwd(dy.contiguous(), y, None, None, slope if slope is not None else 1.0)
        if ctx.needs_input_grad[1]:
            if getattr(weight, "_spk_sn", None) is None:
                dw = ops.conv2d_wgrad(dt, x, Cout, Cin, k, stride)
            else:                                                 # consumed by SpectralNormAllFn.backward only: second stream
                dw = _wgrad_shared(weight, dt, x, lambda out, acc: ops.conv2d_wgrad(dt, x, Cout, Cin, k, stride, out=out, accumulate=acc))
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad(dt, weight, k, stride, tuple(x.shape[-2:]))
        if has_bias and ctx.needs_input_grad[2]:
            if isinstance(ctx.bias_ref, torch.nn.Parameter):
                db = _first_or_add(ctx.bias_ref, "_spk_db", lambda out: ops.plane_sums_reduce(sums, 2, out))
            else:
                db = ops.plane_sums_reduce(sums, 2)
        return dx, dw, db, None, None, None, None


class GlobalAvgPoolFn(torch.autograd.Function):
    """AdaptiveAvgPool2d((1,1)) -- styleganv1.py:676.  The adjoint is a broadcast, written with differentiable torch
    views so that a double backward passes through it."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return ops.global_avgpool(x)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        return (dy / float(H * W)).expand(B, C, H, W)


def _scale_rows(t):
    """[B,C] per-plane factors -> the [B,2C] style rows (s0 | s1) with s0 + 1 = t, s1 = 0."""
    return torch.cat([t - 1.0, torch.zeros_like(t)], dim=1).contiguous()


class ModConvFn(torch.autograd.Function):
    """StyleGAN2 styled conv (build-defined variant; formulas reference/styleganv2.txt:1835,1912):
    y = gain * lrelu(d[b,co] * scale * conv3x3(up?(x) * s[b,ci], w) + noise_w*noise + bias), d = rsqrt(scale^2 sum_{ci,k}
    (w s)^2 + eps) computed HERE from (w, s) (``spk_modconv_demod``; ``demodulate`` False: d = 1).

    Backward, all on the HIP kernels and without a single rescaled activation in HBM:
      1. epilogue adjoint: dz = dy * lrelu' (gain folded later) + the plane sums that give d bias, d noise_w and d d;
      2. data gradient: the same modulated MFMA conv with the roles of s and d swapped (dz * d*gain on the way in);
      3. ``spk_modconv_dx_finish``: the upfirdn2d adjoint, x s on the way out and d s = <up^T(dx~), x> at the LOW resolution;
      4. weight gradient: ``spk_conv2d_wgrad`` with SPK_CONV_IN_BATCH_SCALE -- x * s and dz * d*gain are formed while the
         tiles are staged, the x2 image interpolated LDS -> LDS (zero border);
      5. ``spk_modconv_demod_bwd``: d d into d s and d w.
    Round 2 materialised up(x) (268 MB at [8,128,256,256]), dz * d and up(x) * s per layer and ran d's adjoint as ATen GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, s, bias, noise_w, noise, scale, upsample, slope, gain, fir, packed, demodulate, grad_mode):
        B, Cin, Hs, Ws = x.shape
        Cout = weight.shape[0]
        H, W = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
        d = ops.modconv_demod(weight, s, scale) if demodulate else None
        pw = getattr(weight, "_spk_gate_of", weight)      # (behind a WeightGateFn: the parameter keys the packed images)
        if ops.use_wino(B, Cin, Cout, H, W) and (not upsample or Ws % 4 == 0):
            # fp32 Winograd with the modulation applied to the transformed input and the demodulation in the epilogue; a x2 layer
            # reads the materialised upfirdn2d(up = 2, [1,3,3,1]) image
            xin = ops.upsample2x(x, zero_border=True) if upsample else x
            y = ops.conv3x3_wino(xin, packed.get_wino(pw), Cout, bias=bias, noise_w=noise_w, noise=noise, lrelu_slope=slope,
                                 out_scale=scale, batch_scale=s.contiguous(), demod=d, act_gain=gain)
        else:
            cfg = ops.conv2d_pick_config(3, 1, B, Cin, Cout, H, W)
            cfg = cfg + 4 if cfg < 4 else cfg
            y = ops.conv2d_fused(x, packed.get(pw, cfg), Cout, 3, 1, bias=bias, noise_w=noise_w, noise=noise, lrelu_slope=slope,
                                 out_scale=scale, batch_scale=s, demod=d, act_gain=gain, config=cfg, upsample=upsample, up_fir=True)
        if _needs(ctx, grad_mode):
            ctx.save_for_backward(x, weight, s, d, y, noise, bias, noise_w)
            ctx.conf = (scale, upsample, slope, gain, fir, packed)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, s, d, y, noise, bias, noise_w = ctx.saved_tensors
        scale, upsample, slope, gain, fir, packed = ctx.conf
        B, Cin = x.shape[:2]
        Cout, H, W = y.shape[1:]
        dt, sums = ops.epilogue_bwd(dy.contiguous(), y, noise, None, slope if slope is not None else 1.0)   # dz = gain * dt
        # y/gain = lrelu(d*c + n + b): sum dy*y = sum dz*(d*c + n + b) -> d d = (that - the noise and bias parts) / d; d' = d * gain
        dd, dprime, dbias, dnw = ops.modconv_epi_finish(sums, d, bias, noise_w, gain)
        s = s.contiguous()
        need_dx, need_dw, need_ds = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx = dw = ds = None
        if need_dx or need_ds:
            if ops.use_wino(B, Cout, Cin, H, W):
                dxt = ops.conv3x3_wino(dt, packed.get_wino(getattr(weight, "_spk_gate_of", weight), transpose_flip=True), Cin,
                                       out_scale=scale, batch_scale=dprime.contiguous())
            else:
                cfg = ops.conv2d_pick_config(3, 1, B, Cout, Cin, H, W)
                cfg = cfg + 4 if cfg < 4 else cfg
                dxt = ops.conv2d_fused(dt, packed.get(getattr(weight, "_spk_gate_of", weight), cfg, transpose_flip=True), Cin, 3, 1,
                                       out_scale=scale, batch_scale=dprime, config=cfg)    # d (up(x) * s), at the output resolution
            if upsample and (x.shape[-1] % 2 or dxt.data_ptr() % 16):                  # odd widths: the stand-alone adjoint
                dxu = ops.upfirdn2d(dxt, torch.flip(fir, [0, 1]), up=1, down=2, pad=(1, 1))
                dx, ds = ops.modconv_dx_finish(dxu, x, s, False, need_dx=need_dx)
            else:
                dx, ds = ops.modconv_dx_finish(dxt, x, s, upsample, need_dx=need_dx)
        dw_aside = False
        if need_dw:
            if ops.wgrad_mod_supported(B, Cin, Cout, H, W, upsample) and not (x.data_ptr() % 16 or dt.data_ptr() % 16):
                def launch():
                    t = ops.conv2d_wgrad(dt, x, Cout, Cin, 3, 1, upsample=upsample, up_fir=True, scale=scale, batch_scale=s, g_scale=dprime)
                    if dw_aside and dd is not None:      # the demodulation adjoint's share of dw, behind the kernel on ITS stream
                        ops.modconv_demod_bwd(weight, s, d, dd, scale, ds=None, dw=t)
                    return t
                # behind a WeightGateFn (which joins the second stream before the parameter sees the gradient): beside the data gradient
                dw_aside = getattr(weight, "_spk_gate_of", None) is not None
                # (s and d are saved tensors: autograd releases them when this backward returns, the second stream may still be reading)
                dw = _wgrad_aside(dt, x, launch, extra=[t for t in (s, d, dprime, dd) if t is not None]) if dw_aside else launch()
            else:                # tiny planes (the 4^2 layer): rescale the two (few-KB) operands and run the plain kernel
                x_up = ops.upfirdn2d(x, fir, up=2, down=1, pad=(2, 1)) if upsample else x
                g2 = ops.bias_noise_style(dt, B, None, None, None, _scale_rows(dprime))
                xs = ops.bias_noise_style(x_up, B, None, None, None, _scale_rows(s))
                dw = ops.conv2d_wgrad(g2, xs, Cout, Cin, 3, 1, scale=scale)
        if dd is not None and (need_ds or (need_dw and not dw_aside)):
            if need_ds and ds is None:
                ds = torch.zeros_like(s)
            ops.modconv_demod_bwd(weight, s, d, dd, scale, ds=ds if need_ds else None, dw=dw if (need_dw and not dw_aside) else None)
        return dx, dw, ds, dbias, dnw, None, None, None, None, None, None, None, None, None


class ModToRGBFn(torch.autograd.Function):
    """y = scale * conv1x1(x * s[b,ci], w) + bias [+ upfirdn2d_up2(skip)] to <= 4 channels (StyleGAN2 toRGB, no
    demodulation; the skip image's x2 upsample + add ride in the same launch).  Backward: one streaming pass for dx (the
    modulation folded into the LDS weights), one for the per-image sums P[b,o,c] = scale * <dy[b,o], x[b,c]> -- from which the
    weight and modulation gradients are [B,3,C]-sized contractions -- and the upfirdn2d adjoint for the skip image."""

    @staticmethod
    def forward(ctx, x, weight, s, bias, scale, skip, fir, grad_mode):
        y = ops.conv1x1_small_mod(x, weight, s, bias, in_scale=scale, skip=skip)
        if _needs(ctx, grad_mode):
            ctx.save_for_backward(x, weight, s)
            ctx.conf = (scale, bias is not None, skip is not None, fir)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, s = ctx.saved_tensors
        scale, has_bias, has_skip, fir = ctx.conf
        dy = dy.contiguous()
        dx, P, db = ops.torgb_mod_bwd(x, weight, s.contiguous(), dy, in_scale=scale, need_dx=ctx.needs_input_grad[0])
        O, Cc = weight.shape[:2]
        # ([B,3,C]-sized: elementwise products + sums, not a library GEMM)
        dw = (P * s.unsqueeze(1)).sum(0).view(O, Cc, 1, 1) if ctx.needs_input_grad[1] else None
        ds = (P * weight.view(1, O, Cc)).sum(1) if ctx.needs_input_grad[2] else None
        dskip = None
        if has_skip and ctx.needs_input_grad[5]:      # adjoint of upfirdn2d(up=2, pad (2,1)): down=2 with the flipped FIR, pad (1,1)
            dskip = ops.upfirdn2d(dy, torch.flip(fir, [0, 1]), up=1, down=2, pad=(1, 1))
        return dx, dw, ds, (db if has_bias else None), None, dskip, None, None


class UpFirDnFn(torch.autograd.Function):
    """upfirdn2d with a host FIR; the adjoint is upfirdn2d with up/down swapped and the flipped FIR."""

    @staticmethod
    def forward(ctx, x, fir, up, down, pad):
        ctx.conf = (fir, up, down, pad, x.shape[-2:])
        return ops.upfirdn2d(x, fir, up=up, down=down, pad=pad)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        fir, up, down, pad, (h, w) = ctx.conf
        k = fir.shape[0]
        ho, wo = g.shape[-2:]
        assert h == w and ho == wo, "square images only"
        gp0 = k - pad[0] - 1
        gp1 = h * up - ho * down + pad[0] - up + 1
        return ops.upfirdn2d(g.contiguous(), torch.flip(fir, [0, 1]), up=down, down=up, pad=(gp0, gp1)), None, None, None, None


class InstanceNormAffineFn(torch.autograd.Function):
    """AdaIN (stylegan.py:84-95): InstanceNorm2d(x) * scale[b,c] + bias[b,c]."""

    @staticmethod
    def forward(ctx, x, scale, bias, eps, grad_mode):
        y = ops.instance_norm_affine(x, scale, bias, eps)
        if _needs(ctx, grad_mode):
            ctx.save_for_backward(x, scale)
            ctx.eps = eps
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, scale = ctx.saved_tensors
        dx, dscale, dbias = ops.instance_norm_affine_bwd(x, dy.contiguous(), scale, ctx.eps, need_dx=ctx.needs_input_grad[0])
        return dx, (dscale if ctx.needs_input_grad[1] else None), (dbias if ctx.needs_input_grad[2] else None), None, None


class Upsample2xFn(torch.autograd.Function):
    """F.interpolate(scale_factor=2, mode="bilinear") and its adjoint."""

    @staticmethod
    def forward(ctx, x):
        return ops.upsample2x_bilinear(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        return ops.upsample2x_bilinear_bwd(dy.contiguous())


class SpectralNormAllFn(torch.autograd.Function):
    """W_hat_l = W_l / sigma_l for every spectral-norm wrapped layer of a module in one grouped call (the pre-forward hook of
    torch.nn.utils.spectral_norm, styleganv1.py:644-672): in training mode one power iteration updates the layers' u / v
    buffers in place first.  Backward = autograd of W / sigma with u, v constant (as the hook has it, under ``no_grad``):
    dW = (G - <G, W_hat> u v^T) / sigma, with the u, v, sigma of THIS call (a discriminator runs several forwards before one
    backward, train.py:160-182; torch clones the vectors for the same reason)."""

    @staticmethod
    def forward(ctx, bufs, power_iteration, eps, *weights):
        us, vs = [b[0] for b in bufs], [b[1] for b in bufs]
        hats, sigma = ops.spectral_norm_grouped([w.detach() for w in weights], us, vs, power_iteration, eps)
        inv = sigma.reciprocal()
        for i, (h, w) in enumerate(zip(hats, weights)):
            if w.dim() == 4:                  # conv layers: weight_orig's packed image + 1 / sigma in the epilogue (see _packed)
                h._spk_sn = (w, inv[i:i + 1])
        if any(ctx.needs_input_grad[3:]):
            ru, cv = [u.numel() for u in us], [v.numel() for v in vs]
            ctx.save_for_backward(torch.cat([u.reshape(-1) for u in us]), torch.cat([v.reshape(-1) for v in vs]), sigma, *weights)
            ctx.sizes = (ru, cv)
            ctx.hats = [weakref.ref(h) for h in hats]
            ctx.params = weights              # (the Parameters themselves: their gradients are shared across passes)
        return tuple(hats)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        u_flat, v_flat, sigma, *weights = ctx.saved_tensors
        _side_join(sigma.device)              # the convs' weight gradients were queued on the second stream
        for ref in ctx.hats:                  # _wgrad_shared's notes: consumed here, must not pin a gradient past this pass
            h = ref()
            if h is not None and getattr(h, "_spk_dw", None) is not None:
                h._spk_dw = None
        ru, cv = ctx.sizes
        us, vs = list(u_flat.split(ru)), list(v_flat.split(cv))
        gs = [None if (g is None or not ctx.needs_input_grad[3 + i]) else g.contiguous() for i, g in enumerate(grads)]
        # the passes of one backward share ONE gradient tensor per weight_orig (see _first_or_add): the first node's goes to autograd,
        # the others accumulate into it inside the kernel
        task = torch._C._current_graph_task_id()
        into = []
        for w, g in zip(ctx.params, gs):
            held = getattr(w, "_spk_sn_dw", None) if (g is not None and isinstance(w, torch.nn.Parameter)) else None
            into.append(held[1] if (held is not None and held[0] == task) else None)
        dws = ops.spectral_norm_grouped_bwd(gs, weights, us, vs, sigma, into=into)
        for w, t in zip(ctx.params, dws):
            if t is not None and isinstance(w, torch.nn.Parameter):
                w._spk_sn_dw = (task, t.detach())     # (an alias: see _first_or_add)
        return (None, None, None) + tuple(dws)


def spectral_norm_all(modules, training, eps=1e-12):
    """Normalised weights of torch ``spectral_norm``-wrapped ``modules`` (attributes ``weight_orig`` / ``weight_u`` /
    ``weight_v``), in order."""
    bufs = [(m.weight_u, m.weight_v) for m in modules]
    return SpectralNormAllFn.apply(bufs, bool(training), eps, *[m.weight_orig for m in modules])


class PixelNormFn(torch.autograd.Function):
    """x * rsqrt(mean_c x^2 + eps) -- styleganv1.py:132-136 / stylegan.py:28-29."""

    @staticmethod
    def forward(ctx, x, eps, sqrt_form):
        ctx.save_for_backward(x)
        ctx.eps = eps
        return ops.pixelnorm(x, eps, sqrt_form)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.pixelnorm_bwd(x, dy.contiguous(), ctx.eps), None, None


class Blur2dFn(torch.autograd.Function):
    """Depthwise FIR (host filter), zero pad (k-1)/2, stride 1 or 2 -- styleganv1.py:52-63."""

    @staticmethod
    def forward(ctx, x, filt, stride):
        ctx.conf = (filt, stride, tuple(x.shape[-2:]))
        return ops.blur2d(x, filt, stride)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        filt, stride, hw = ctx.conf
        return ops.blur2d_bwd(dy.contiguous(), filt, stride, hw), None, None


class Upscale2dFn(torch.autograd.Function):
    """Nearest-neighbour repeat with a gain -- styleganv1.py:113-120."""

    @staticmethod
    def forward(ctx, x, factor, gain):
        ctx.conf = (factor, gain)
        return ops.upscale2d_nearest(x, factor, gain)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        factor, gain = ctx.conf
        return ops.upscale2d_nearest_bwd(dy.contiguous(), factor, gain), None, None


class FusedUpscaleFn(torch.autograd.Function):
    """y = conv_transpose2d(x, w[Cin,Cout,4,4], bias, stride 2, padding 1) -- ``GBlock.up_sample`` for res >= 7,
    styleganv1.py:231,258.  Forward: four output-parity 2x2 MFMA kernels in one launch.  Backward: the transposed conv's
    adjoint is the plain 4x4 stride-2 pad-1 conv with the SAME parameter read as [out = Cin][in = Cout] (dx, on the MFMA conv
    kernel), the weight gradient is that conv's weight gradient with the roles of input and output gradient exchanged
    (dw[ci,co,ky,kx] = sum x[ci,m,n] dy[co,2m+ky-1,2n+kx-1], on the MFMA wgrad kernel, one tap row per workgroup), the bias
    gradient the plane sums of dy."""

    @staticmethod
    def forward(ctx, x, weight, bias, packed, grad_mode):
        y = ops.conv_transpose4x4_s2(x, weight, bias, packed)
        if _needs(ctx, grad_mode):
            ctx.save_for_backward(x, weight)
            ctx.conf = (packed, bias is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        packed, has_bias = ctx.conf
        B, Cin, H, W = x.shape
        Cout = weight.shape[1]
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            cfg = ops.conv2d_pick_config(4, 2, B, Cout, Cin, H, W)
            dx = ops.conv2d_fused(dy, packed.get(weight, cfg), Cin, 4, 2, config=cfg)
        if ctx.needs_input_grad[1]:
            dw = ops.conv2d_wgrad(x, dy, Cin, Cout, 4, 2)
        if has_bias and ctx.needs_input_grad[2]:
            db = ops.epilogue_bwd(dy, None, None, None, 1.0)[1][:, 2].sum(0)
        return dx, dw, db, None, None


def fused_upscale(x, weight, bias, packed):
    return FusedUpscaleFn.apply(x, weight, bias, packed, torch.is_grad_enabled())


# call-site spellings: sample the grad mode where it is still visible
def fused_conv(x, weight, bias, noise_w, noise, style, upsample, slope, packed, w_scale=1.0):
    return FusedConvFn.apply(x, weight, bias, noise_w, noise, style, upsample, slope, packed, torch.is_grad_enabled(), w_scale)


def fc(x, weight, bias, wmul, bmul, slope):
    return FCFn.apply(x, weight, bias, wmul, bmul, slope, torch.is_grad_enabled())


def to_rgb(x, weight, bias):
    return ToRGBFn.apply(x, weight, bias, torch.is_grad_enabled())


def bias_noise_style(x, bias, noise_w, noise, style, B):
    return BiasNoiseStyleFn.apply(x, bias, noise_w, noise, style, B, torch.is_grad_enabled())


def conv_bias_lrelu(x, weight, bias, k, stride, slope):
    return ConvBiasLReLUFn.apply(x, weight, bias, k, stride, slope, torch.is_grad_enabled())


def global_avgpool(x):
    return GlobalAvgPoolFn.apply(x)


def mod_conv(x, weight, s, bias, noise_w, noise, scale, upsample, slope, gain, fir, packed, demodulate=True):
    return ModConvFn.apply(x, weight, s, bias, noise_w, noise, scale, upsample, slope, gain, fir, packed, demodulate,
                           torch.is_grad_enabled())


def mod_to_rgb(x, weight, s, bias, scale, skip=None, fir=None):
    return ModToRGBFn.apply(x, weight, s, bias, scale, skip, fir, torch.is_grad_enabled())


def upfirdn(x, fir, up, down, pad):
    return UpFirDnFn.apply(x, fir, up, down, pad)


def instance_norm_affine(x, scale, bias, eps):
    return InstanceNormAffineFn.apply(x, scale, bias, eps, torch.is_grad_enabled())


def upsample2x(x):
    return Upsample2xFn.apply(x)


def pixelnorm(x, eps, sqrt_form=False):
    return PixelNormFn.apply(x, eps, sqrt_form)


def blur2d(x, filt, stride):
    return Blur2dFn.apply(x, filt, stride)


def upscale2d(x, factor, gain):
    return Upscale2dFn.apply(x, factor, gain)
