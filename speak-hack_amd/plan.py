"""Launch plans: a module's inference forward as ONE call across the C boundary (include/spk.h, ``spk_launch_list``).

The reference's callers run a decoder pass as one Python call -- ``self.Gd(gen_input)`` at model.py:113-114,
``SynthesisNetwork.forward`` at styleganv1.py:593-610.  Behind that call sit ~25 kernel launches whose descriptors depend
only on (module, batch size, device): tile configs, packed weights, split-K scratch, the intermediate activations.  Issued
one by one, every launch rebuilt its 35-field descriptor and re-queried config / workspace in Python (~50 us each: 5.7 ms
per B=8 step against 3.8 ms of GPU work).  A plan builds all of that ONCE and keeps it; a call patches the handful of
pointers that change (input, explicit noise, the fresh output tensor) and hands the array to ``spk_launch_list`` -- the
same kernels with the same arguments as the one-by-one path.

What a plan owns: the packed conv weights (re-packed in place when a parameter's version counter moves, e.g. after an
optimizer step or ``load_state_dict``), two ping-pong activation buffers (each layer's output is read only by the next
launch, all on one stream), the style / modulation vectors, the noise buffer (one device draw per call, as
``SynthesisNetwork.forward`` does) and the split-K scratch.  The RESULT is a fresh tensor per call, so callers may keep
outputs of earlier calls.  Plans are per (batch, device, stream) and used only when no gradient is required; training
goes through the autograd Functions.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import ops

LRELU = 0.2
FUSE_TORGB = os.environ.get("SPK_FUSE_TORGB", "1") != "0"     # toRGB inside the last Winograd launch's epilogue (SPK_EPI_TORGB)
MAX_PLANS = 4        # per module: (batch, device, stream, entry) combinations kept


def _sig(params):
    return tuple((p.data_ptr(), p._version) for p in params)


class LaunchPlan:
    def __init__(self, device):
        self.device = device
        self.ops = []            # (kind, ctypes struct)
        self.keep = []           # tensors / ctypes arrays the descriptors point into
        self._array = None
        self._refreshers = []    # callables re-deriving plan-owned copies of parameters (packed weights, expanded constants)
        self._params = []
        self._where = []         # per tracked parameter: (module, attribute name) it was found under at build time
        self._stamp = None
        self.captured = False    # a hipGraph recorded this plan's launches: its buffers must outlive the plan cache
        self.ws = None
        self._ws_bytes = 0
        self._up_scratch, self._up_users = None, []   # the x2 image of the Winograd x2 layers (one scratch buffer) and its users

    # ---- building ----------------------------------------------------------------------------------
    def buf(self, *shape):
        t = torch.empty(shape, device=self.device, dtype=torch.float32)
        self.keep.append(t)
        return t

    def add(self, kind, desc):
        self.ops.append((kind, desc))
        return desc

    def track(self, *params):
        self._params.extend(params)

    # layers with at least this many output pixels (batch x H x W) take the split-precision path when the plan asks for it:
    # below it the 64co x 256px blocks cannot fill 256 CUs without a split-K the bf16x3 kernel does not have (B = 8: the
    # 16^2 layers up -- 93-96 us against 104-109 for the f32 kernel there, 83-88 against 40-43 at 8^2)
    BF16X3_MIN_PIXELS = 2048

    def conv(self, x, weight, Cout, *, bias=None, noise_w=None, noise=None, style=None, upsample=False, up_fir=False, slope=None,
             out, out_scale=1.0, batch_scale=None, demod=None, act_gain=1.0, precision="f32"):
        """One fused 3x3 stride-1 conv launch (the descriptor ``ops.conv2d_fused`` would build), weights packed here.
        ``precision`` "bf16x3": the opt-in split-precision kernel (include/spk.h SPK_CONV_BF16X3) where it serves the shape."""
        B, Cin, Hs, Ws = x.shape
        H, W = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
        # (bf16x3 where it is the FASTER form: a <= 16^2 layer runs quicker -- and exactly -- on the sliced fp32 Winograd kernel)
        if (precision == "bf16x3" and B * H * W >= self.BF16X3_MIN_PIXELS and ops.bf16x3_supported(B, Cin, Cout, H, W)
                and not (H * W <= 256 and ops.use_wino(B, Cin, Cout, H, W))):
            packed = torch.empty(L.lib().spk_conv2d_packed_bytes_bf16x3(Cin, Cout), device=self.device, dtype=torch.uint8)
            self.keep.append(packed)
            self._refreshers.append(lambda w=weight, p=packed: ops.pack_conv_weight_bf16x3(w.detach(), out=p))
            self.track(weight)
            flags = L.CONV_BF16X3 | (L.EPI_BIAS if bias is not None else 0) | (L.EPI_NOISE if noise is not None else 0) | \
                (L.EPI_LRELU if slope is not None else 0) | (L.EPI_STYLE if style is not None else 0) | \
                (L.CONV_UPSAMPLE2X if upsample else 0) | (L.CONV_UP_FIR1331 if (upsample and up_fir) else 0) | \
                (L.CONV_IN_BATCH_SCALE if batch_scale is not None else 0)
            d = L.Conv2dDesc(x=x.data_ptr(), w_packed=packed.data_ptr(), bias=L.dptr(bias, "bias"),
                             noise_w=L.dptr(noise_w, "noise_w") if noise is not None else None,
                             noise=noise.data_ptr() if noise is not None else None,
                             style=style.data_ptr() if style is not None else None,
                             in_scale=batch_scale.data_ptr() if batch_scale is not None else None, in_shift=None,
                             out_scale_bc=demod.data_ptr() if demod is not None else None, act_gain=float(act_gain), stats=None,
                             y=out.data_ptr(), y_pre=None, B=B, Cin=Cin, Cout=Cout, H=H, W=W, Hin=Hs, Win=Ws, kh=3, kw=3, stride=1,
                             style_stride=int(style.stride(0)) if style is not None else 0, flags=flags,
                             lrelu_slope=float(slope if slope is not None else 1.0), out_scale=float(out_scale), config=-1,
                             ksplit=1, workspace=None, workspace_bytes=0, groups=1, group_in_stride=0, stats_slots=0)
            return self.add(L.OP_CONV2D, d)
        if ops.use_wino(B, Cin, Cout, H, W) and (not upsample or Ws % 4 == 0):
            # fp32 Winograd F(2x2, 3x3) (include/spk.h SPK_CONV_WINOGRAD); a x2 layer first writes its upsampled input (one
            # HBM-bound launch: the separate nn.Upsample of styleganv1.py:621,624) -- the transform would cost more inside the
            # MFMA kernel than this pass does beside it
            if upsample:
                need = B * Cin * H * W                    # one scratch image for all x2 layers: launches are stream-ordered
                if self._up_scratch is None or self._up_scratch.numel() < need:
                    self._up_scratch = self.buf(need)
                    for a_, n_ in self._up_users:           # earlier, smaller users move into the larger buffer
                        a_.y = self._up_scratch.data_ptr()
                        n_.x = self._up_scratch.data_ptr()
                xu = self._up_scratch[:need].view(B, Cin, H, W)
                up_op = self.add(L.OP_UPSAMPLE2X, L.Upsample2xArgs(x=x.data_ptr(), y=xu.data_ptr(), planes=B * Cin, Hin=Hs, Win=Ws,
                                                                   zero_border=1 if up_fir else 0))
                x = xu
            packed = self.buf(L.lib().spk_conv2d_packed_bytes_wino(Cin, Cout) // 4)
            self._refreshers.append(lambda w=weight, p=packed: ops.pack_conv_weight_wino(w.detach(), out=p))
            self.track(weight)
            flags = L.CONV_WINOGRAD | (L.EPI_BIAS if bias is not None else 0) | (L.EPI_NOISE if noise is not None else 0) | \
                (L.EPI_LRELU if slope is not None else 0) | (L.EPI_STYLE if style is not None else 0) | \
                (L.CONV_IN_BATCH_SCALE if batch_scale is not None else 0)
            d = L.Conv2dDesc(x=x.data_ptr(), w_packed=packed.data_ptr(), bias=L.dptr(bias, "bias"),
                             noise_w=L.dptr(noise_w, "noise_w") if noise is not None else None,
                             noise=noise.data_ptr() if noise is not None else None,
                             style=style.data_ptr() if style is not None else None,
                             in_scale=batch_scale.data_ptr() if batch_scale is not None else None, in_shift=None,
                             out_scale_bc=demod.data_ptr() if demod is not None else None, act_gain=float(act_gain), stats=None,
                             y=out.data_ptr(), y_pre=None, B=B, Cin=Cin, Cout=Cout, H=H, W=W, Hin=H, Win=W, kh=3, kw=3, stride=1,
                             style_stride=int(style.stride(0)) if style is not None else 0, flags=flags,
                             lrelu_slope=float(slope if slope is not None else 1.0), out_scale=float(out_scale), config=-1,
                             ksplit=0, workspace=None, workspace_bytes=0, groups=1, group_in_stride=0, stats_slots=0)
            # (few regions: the contraction runs in slices through the plan's split-K workspace, spk_conv2d_wino_ksplit)
            self._ws_bytes = max(self._ws_bytes, L.lib().spk_conv2d_wino_workspace_bytes(0, B, Cin, Cout, H, W))
            if upsample:
                self._up_users.append((up_op, d))
            return self.add(L.OP_CONV2D, d)
        cfg = ops.conv2d_pick_config(3, 1, B, Cin, Cout, H, W)
        if batch_scale is not None and cfg < 4:
            cfg += 4
        n = L.lib().spk_conv2d_packed_floats(cfg, 3, 3, Cin, Cout)
        packed = self.buf(n)
        self._refreshers.append(lambda w=weight, p=packed, c=cfg: ops.pack_conv_weight(w.detach(), c, out=p))
        self.track(weight)
        ws_bytes = L.lib().spk_conv2d_workspace_bytes_grouped(int(cfg), 0, 3, 3, 1, B, Cin, Cout, H, W, 1)
        if ws_bytes < 0:
            raise L.SpkError(f"plan: config {cfg} cannot host 3x3 shape {(B, Cin, Cout, H, W)}")
        self._ws_bytes = max(self._ws_bytes, ws_bytes)
        flags = (L.EPI_BIAS if bias is not None else 0) | (L.EPI_NOISE if noise is not None else 0) | \
                (L.EPI_LRELU if slope is not None else 0) | (L.EPI_STYLE if style is not None else 0) | \
                (L.CONV_UPSAMPLE2X if upsample else 0) | (L.CONV_UP_FIR1331 if (upsample and up_fir) else 0) | \
                (L.CONV_IN_BATCH_SCALE if batch_scale is not None else 0)
        d = L.Conv2dDesc(x=x.data_ptr(), w_packed=packed.data_ptr(), bias=L.dptr(bias, "bias"),
                         noise_w=L.dptr(noise_w, "noise_w") if noise is not None else None,
                         noise=noise.data_ptr() if noise is not None else None,
                         style=style.data_ptr() if style is not None else None,
                         in_scale=batch_scale.data_ptr() if batch_scale is not None else None, in_shift=None,
                         out_scale_bc=demod.data_ptr() if demod is not None else None, act_gain=float(act_gain), stats=None,
                         y=out.data_ptr(), y_pre=None, B=B, Cin=Cin, Cout=Cout, H=H, W=W, Hin=Hs, Win=Ws, kh=3, kw=3, stride=1,
                         style_stride=int(style.stride(0)) if style is not None else 0, flags=flags,
                         lrelu_slope=float(slope if slope is not None else 1.0), out_scale=float(out_scale), config=int(cfg),
                         ksplit=0, workspace=None, workspace_bytes=0, groups=1, group_in_stride=0, stats_slots=0)
        return self.add(L.OP_CONV2D, d)

    def fc(self, x, lin_weight, lin_bias, wmul, bmul, slope, out):
        self.track(lin_weight, *([lin_bias] if lin_bias is not None else []))
        O, I = lin_weight.shape
        return self.add(L.OP_FC, L.FcArgs(x=x.data_ptr(), x_stride=x.stride(0), w=L.dptr(lin_weight, "weight"),
                                           bias=L.dptr(lin_bias, "bias"), out=out.data_ptr(), out_stride=out.stride(0),
                                           B=x.shape[0], I=I, O=O, wmul=float(wmul), bmul=float(bmul), slope=float(slope)))

    def fc_grouped(self, items, B):
        """``items``: (x tensor or None (patched per call), x_stride, weight, bias, wmul, bmul, slope, out)."""
        arr = (L.FcGroup * len(items))()
        for g, (x, xs, weight, bias, wmul, bmul, slope, out) in zip(arr, items):
            self.track(weight, *([bias] if bias is not None else []))
            O, I = weight.shape
            g.x, g.x_stride, g.w, g.bias = (x.data_ptr() if x is not None else None), xs, L.dptr(weight, "weight"), L.dptr(bias, "bias")
            g.out, g.out_stride, g.I, g.O = out.data_ptr(), out.stride(0), I, O
            g.wmul, g.bmul, g.slope = float(wmul), float(bmul), float(slope)
        self.keep.append(arr)
        self.add(L.OP_FC_GROUPED, L.FcGroupedArgs(groups=C.addressof(arr), n_groups=len(items), B=B))
        return arr

    def finish(self, *owners):
        """``owners``: the modules whose parameters the plan reads -- every tracked parameter is looked up in them so that
        ``valid_for`` can tell a re-assigned parameter (``m.weight = nn.Parameter(...)``, ``load_state_dict(assign=True)``, a
        parametrization swap) from the one whose storage the descriptors point at."""
        loc = {}
        for owner in owners:
            for m in owner.modules():
                for n, q in m._parameters.items():
                    if q is not None:
                        loc.setdefault(id(q), (m, n))
        self._where = [loc.get(id(q)) for q in self._params]
        if self._ws_bytes:
            self.ws = self.buf((self._ws_bytes + 3) // 4)
            for kind, d in self.ops:
                if kind == L.OP_CONV2D:
                    d.workspace, d.workspace_bytes = self.ws.data_ptr(), self.ws.numel() * 4
        self._array = (L.Op * len(self.ops))()
        for slot, (kind, d) in zip(self._array, self.ops):
            slot.kind, slot.desc = kind, C.addressof(d)
        self.refresh(force=True)

    # ---- per call ----------------------------------------------------------------------------------
    def refresh(self, force=False):
        """Parameters are read by pointer; what the plan DERIVED from them (packed weights, expanded constants) is rebuilt
        in place when any tracked parameter's version counter moved."""
        stamp = _sig(self._params)
        if force or stamp != self._stamp:
            with torch.no_grad():
                for f in self._refreshers:
                    f()
            self._stamp = stamp

    def valid_for(self):
        """Pointers baked into the descriptors must still be the storage of the parameters the modules hold NOW: the
        tracked Parameter object is still the one registered under its module attribute, and it has not moved."""
        if self._stamp is None:
            return False
        for a, q, w in zip(self._stamp, self._params, self._where):
            if a[0] != q.data_ptr() or (w is not None and w[0]._parameters.get(w[1]) is not q):
                return False
        return True

    def launch(self, kind_mask=L.ALL_OPS):
        L.check(L.lib().spk_launch_list(C.addressof(self._array), len(self.ops), int(kind_mask) & 0xFFFFFFFF, L.stream_ptr()),
                "spk_launch_list")


# ---------------------------------------------------------------------------------------------------------------------
class DecoderPlan(LaunchPlan):
    """``SynthesisNetwork.forward`` (styleganv1.py:593-610), optionally preceded by ``StyleGenerator``'s mapping stack and
    truncation (styleganv1.py:528-543): 8 FC launches (with the mapping), one grouped launch for the 13+ style affines,
    the constant prologue, 2 fused conv launches per block, toRGB."""

    def __init__(self, synthesis, B, device, generator=None, precision="f32"):
        super().__init__(device)
        s = self.synthesis = synthesis
        self.precision = precision
        self.B, self.with_mapping = B, generator is not None
        mods = [s.style_mod] + [m for layer in s.layers for m in (layer.style_mod1, layer.style_mod2)]
        if len(mods) > L.FC_MAX_GROUPS:
            raise L.SpkError("DecoderPlan: too many style affines for one grouped launch")
        # ---- mapping stack: w512 [B,512]; the truncation scale of rows < cutoff rides on the style FCs' multiplier ----
        psi = [1.0] * len(mods)
        self.feat_fc = None
        if generator is not None:
            w_in = None
            for i, fc in enumerate(generator.mapping):
                out = self.buf(B, fc.weight.shape[0])
                if i == 0:      # input pointer patched per call
                    self.feat_fc = self.add(L.OP_FC, L.FcArgs(x=None, x_stride=fc.weight.shape[1], w=L.dptr(fc.weight, "weight"),
                                                              bias=L.dptr(fc.bias, "bias"), out=out.data_ptr(), out_stride=out.stride(0),
                                                              B=B, I=fc.weight.shape[1], O=fc.weight.shape[0], wmul=float(fc.w_lrmul),
                                                              bmul=float(fc.b_lrmul), slope=LRELU))
                    self.track(fc.weight, fc.bias)
                else:
                    self.fc(w_in, fc.weight, fc.bias, fc.w_lrmul, fc.b_lrmul, LRELU, out)
                w_in = out
            self.w512 = w_in
            if generator.truncation_psi and generator.truncation_cutoff:
                psi = [generator.truncation_psi if j < generator.truncation_cutoff else 1.0 for j in range(len(mods))]
        # ---- the style affines: one grouped launch ----
        self.styles = [self.buf(B, m.linear.weight.shape[0]) for m in mods]
        items = []
        for j, (m, st) in enumerate(zip(mods, self.styles)):
            lin = m.linear
            if generator is not None:      # every row of the broadcast dlatents is w512 (scaled by psi_j)
                items.append((self.w512, self.w512.stride(0), lin.weight, lin.bias, lin.w_lrmul * psi[j], lin.b_lrmul, LRELU, st))
            else:                          # rows of the caller's w [B,L,512]: patched per call
                items.append((None, 0, lin.weight, lin.bias, lin.w_lrmul, lin.b_lrmul, LRELU, st))
        self.style_groups = self.fc_grouped(items, B)
        # ---- noise: one flat buffer, one device draw per call ----
        shapes = s.noise_shapes(B)
        sizes = [sh[0] * sh[2] * sh[3] for sh in shapes]
        self.noise_flat = self.buf(sum(sizes))
        self.noise_views = [t.view(sh) for t, sh in zip(self.noise_flat.split(sizes), shapes)]
        # ---- prologue (styleganv1.py:596-599) ----
        C0 = s.const_input.shape[1]
        pp = [self.buf(B * max(self._act_floats(s)))  for _ in range(2)]
        x = pp[0][:B * C0 * 16].view(B, C0, 4, 4)
        self.track(s.const_input, s.bias, s.noise_input1.weight)
        self.noise_ops = []          # (descriptor, field) whose noise pointer follows an explicit-noise argument
        d = self.add(L.OP_BIAS_NOISE_STYLE, L.BiasNoiseStyleArgs(
            x=L.dptr(s.const_input, "const_input"), x_batch_stride=0 if B > 1 else C0 * 16, bias=L.dptr(s.bias, "bias"),
            noise_w=L.dptr(s.noise_input1.weight, "noise_w"), noise=self.noise_views[0].data_ptr(), style=self.styles[0].data_ptr(),
            style_stride=self.styles[0].stride(0), y=x.data_ptr(), B=B, C=C0, HW=16))
        self.noise_ops.append(d)
        # ---- blocks ----
        cur = 0
        for i, layer in enumerate(s.layers):
            for half, (conv, nmod) in enumerate(((layer.conv1, layer.noise1), (layer.conv2, layer.noise2))):
                Cout = conv.weight.shape[0]
                up = half == 0
                H = x.shape[2] * (2 if up else 1)
                y = pp[1 - cur][:B * Cout * H * H].view(B, Cout, H, H)
                k = 1 + 2 * i + half
                self.track(conv.bias, nmod.weight)
                # the LAST conv (64 channels at full resolution): fp32 Winograd with toRGB in its epilogue beats the bf16x3 kernel + a toRGB pass
                last_conv = i == len(s.layers) - 1 and half == 1 and FUSE_TORGB and Cout <= 64 and s.to_rgb.weight.shape[0] == 3 and \
                    ops.use_wino(B, x.shape[1], Cout, H, H) and ops.wino_ksplit(B, x.shape[1], Cout, H, H) == 1
                d = self.conv(x, conv.weight, Cout, bias=conv.bias, noise_w=nmod.weight, noise=self.noise_views[k],
                              style=self.styles[k], upsample=up, slope=LRELU, out=y, precision="f32" if last_conv else precision)
                self.noise_ops.append(d)
                x, cur = y, 1 - cur
        # ---- toRGB (styleganv1.py:607) ----
        self.track(s.to_rgb.weight, s.to_rgb.bias)
        O, Cc = s.to_rgb.weight.shape[:2]
        self.out_shape = (B, O, x.shape[2], x.shape[3])
        last = d                    # the last block's conv2 launch
        self.rgb_fused = None
        if (FUSE_TORGB and O == 3 and Cc <= 64 and (last.flags & L.CONV_WINOGRAD)
                and ops.wino_ksplit(B, last.Cin, Cc, x.shape[2], x.shape[3]) == 1):
            # the 1x1 rides in that launch's epilogue (SPK_EPI_TORGB): the last activation is neither written nor read back
            last.flags |= L.EPI_TORGB
            last.rgb_w, last.rgb_bias, last.rgb_channels = L.dptr(s.to_rgb.weight, "weight"), L.dptr(s.to_rgb.bias, "bias"), 3
            last.y, last.ksplit = None, 1
            self.rgb_fused = last
        else:
            self.torgb = self.add(L.OP_TORGB, L.ToRGBArgs(x=x.data_ptr(), w=L.dptr(s.to_rgb.weight, "weight"), mod=None,
                                                          bias=L.dptr(s.to_rgb.bias, "bias"), skip=None, y=None, B=B, C=Cc, O=O,
                                                          H=x.shape[2], W=x.shape[3], in_scale=1.0))
        self.finish(*([synthesis] + ([generator] if generator is not None else [])))

    @staticmethod
    def _act_floats(s):
        """Per-sample float counts of every activation the ping-pong buffers must hold."""
        sizes = [s.const_input.shape[1] * 16]
        res = 4
        for layer in s.layers:
            res *= 2
            sizes.append(layer.out_channels * res * res)
        return sizes

    def run(self, x, noises=None, kind_mask=L.ALL_OPS):
        """``x``: features [B,input_dim] (plan built with the generator) or dlatents w [B,L,512]."""
        self.refresh()
        self.captured = self.captured or torch.cuda.is_current_stream_capturing()
        if self.with_mapping:
            if x.dim() != 2 or x.stride(1) != 1:
                raise L.SpkError("DecoderPlan: features must be [B,input_dim] with unit inner stride")
            if not x.is_cuda or x.dtype != torch.float32:
                raise L.SpkError(f"features: expected a float32 HIP tensor, got {x.dtype} on {x.device} (no CPU path)")
            self.feat_fc.x, self.feat_fc.x_stride = x.data_ptr(), x.stride(0)
        else:
            if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.size(1) < len(self.styles):
                raise L.SpkError("DecoderPlan: w must be a contiguous float32 HIP tensor [B, >= num style layers, 512]")
            base, row, bs = x.data_ptr(), x.stride(1) * 4, x.stride(0)
            for j, g in enumerate(self.style_groups):
                g.x, g.x_stride = base + j * row, bs
        if noises is None:
            self.noise_flat.normal_()
            for d, nv in zip(self.noise_ops, self.noise_views):
                d.noise = nv.data_ptr()
        else:
            if len(noises) != len(self.noise_ops):
                raise ValueError(f"expected {len(self.noise_ops)} noise tensors, got {len(noises)}")
            for d, nz, nv in zip(self.noise_ops, noises, self.noise_views):
                if nz.numel() != nv.numel():
                    raise L.SpkError(f"noise must be {tuple(nv.shape)}, got {tuple(nz.shape)}")
                d.noise = L.dptr(nz, "noise")
        y = torch.empty(self.out_shape, device=self.device, dtype=torch.float32)
        if self.rgb_fused is not None:
            self.rgb_fused.rgb_y = y.data_ptr()
        else:
            self.torgb.y = y.data_ptr()
        self.launch(kind_mask)
        return y


_retired = []        # plans a hipGraph has recorded, dropped from their cache: kept alive (the graph holds raw addresses)


def _drop(plan):
    if plan is not None and plan.captured:
        _retired.append(plan)


def plan_for(owner, key, build):
    """Per-module plan cache (``owner.__dict__['_plans']``): at most MAX_PLANS entries, least recently used dropped; an
    entry whose baked parameter pointers went stale (``.to()``, re-assigned parameters) is rebuilt.  A plan whose launches
    were recorded into a hipGraph is never freed when it leaves the cache -- its packed weights, activation and noise
    buffers stay where the graph's kernels will read and write them (as ``ops._workspace`` retires outgrown scratch).  Such
    a graph replays the weights as they were packed at ITS capture: re-capture after an optimizer step."""
    cache = owner.__dict__.setdefault("_plans", {})
    key = key + (ops.CONV3X3_ALGO,)
    plan = cache.pop(key, None)
    if plan is None or not plan.valid_for():
        _drop(plan)
        plan = build()
        if torch.cuda.is_current_stream_capturing():
            # built inside a stream capture: its buffers belong to that graph's memory pool and its packing launches were
            # recorded into the graph -- correct, but not something to keep for later eager calls.  Warm the stream up
            # before capturing (as bench.py does) to capture a plain launch list.
            return plan
    cache[key] = plan                      # re-insert: most recently used last
    while len(cache) > MAX_PLANS:
        _drop(cache.pop(next(iter(cache))))
    return plan


class StyleGAN2Plan(LaunchPlan):
    """Inference forward of the build-defined StyleGAN2 variant (stylegan2.StyleGAN2Generator): PixelNorm, the 8-layer
    style MLP, every modulation affine in two grouped launches, the 13 demodulation vectors in one, then per resolution
    two modulated MFMA convs (upfirdn2d x2 folded into the first one's staging) and ONE toRGB launch that also upsamples
    and adds the skip image."""

    def __init__(self, gen, B, device, precision="f32"):
        super().__init__(device)
        from .stylegan2 import SQRT2, StyledConv
        self.B = B
        self.precision = precision
        # ---- PixelNorm + style MLP ----
        self.pn = self.add(L.OP_PIXELNORM, L.PixelNormArgs(x=None, y=None, B=B, C=gen.input_dim, HW=1, eps=1e-8, sqrt_form=0))
        wn = self.buf(B, gen.input_dim)
        self.pn.y = wn.data_ptr()
        w = wn
        for lin in gen.style:
            out = self.buf(B, lin.weight.shape[0])
            if lin.activation:
                self.fc(w, lin.weight, lin.bias, lin.scale * SQRT2, lin.lr_mul * SQRT2, 0.2, out)
            else:
                self.fc(w, lin.weight, lin.bias, lin.scale, lin.lr_mul, 1.0, out)
            w = out
        # ---- modulations (every one is an affine of the same w) and demodulation vectors ----
        layers = [gen.conv1, gen.to_rgb1] + [m for i in range(len(gen.to_rgbs)) for m in (gen.convs[2 * i], gen.convs[2 * i + 1], gen.to_rgbs[i])]
        mods = [m.conv.modulation for m in layers]
        svec = [self.buf(B, m.weight.shape[0]) for m in mods]
        items = [(w, w.stride(0), m.weight, m.bias, m.scale, m.lr_mul, 1.0, s) for m, s in zip(mods, svec)]
        for k in range(0, len(items), L.FC_MAX_GROUPS):
            self.fc_grouped(items[k:k + L.FC_MAX_GROUPS], B)
        s_of = {id(m): s for m, s in zip(layers, svec)}
        styled = [m for m in layers if isinstance(m, StyledConv)]
        dvec = {id(m): self.buf(B, m.conv.out_channel) for m in styled}
        for k in range(0, len(styled), L.DEMOD_MAX_GROUPS):
            part = styled[k:k + L.DEMOD_MAX_GROUPS]
            arr = (L.DemodGroup * len(part))()
            for q, m in zip(arr, part):
                wt = m.conv.weight
                self.track(wt)
                # d[b,co] = rsqrt(scale^2 sum_ci s^2 (sum_k w^2) + eps): the tap sum depends on the weights only, so the plan
                # keeps n[co,ci] = sqrt(sum_k w[co,ci,k]^2) (refreshed with the parameter) and the demodulation launch reads
                # it as a one-tap kernel -- a ninth of the bytes of the 3x3 weights on every call
                wn = self.buf(wt.shape[0], wt.shape[1])
                self._refreshers.append(lambda src=wt, dst=wn: torch.sqrt(src.detach().pow(2).sum((2, 3)), out=dst))
                q.w, q.s, q.d = wn.data_ptr(), s_of[id(m)].data_ptr(), dvec[id(m)].data_ptr()
                q.Cin, q.Cout, q.taps, q.scale = wt.shape[1], wt.shape[0], 1, float(m.conv.scale)
            self.keep.append(arr)
            self.add(L.OP_DEMOD_GROUPED, L.DemodGroupedArgs(groups=C.addressof(arr), n_groups=len(part), B=B, eps=1e-8))
        # ---- noise ----
        shapes = [(B, 1, 4, 4)] + [(B, 1, 8 << i, 8 << i) for i in range(len(gen.to_rgbs)) for _ in range(2)]
        sizes = [sh[0] * sh[2] * sh[3] for sh in shapes]
        self.noise_flat = self.buf(sum(sizes))
        self.noise_views = [t.view(sh) for t, sh in zip(self.noise_flat.split(sizes), shapes)]
        self.noise_ops = []
        # ---- constant input, expanded over the batch (plan-owned copy, refreshed with the parameter) ----
        cin = gen.input.input
        x = self.buf(B, cin.shape[1], cin.shape[2], cin.shape[3])
        self._refreshers.append(lambda src=cin, dst=x: dst.copy_(src.detach().expand_as(dst)))
        self.track(cin)
        res_max = 4 << len(gen.to_rgbs)
        act = max([cin.shape[1] * 16] + [m.conv.out_channel * (8 << (i // 2)) ** 2 for i, m in enumerate(gen.convs)])
        pp = [self.buf(B * act) for _ in range(2)]
        cur = 0

        def styled_conv(m, x, k, cur):
            Cout = m.conv.out_channel
            H = x.shape[2] * (2 if m.upsample else 1)
            y = pp[cur][:B * Cout * H * H].view(B, Cout, H, H)
            nw = self.buf(Cout)                                   # the scalar noise weight, one copy per output channel
            self._refreshers.append(lambda src=m.noise.weight, dst=nw: dst.copy_(src.detach().expand_as(dst)))
            self.track(m.noise.weight, m.activate.bias)
            d = self.conv(x, m.conv.weight, Cout, bias=m.activate.bias, noise_w=nw, noise=self.noise_views[k], upsample=m.upsample,
                          up_fir=True, slope=0.2, out=y, out_scale=m.conv.scale, batch_scale=s_of[id(m)], demod=dvec[id(m)],
                          act_gain=SQRT2, precision=precision)
            self.noise_ops.append(d)
            return y

        def to_rgb(m, x, skip, y):
            wt = m.conv.weight
            self.track(wt, m.bias)
            return self.add(L.OP_TORGB, L.ToRGBArgs(x=x.data_ptr(), w=L.dptr(wt.reshape(wt.shape[0], wt.shape[1]), "weight"),
                                                    mod=s_of[id(m)].data_ptr(), bias=L.dptr(m.bias.view(-1), "bias"),
                                                    skip=skip.data_ptr() if skip is not None else None,
                                                    y=y.data_ptr() if y is not None else None, B=B, C=wt.shape[1], O=wt.shape[0],
                                                    H=x.shape[2], W=x.shape[3], in_scale=float(m.conv.scale)))

        x = styled_conv(gen.conv1, x, 0, cur)
        skip = self.buf(B, 3, 4, 4)
        self.torgb = to_rgb(gen.to_rgb1, x, None, skip if gen.to_rgbs else None)
        for i, rgb in enumerate(gen.to_rgbs):
            cur ^= 1
            x = styled_conv(gen.convs[2 * i], x, 1 + 2 * i, cur)
            cur ^= 1
            x = styled_conv(gen.convs[2 * i + 1], x, 2 + 2 * i, cur)
            last = i == len(gen.to_rgbs) - 1
            nxt = None if last else self.buf(B, 3, x.shape[2], x.shape[3])
            self.torgb = to_rgb(rgb, x, skip, nxt)
            skip = nxt
        self.out_shape = (B, 3, x.shape[2], x.shape[3])
        assert x.shape[2] == res_max
        self.finish(gen)

    def run(self, features, noises=None, kind_mask=L.ALL_OPS):
        self.refresh()
        self.captured = self.captured or torch.cuda.is_current_stream_capturing()
        self.pn.x = L.dptr(features, "features")
        if noises is None:
            self.noise_flat.normal_()
            for d, nv in zip(self.noise_ops, self.noise_views):
                d.noise = nv.data_ptr()
        else:
            if len(noises) != len(self.noise_ops):
                raise ValueError(f"expected {len(self.noise_ops)} noise tensors, got {len(noises)}")
            for d, nz, nv in zip(self.noise_ops, noises, self.noise_views):
                if nz.numel() != nv.numel():
                    raise L.SpkError(f"noise must be {tuple(nv.shape)}, got {tuple(nz.shape)}")
                d.noise = L.dptr(nz, "noise")
        y = torch.empty(self.out_shape, device=self.device, dtype=torch.float32)
        self.torgb.y = y.data_ptr()
        self.launch(kind_mask)
        return y
