"""Host-side mirror of the reference's live decoder (``styleganv1.py:448-635``): same class names,
constructor arguments, ``forward`` signatures and ``state_dict`` keys, but every forward runs on the
HIP kernels of libspk_hip.so.  The reference's per-op chain

    up -> conv3x3 -> +bias -> +noise -> lrelu -> *(s0+1)+s1          (styleganv1.py:624-633)

is ONE kernel launch per half-block here (ops.conv3x3_fused); the 13 style affines and the mapping
stack run on the weight-streaming FC kernel.

Differences a caller can observe, all opt-in or device-side only:
  * every ``forward`` that draws noise accepts an optional explicit ``noise``/``noises`` argument
    (the reference draws ``torch.randn`` inside ``ApplyNoise.forward``, styleganv1.py:454-455);
    without it noise is drawn on the device exactly as the reference does;
  * no DEBUG f-string logging on the hot path (the reference formats ~20 strings per forward).
There is no CPU path: a CPU tensor raises (``_lib.SpkError``).
"""
from __future__ import annotations

import logging
import math

import torch
import torch.nn as nn

from . import ops
from . import autograd as AG
from . import plan as PL

LRELU = 0.2


class FC(nn.Module):
    """``FC`` (styleganv1.py:471-495): y = lrelu_0.2(x @ (W*w_lrmul)^T + b*b_lrmul)."""

    def __init__(self, in_channels, out_channels, gain=2 ** 0.5, use_wscale=False, lrmul=1.0, bias=True):
        super().__init__()
        he_std = gain * in_channels ** (-0.5)
        if use_wscale:
            init_std, self.w_lrmul = 1.0 / lrmul, he_std * lrmul
        else:
            init_std, self.w_lrmul = he_std / lrmul, lrmul
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels) * init_std)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
            self.b_lrmul = lrmul
        else:
            self.bias = None
            self.b_lrmul = 1.0

    def forward(self, x):
        lead = x.shape[:-1]
        x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
        y = AG.fc(x2, self.weight, self.bias, self.w_lrmul, self.b_lrmul, LRELU)
        return y.view(*lead, -1)


class ApplyNoise(nn.Module):
    """``ApplyNoise`` (styleganv1.py:448-456).  Stand-alone use only; inside the network the noise
    add is part of the conv epilogue."""

    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(channels))

    def forward(self, x, noise):
        if noise is None:
            noise = torch.randn(x.size(0), 1, x.size(2), x.size(3), device=x.device, dtype=x.dtype)
        return AG.bias_noise_style(x.contiguous(), None, self.weight, noise.to(x.device).contiguous(), None, x.size(0))


class ApplyStyle(nn.Module):
    """``ApplyStyle`` (styleganv1.py:458-468): x*(s0+1)+s1 with [s0|s1] = FC_gain1(latent)."""

    def __init__(self, latent_size, channels, use_wscale):
        super().__init__()
        self.linear = FC(latent_size, channels * 2, gain=1.0, use_wscale=use_wscale)

    def style(self, latent):
        return self.linear(latent)

    def forward(self, x, latent):
        return AG.bias_noise_style(x.contiguous(), None, None, None, self.style(latent), x.size(0))


class SynthesisBlock(nn.Module):
    """``SynthesisBlock`` (styleganv1.py:612-635): two fused launches."""

    def __init__(self, in_channels, out_channels, resolution):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.noise1 = ApplyNoise(out_channels)
        self.noise2 = ApplyNoise(out_channels)
        self.style_mod1 = ApplyStyle(512, out_channels, use_wscale=True)
        self.style_mod2 = ApplyStyle(512, out_channels, use_wscale=True)
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        self.out_channels = out_channels
        self._pk1, self._pk2 = ops.PackedConvWeight(), ops.PackedConvWeight()

    def _half(self, x, conv, pk, noise_mod, noise, style, upsample, weight=None):
        B, Cin, Hs, Ws = x.shape
        H, W = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
        if noise is None:
            noise = torch.randn(B, 1, H, W, device=x.device, dtype=torch.float32)
        return AG.fused_conv(x, conv.weight if weight is None else weight, conv.bias, noise_mod.weight, noise, style, upsample, LRELU, pk)

    def forward(self, x, w, noise1=None, noise2=None, styles=None, weights=None):
        """``w`` is [B,2,512].  ``styles`` (optional) = precomputed ([B,2C],[B,2C]) affine outputs; ``weights`` (optional) = the two
        conv weights behind the network's ``AG.gate_weights`` node (SynthesisNetwork.forward)."""
        if styles is None:
            styles = (self.style_mod1.style(w[:, 0]), self.style_mod2.style(w[:, 1]))
        w1, w2 = weights if weights is not None else (None, None)
        x = self._half(x.contiguous(), self.conv1, self._pk1, self.noise1, noise1, styles[0], True, w1)
        x = self._half(x, self.conv2, self._pk2, self.noise2, noise2, styles[1], False, w2)
        return x


class SynthesisNetwork(nn.Module):
    """``SynthesisNetwork`` (styleganv1.py:569-610)."""

    def __init__(self, resolution=256, fmap_base=8192, fmap_max=512):
        super().__init__()
        self.resolution_log2 = int(math.log2(resolution))
        self.num_layers = self.resolution_log2 * 2 - 2

        def nf(stage):
            return min(int(fmap_base / (2.0 ** stage)), fmap_max)

        self.const_input = nn.Parameter(torch.ones(1, nf(1), 4, 4))
        self.bias = nn.Parameter(torch.zeros(nf(1)))
        self.style_mod = ApplyStyle(512, nf(1), use_wscale=True)
        self.noise_input1 = ApplyNoise(nf(1))
        self.layers = nn.ModuleList(
            SynthesisBlock(nf(res - 2), nf(res - 1), res) for res in range(3, self.resolution_log2 + 1))
        self.to_rgb = nn.Conv2d(nf(self.resolution_log2 - 1), 3, kernel_size=1)
        self.logger = logging.getLogger(__name__)

    def noise_shapes(self, batch):
        shapes = [(batch, 1, 4, 4)]
        for i in range(len(self.layers)):
            s = 8 << i
            shapes += [(batch, 1, s, s)] * 2
        return shapes

    use_plan = True      # inference forwards go out as one pre-built launch list (plan.DecoderPlan); False: launch by launch
    # "f32": exact fp32 arithmetic on the f32 MFMA pipe (the default, the reference's precision).  "bf16x3": OPT-IN speed path for
    # inference -- the 3x3 convs of the >= 32^2 layers on the bf16 matrix pipe with every operand split hi + lo (three MFMAs per
    # product, fp32 accumulation; csrc/conv3x3_bf16x3.hip): ~3e-5 rel-L2 against the reference where the bound is 1e-3.
    precision = "f32"

    def _inference(self, x):
        return not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())))

    def forward(self, w, noises=None):
        """``w``: [B, num_layers, 512]; ``noises``: optional list of the 2*len(layers)+1 noise
        tensors in call order (drawn on the device when omitted, as the reference does)."""
        B = w.size(0)
        w = w.contiguous()
        if noises is not None and len(noises) != 2 * len(self.layers) + 1:
            raise ValueError(f"expected {2 * len(self.layers) + 1} noise tensors, got {len(noises)}")
        if self.use_plan and w.is_cuda and len(self.layers) * 2 + 1 <= ops.L.FC_MAX_GROUPS and self._inference(w):
            key = (B, w.device, torch.cuda.current_stream(w.device).cuda_stream, "w", self.precision)
            p = PL.plan_for(self, key, lambda: PL.DecoderPlan(self, B, w.device, precision=self.precision))
            return p.run(w, None if noises is None else [n.contiguous() for n in noises])
        if noises is None:
            # one device draw for the whole step, cut into the 13 per-layer tensors (the reference draws them one by
            # one inside ApplyNoise.forward, styleganv1.py:455; device RNG streams are not comparable across
            # implementations anyway -- parity tests pass explicit noise)
            shapes = self.noise_shapes(B)
            sizes = [s[0] * s[2] * s[3] for s in shapes]
            flat = torch.randn(sum(sizes), device=w.device, dtype=torch.float32)
            noises = [t.view(s) for t, s in zip(flat.split(sizes), shapes)]
        elif len(noises) != 2 * len(self.layers) + 1:
            raise ValueError(f"expected {2 * len(self.layers) + 1} noise tensors, got {len(noises)}")
        styles = None
        # the 13 style affines depend only on w -> one grouped launch instead of 13 (and, under autograd, one node whose
        # backward is two launches writing the rows of a single [B, L, 512] latent gradient)
        mods = [self.style_mod] + [m for layer in self.layers for m in (layer.style_mod1, layer.style_mod2)]
        if w.is_cuda and len(mods) <= ops.L.FC_MAX_GROUPS and w.size(1) >= len(mods):
            if torch.is_grad_enabled() and (w.requires_grad or any(p.requires_grad for p in self.parameters())):
                styles = AG.style_fc_group(w, [m.linear for m in mods], LRELU)
            else:
                styles = ops.fc_grouped((w[:, j], m.linear.weight, m.linear.bias, m.linear.w_lrmul, m.linear.b_lrmul, LRELU)
                                        for j, m in enumerate(mods))
        x = AG.bias_noise_style(self.const_input, self.bias, self.noise_input1.weight, noises[0],
                                   styles[0] if styles is not None else self.style_mod.style(w[:, 0]), B)
        # a training pass: the conv weights go through ONE gate node, whose backward (after every conv's) joins the second stream
        # that the convs' weight gradients are queued on (autograd.WeightGateFn)
        gated = None
        if torch.is_grad_enabled() and w.is_cuda and ops.side_stream(w.device) is not None:
            cw = [c.weight for layer in self.layers for c in (layer.conv1, layer.conv2)]
            if all(t.requires_grad for t in cw):
                gated = AG.gate_weights(cw)
        if w.is_cuda:
            # every stale Winograd weight image of the pass (forward and, under autograd, data-gradient orientation) in ONE launch: after
            # an optimizer step that is all of them -- 20 launches otherwise
            items, res, grad = [], 4, torch.is_grad_enabled()
            for layer in self.layers:
                res *= 2
                for conv, pk in ((layer.conv1, layer._pk1), (layer.conv2, layer._pk2)):
                    Co, Ci = conv.weight.shape[:2]
                    if not ops.train_bf16x3(B, Ci, Co, res, res) and ops.use_wino(B, Ci, Co, res, res):
                        items.append((pk, conv.weight, False))
                    if grad and not ops.train_bf16x3(B, Co, Ci, res, res) and ops.use_wino(B, Co, Ci, res, res):
                        items.append((pk, conv.weight, True))
            if items:
                ops.prepack_wino(items)
        for i, layer in enumerate(self.layers):
            x = layer(x, w[:, 2 * i + 1:2 * i + 3], noises[1 + 2 * i], noises[2 + 2 * i],
                      styles=(styles[1 + 2 * i], styles[2 + 2 * i]) if styles is not None else None,
                      weights=(gated[2 * i], gated[2 * i + 1]) if gated is not None else None)
        return AG.to_rgb(x, self.to_rgb.weight, self.to_rgb.bias)


class StyleGenerator(nn.Module):
    """``StyleGenerator`` (styleganv1.py:497-567): mapping -> broadcast -> truncation (a plain
    scale of rows [:cutoff], :540-543) -> train-only style mixing (:547-554) -> synthesis."""

    def __init__(self, input_dim=6144, latent_dim=512, mapping_layers=8, style_mixing_prob=0.9,
                 truncation_psi=0.7, truncation_cutoff=8):
        super().__init__()
        self.input_dim = input_dim
        self.latent_dim = latent_dim
        self.style_mixing_prob = style_mixing_prob
        self.truncation_psi = truncation_psi
        self.truncation_cutoff = truncation_cutoff
        self.mapping = nn.Sequential(*[FC(input_dim if i == 0 else latent_dim, latent_dim, lrmul=0.01, use_wscale=True)
                                       for i in range(mapping_layers)])
        self.synthesis = SynthesisNetwork()
        self.bn = None
        self.logger = logging.getLogger(__name__)

    def forward(self, features, noises=None, style_mix=None):
        """``style_mix`` (optional, tests): ``(mix_features, mix_layer)`` replacing the three RNG draws of the
        train-mode mixing branch, or ``False`` to skip the branch."""
        L = self.synthesis.num_layers
        syn = self.synthesis
        if (not self.training and syn.use_plan and features.is_cuda and features.dim() == 2 and len(syn.layers) * 2 + 1 <= ops.L.FC_MAX_GROUPS
                and not (torch.is_grad_enabled() and (features.requires_grad or any(p.requires_grad for p in self.parameters())))):
            # eval + no gradient: mapping, truncation and synthesis as ONE launch list (one crossing of the C boundary)
            if noises is not None and len(noises) != 2 * len(syn.layers) + 1:
                raise ValueError(f"expected {2 * len(syn.layers) + 1} noise tensors, got {len(noises)}")
            B = features.size(0)
            # (the truncation scale is folded into the style FCs' multipliers when the plan is built: part of the key)
            key = (B, features.device, torch.cuda.current_stream(features.device).cuda_stream, "features", syn.precision,
                   self.truncation_psi, self.truncation_cutoff)
            p = PL.plan_for(self, key, lambda: PL.DecoderPlan(syn, B, features.device, generator=self, precision=syn.precision))
            return p.run(features if features.stride(1) == 1 else features.contiguous(),
                         None if noises is None else [n.contiguous() for n in noises])
        w = self.mapping(features).unsqueeze(1).repeat(1, L, 1)
        if self.truncation_psi and self.truncation_cutoff:
            coefs = torch.ones_like(w)
            coefs[:, :self.truncation_cutoff] *= self.truncation_psi
            w = coefs * w
        if self.training and self.style_mixing_prob > 0 and style_mix is not False:
            # RNG draws in the reference's order: rand(1), randn_like(features), randint (styleganv1.py:548-552)
            if style_mix is not None or torch.rand(1) < self.style_mixing_prob:
                with torch.no_grad():
                    z2 = style_mix[0] if style_mix is not None else torch.randn_like(features)
                    w2 = self.mapping(z2).unsqueeze(1).repeat(1, L, 1)
                    mix_layer = style_mix[1] if style_mix is not None else torch.randint(1, w.size(1), (1,)).item()
                    # in place and under no_grad, as the reference: autograd never sees the overwrite, so the
                    # gradient of the mixed rows still flows into the first mapping pass (a reference quirk)
                    w[:, mix_layer:] = w2[:, mix_layer:]
        return self.synthesis(w, noises)

    def forward_pair(self, features_a, features_b, noises_a=None, noises_b=None):
        """``(self(features_a, noises_a), self(features_b, noises_b))`` as ONE pass over the concatenated batch -- the two
        decoder calls of ``IRFD.forward`` (model.py:107-108).  The decoder has no cross-sample operation, so the frames are
        those of the two calls; the HOST RNG is consumed exactly as by two calls in sequence (rand, randint of call a, then of
        call b: styleganv1.py:548-552) and each half gets its own style-mixing decision and layer.  Device noise is one draw
        for both halves (device RNG streams are not comparable with the reference's per-layer draws either way; parity tests
        pass explicit noise).  Half the launches, twice the work per launch (the 4^2 .. 16^2 layers at batch 8 leave most
        of the chip idle), and the shared parameters' gradients come out of one backward instead of two plus an add."""
        if (noises_a is None) != (noises_b is None) or features_a.shape != features_b.shape:
            return self.forward(features_a, noises_a), self.forward(features_b, noises_b)
        B = features_a.size(0)
        feats = torch.cat([features_a, features_b], 0)
        noises = None if noises_a is None else [torch.cat([na, nb], 0) for na, nb in zip(noises_a, noises_b)]
        if not (self.training and self.style_mixing_prob > 0):
            y = self.forward(feats, noises, style_mix=False)
            return y[:B], y[B:]
        L = self.synthesis.num_layers
        w = self.mapping(feats).unsqueeze(1).repeat(1, L, 1)
        if self.truncation_psi and self.truncation_cutoff:
            coefs = torch.ones_like(w)
            coefs[:, :self.truncation_cutoff] *= self.truncation_psi
            w = coefs * w
        # the draws of call a, then of call b, in the reference's order; the second mapping passes run as one
        picks = []
        for half, f in enumerate((features_a, features_b)):
            if torch.rand(1) < self.style_mixing_prob:
                z2 = torch.randn_like(f)
                picks.append((half, z2, torch.randint(1, L, (1,)).item()))
        if picks:
            with torch.no_grad():
                w2 = self.mapping(torch.cat([z for _, z, _ in picks], 0)).unsqueeze(1).repeat(1, L, 1)
                for j, (half, _, mix_layer) in enumerate(picks):
                    # in place and under no_grad, as the reference (the gradient quirk of the overwrite is kept)
                    w[half * B:(half + 1) * B, mix_layer:] = w2[j * B:(j + 1) * B, mix_layer:]
        y = self.synthesis(w, noises)
        return y[:B], y[B:]
