"""``IRFD`` (model.py:28-126): three ResNet-50 trunk encoders (identity / emotion / pose), the
StyleGAN decoder ``Gd`` on their concatenated 6144-d latent, the discriminator ``D`` and the
emotion classifier ``Cm`` -- same attributes, methods, ``forward`` signature and 10-tuple result.

What differs from the reference, all on the host side and none in the arithmetic:
  * the per-forward debug work is gone: 12 ``.item()`` host syncs (``_log_feature_stats``,
    model.py:72-73,93-95), ~15 eagerly formatted DEBUG strings and two PNG files written to the CWD
    (``_visualize_feature_maps``, model.py:75-78,117-118).  ``IRFD.debug_side_effects = True``
    restores the statistics logging for anyone who wants it;
  * ``torch.utils.checkpoint`` (model.py:84-90) only changes *when* encoder activations exist, not
    their values; here the forward keeps no encoder activation beyond each block's output either
    way (BatchNorm/ReLU are folded into the consumers), and the backward pass recomputes.
The encoders are built with torchvision's own init (no download: ``resnet50(pretrained=True)`` at
model.py:61 needs the network, and ``self.apply(_init_weights)`` at model.py:48 re-initialises every
conv anyway, discarding the pretrained conv weights -- SURVEY.md 3.1 (iii)).
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn

from . import autograd as AG
from .decoder import StyleGenerator
from .discriminator import StyleDiscriminator
from .encoder import GroupedTrunks, ResNet50Trunk


class IRFD(nn.Module):
    debug_side_effects = False
    # Ei, Ee, Ep run the same ResNet-50 on the same image: by default every layer of the three is ONE grouped launch
    # (encoder.GroupedTrunks); False runs them one after another as the reference does.  Same parameters, same results.
    group_encoders = True
    pair_decoder = True          # the two Gd calls of a forward as one pass over both batches (StyleGenerator.forward_pair)

    def __init__(self, max_resolution=256):
        super().__init__()
        self.Ei = self._create_encoder()   # identity
        self.Ee = self._create_encoder()   # emotion
        self.Ep = self._create_encoder()   # pose
        self.Gd = StyleGenerator(input_dim=6144)
        self.D = StyleDiscriminator()
        self.Cm = nn.Linear(2048, 8)
        self.max_resolution = max_resolution
        self.current_resolution = max_resolution
        self.logger = logging.getLogger(__name__)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):          # model.py:50-54
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def adjust_for_resolution(self, resolution):
        self.current_resolution = resolution

    def _create_encoder(self):
        return ResNet50Trunk()

    def _prepare_generator_input(self, *features):
        return torch.cat([f.view(f.size(0), -1) for f in features], dim=1)

    def _log_feature_stats(self, tensor, name):
        if self.debug_side_effects and self.logger.isEnabledFor(logging.DEBUG):
            self.logger.debug("%s stats: mean=%.4f, std=%.4f, min=%.4f, max=%.4f", name, tensor.mean().item(),
                              tensor.std().item(), tensor.min().item(), tensor.max().item())

    def _emotion(self, fe):
        logits = AG.fc(fe.view(fe.size(0), -1), self.Cm.weight, self.Cm.bias, 1.0, 1.0, 1.0)
        return torch.softmax(logits, dim=1)

    def forward(self, x_s, x_t, swap_type=None, noises_s=None, noises_t=None):
        """-> (x_s_recon, x_t_recon, fi_s, fe_s, fp_s, fi_t, fe_t, fp_t, emotion_pred_s, emotion_pred_t).

        ``swap_type`` / ``noises_*`` are optional hooks for reproducible tests; by default the swap is
        drawn from the host RNG exactly as model.py:98 does and noise is drawn on the device."""
        if self.group_encoders:
            enc = self.__dict__.get("_enc_group")
            if enc is None or enc.trunks != [self.Ei, self.Ee, self.Ep]:
                enc = self.__dict__["_enc_group"] = GroupedTrunks([self.Ei, self.Ee, self.Ep], images=2)
            fi_s, fe_s, fp_s, fi_t, fe_t, fp_t = enc(x_s, x_t).split(2048, dim=1)
        else:
            fi_s, fe_s, fp_s = self.Ei(x_s), self.Ee(x_s), self.Ep(x_s)
            fi_t, fe_t, fp_t = self.Ei(x_t), self.Ee(x_t), self.Ep(x_t)
        self._log_feature_stats(fi_s, "Identity features")
        self._log_feature_stats(fe_s, "Emotion features")
        self._log_feature_stats(fp_s, "Pose features")
        if swap_type is None:
            swap_type = torch.randint(0, 3, (1,)).item()
        if swap_type == 0:
            fi_s, fi_t = fi_t, fi_s
        elif swap_type == 1:
            fe_s, fe_t = fe_t, fe_s
        else:
            fp_s, fp_t = fp_t, fp_s
        gin_s, gin_t = self._prepare_generator_input(fi_s, fe_s, fp_s), self._prepare_generator_input(fi_t, fe_t, fp_t)
        if self.pair_decoder and hasattr(self.Gd, "forward_pair"):
            # model.py:107-108's two decoder calls as one pass over both batches (decoder.StyleGenerator.forward_pair)
            x_s_recon, x_t_recon = self.Gd.forward_pair(gin_s, gin_t, noises_s, noises_t)
        else:
            x_s_recon = self.Gd(gin_s, noises_s)
            x_t_recon = self.Gd(gin_t, noises_t)
        return (x_s_recon, x_t_recon, fi_s, fe_s, fp_s, fi_t, fe_t, fp_t, self._emotion(fe_s), self._emotion(fe_t))
