"""ResNet-50 trunk encoder -- the module ``IRFD._create_encoder`` builds at ``model.py:60-62``
(``nn.Sequential(*list(resnet50().children())[:-1])``), on the HIP kernels.

``ResNet50Trunk`` IS an ``nn.Sequential`` whose children sit at torchvision's indices (0 conv1,
1 bn1, 2 relu, 3 maxpool, 4-7 layer1..4, 8 avgpool), so its ``state_dict`` keys are the reference's
(``Ei.0.weight``, ``Ei.1.running_mean``, ``Ei.4.0.conv1.weight``, ``Ei.5.0.downsample.0.weight`` ...)
and ``IRFD.apply(_init_weights)`` (model.py:48-54) finds the same ``nn.Conv2d`` modules.  The
children only hold parameters/buffers; ``forward`` runs the whole trunk MI355X-style:

* every conv is one launch of the MFMA implicit-GEMM kernel; its epilogue accumulates the
  BatchNorm batch sums (train mode) -- no statistics pass over the activation;
* BatchNorm + ReLU are never materialised: ``bn_finalize`` turns the sums into a per-channel affine
  that the *consumer* conv (or the max-pool) applies while staging its input;
* only each block's output (bn3 + identity + ReLU) is written, by one HBM-bound pass.

A bottleneck is 3-4 conv launches + 3-4 tiny finalize launches + 1 elementwise pass, against the
reference's 10-13 ATen kernels with a full HBM round trip each.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

LAYERS = (3, 4, 6, 3)
WIDTHS = (64, 128, 256, 512)
EXPANSION = 4


class Bottleneck(nn.Module):
    """Parameter holder with torchvision's Bottleneck attribute names (v1.5: stride on conv2)."""

    def __init__(self, inplanes, width, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * EXPANSION, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * EXPANSION)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, width * EXPANSION, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(width * EXPANSION))
        self.stride = stride


class _ConvBN:
    """One conv + its BatchNorm: launches the conv (statistics in the epilogue when training) and
    returns (raw output, (scale, shift)) -- the affine the consumer folds into its staging."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d):
        self.conv, self.bn = conv, bn
        self.k, self.stride = conv.kernel_size[0], conv.stride[0]
        self.packed = ops.PackedConvWeight()

    def __call__(self, x, in_affine, training, stats_pool):
        conv, bn = self.conv, self.bn
        B, Cin, H, W = x.shape
        Cout = conv.out_channels
        Ho, Wo = ops.conv_out_size(H, self.k, self.stride), ops.conv_out_size(W, self.k, self.stride)
        cfg = ops.conv2d_pick_config(self.k, self.stride, B, Cin, Cout, Ho, Wo)
        stats = stats_pool.take(2 * Cout) if training else None
        y = ops.conv2d_fused(x, self.packed.get(conv.weight, cfg), Cout, self.k, self.stride, in_affine=in_affine,
                             stats=stats, config=cfg)
        if training:
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            affine = ops.bn_finalize(stats, B * Ho * Wo, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     momentum, bn.eps)
        else:
            affine = ops.bn_finalize(None, 1, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.0, bn.eps)
        return y, affine


class _StatsPool:
    """One zeroed fp64 buffer per forward, sliced per BatchNorm (a single memset instead of 53)."""

    def __init__(self, device, total):
        self.buf = torch.zeros(total, device=device, dtype=torch.float64)
        self.pos = 0

    def take(self, n):
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out


class ResNet50Trunk(nn.Sequential):
    def __init__(self):
        layers = []
        inplanes = 64
        for li, (nblk, width) in enumerate(zip(LAYERS, WIDTHS)):
            blocks = []
            for bi in range(nblk):
                blocks.append(Bottleneck(inplanes, width, 2 if (bi == 0 and li > 0) else 1, bi == 0))
                inplanes = width * EXPANSION
            layers.append(nn.Sequential(*blocks))
        super().__init__(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                         nn.MaxPool2d(3, stride=2, padding=1), *layers, nn.AdaptiveAvgPool2d(1))
        # torchvision's ResNet.__init__ init: kaiming-normal fan_out convs, BN weight 1 / bias 0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._plan = None

    # -- launch plan: built lazily so that parameter replacement (.to(), load_state_dict) is honoured
    def _build_plan(self):
        stem = _ConvBN(self[0], self[1])
        blocks = []
        for li in range(4):
            for blk in self[4 + li]:
                blocks.append((_ConvBN(blk.conv1, blk.bn1), _ConvBN(blk.conv2, blk.bn2), _ConvBN(blk.conv3, blk.bn3),
                               _ConvBN(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None))
        self._stats_total = 2 * sum(m.num_features for m in self.modules() if isinstance(m, nn.BatchNorm2d))
        self._plan = (stem, blocks)

    def forward(self, x):
        if self._plan is None:
            self._build_plan()
        stem, blocks = self._plan
        training = self.training
        x = x.contiguous()
        pool = _StatsPool(x.device, self._stats_total) if training else None
        y, aff = stem(x, None, training, pool)
        cur = ops.maxpool3x3s2(y, aff[0], aff[1])                    # bn1 + relu folded into the pool's loads
        for c1, c2, c3, down in blocks:
            r1, a1 = c1(cur, None, training, pool)                    # block input is materialised (post-ReLU)
            r2, a2 = c2(r1, a1, training, pool)
            r3, a3 = c3(r2, a2, training, pool)
            if down is not None:
                rd, ad = down(cur, None, training, pool)
                cur = ops.bn_add_relu(r3, a3[0], a3[1], rd, ad[0], ad[1], relu=True)
            else:
                cur = ops.bn_add_relu(r3, a3[0], a3[1], cur, None, None, relu=True)
        return ops.global_avgpool(cur)
