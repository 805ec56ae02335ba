"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the encoder the reference builds at
``model.py:60-62``: ``nn.Sequential(*list(torchvision.models.resnet50().children())[:-1])``.

The arithmetic lives in third-party ``torchvision`` (not vendored in /root/reference, no version
pinned, not installed here -- SURVEY.md 8c), so this file restates the *published* ResNet-50 v1.5
definition (He et al. 2015 bottleneck; stride on the 3x3 as torchvision does): children
0 conv 7x7 s2 p3 (no bias), 1 BatchNorm2d, 2 ReLU, 3 MaxPool 3x3 s2 p1, 4-7 layer1..4 with
(3,4,6,3) bottlenecks of widths (64,128,256,512) x4 expansion, a 1x1 (strided) downsample branch
on the first block of each layer, 8 AdaptiveAvgPool2d(1).  State-dict keys are torchvision's
Sequential keys (``0.weight``, ``1.running_mean``, ``4.0.conv1.weight``, ``5.0.downsample.0.weight`` ...).

PARITY UNPINNED by the reference (it holds no fixture or test for this boundary).  Pinned instead
against an independent implementation of the same architecture that IS installed here
(``transformers.models.resnet.ResNetModel`` built from config, random init; tests/test_resnet_oracle.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
WIDTHS = (64, 128, 256, 512)
EXPANSION = 4
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def trunk_param_shapes():
    """key -> shape of every parameter / buffer of the 9-child Sequential (318 tensors incl.
    ``num_batches_tracked``; 23,508,032 parameters)."""
    sd = {}

    def bn(prefix, c):
        sd[prefix + ".weight"] = (c,)
        sd[prefix + ".bias"] = (c,)
        sd[prefix + ".running_mean"] = (c,)
        sd[prefix + ".running_var"] = (c,)
        sd[prefix + ".num_batches_tracked"] = ()

    sd["0.weight"] = (64, 3, 7, 7)
    bn("1", 64)
    inplanes = 64
    for li, (nblk, width) in enumerate(zip(LAYERS, WIDTHS)):
        for bi in range(nblk):
            p = f"{4 + li}.{bi}."
            sd[p + "conv1.weight"] = (width, inplanes, 1, 1)
            bn(p + "bn1", width)
            sd[p + "conv2.weight"] = (width, width, 3, 3)
            bn(p + "bn2", width)
            sd[p + "conv3.weight"] = (width * EXPANSION, width, 1, 1)
            bn(p + "bn3", width * EXPANSION)
            if bi == 0:
                sd[p + "downsample.0.weight"] = (width * EXPANSION, inplanes, 1, 1)
                bn(p + "downsample.1", width * EXPANSION)
            inplanes = width * EXPANSION
    return sd


def _bn(x, sd, prefix, training, update):
    """nn.BatchNorm2d.forward: batch statistics in training mode (biased variance for the
    normalisation, unbiased for the running estimate, momentum 0.1), running statistics in eval."""
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training and not update:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)


def bottleneck(x, sd, p, stride, has_down, training=False, update=False):
    """torchvision Bottleneck.forward (v1.5: the stride sits on conv2)."""
    out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1", training, update))
    out = F.relu(_bn(F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2", training, update))
    out = _bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3", training, update)
    identity = x
    if has_down:
        identity = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1", training, update)
    return F.relu(out + identity)


def resnet50_trunk(x, sd, prefix="", training=False, update_running_stats=False, return_stages=False):
    """[B,3,H,W] -> [B,2048,1,1] (model.py:60-62; called 6x per IRFD.forward, model.py:84-90)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    stages = []
    y = F.conv2d(x, sd["0.weight"], stride=2, padding=3)
    y = F.relu(_bn(y, sd, "1", training, update_running_stats))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    stages.append(y)
    for li, nblk in enumerate(LAYERS):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 0) else 1
            y = bottleneck(y, sd, f"{4 + li}.{bi}.", stride, bi == 0, training, update_running_stats)
        stages.append(y)
    out = F.adaptive_avg_pool2d(y, 1)
    return (out, stages) if return_stages else out


def trunk_flops(h=256, w=256):
    """2*MAC of one trunk forward per image (10.677 GFLOP at 256^2)."""
    total = 0
    ho, wo = h // 2, w // 2
    total += 2 * 64 * 3 * 49 * ho * wo
    ho, wo = ho // 2, wo // 2
    inplanes = 64
    for li, (nblk, width) in enumerate(zip(LAYERS, WIDTHS)):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 0) else 1
            total += 2 * width * inplanes * ho * wo                       # conv1 at input resolution
            h2, w2 = ho // stride, wo // stride
            total += 2 * width * width * 9 * h2 * w2
            total += 2 * width * EXPANSION * width * h2 * w2
            if bi == 0:
                total += 2 * width * EXPANSION * inplanes * h2 * w2
            inplanes, ho, wo = width * EXPANSION, h2, w2
    return total
