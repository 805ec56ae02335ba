"""ResNet-50 trunk encoder -- the module ``IRFD._create_encoder`` builds at ``model.py:60-62``
(``nn.Sequential(*list(resnet50().children())[:-1])``), on the HIP kernels, forward and backward.

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

Backward (``TrunkFn``) has the reference's ``torch.utils.checkpoint`` semantics (model.py:84-90): the
forward keeps only the input image; backward re-runs the forward (which, exactly as the reference's
re-entrant checkpoint does, updates the BatchNorm running statistics a second time), then walks the
blocks in reverse: BatchNorm(+ReLU) backward as a two-pass reduce/apply pair on the raw conv output,
data gradients on the forward MFMA kernel with transpose-flipped weights (stride 2: zero-dilated
gradient), weight gradients on the MFMA wgrad kernel with the BatchNorm+ReLU of the input re-formed
in its staging.  The residual sum is folded into the data-gradient epilogue (accumulate flag).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

LAYERS = (3, 4, 6, 3)
WIDTHS = (64, 128, 256, 512)
EXPANSION = 4



def _wgrad_aside(side, dr, launch):
    """``launch()`` (a weight-gradient launch reading ``dr``) on the stream ``side``, behind everything queued on the current
    stream so far; None: in order on the current stream.  ``dr`` is a temporary of the backward: the allocator must not hand
    its memory out again before the side stream is done with it.  The caller joins ``side`` before it returns gradients."""
    if side is None:
        return launch()
    side.wait_stream(torch.cuda.current_stream(dr.device))
    with torch.cuda.stream(side):
        ops.side_stream_delay()
        dw = launch()
    dr.record_stream(side)
    return dw


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
    """One conv + its BatchNorm in the folded form."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d):
        self.conv, self.bn = conv, bn
        self.k, self.stride = conv.kernel_size[0], conv.stride[0]
        self.packed = ops.PackedConvWeight()

    def fwd(self, x, in_affine, training, stats_pool, keep):
        """-> record {y: raw conv output, affine: (scale, shift) its consumer applies, ...}."""
        conv, bn = self.conv, self.bn
        B, Cin, H, W = x.shape
        Cout = conv.out_channels
        Ho, Wo = ops.conv_out_size(H, self.k, self.stride), ops.conv_out_size(W, self.k, self.stride)
        cfg = ops.conv2d_pick_config(self.k, self.stride, B, Cin, Cout, Ho, Wo)
        stats = stats_pool.take(ops.stats_slots(cfg, self.k, self.stride, B, Cin, Cout, Ho, Wo) * 2 * Cout) if training else None
        y = ops.conv2d_fused(x, self.packed.get(conv.weight, cfg), Cout, self.k, self.stride, in_affine=in_affine,
                             stats=stats, config=cfg)
        if training:
            if bn.num_batches_tracked is not None:
                if bn.momentum is None:
                    bn.num_batches_tracked += 1                      # cumulative average: the count is needed right now
                else:
                    stats_pool.counters.append(bn.num_batches_tracked)   # bumped together at the end of the pass
            momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            fin = ops.bn_finalize(stats, B * Ho * Wo, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum,
                                  bn.eps, save=keep)
        else:
            fin = ops.bn_finalize(None, 1, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.0, bn.eps, save=keep)
        rec = {"y": y, "affine": (fin[0], fin[1])}
        if keep:
            rec.update(x=x, in_affine=in_affine, mean=fin[2], invstd=fin[3], training=training)
            if training:
                rec.update(stats=stats[:2 * Cout], count=B * Ho * Wo, momentum=momentum)   # copy 0 = the totals now
        return rec

    def second_update(self, rec, counters, replay):
        """The running-statistics update the reference's re-entrant checkpoint performs a second time when it re-runs
        the forward inside backward (model.py:84-90): same batch sums, same arithmetic, without recomputing the conv.
        Queued on ``replay``: every BatchNorm of the pass is updated by one launch (``ops.bn_replay_running``)."""
        bn = self.bn
        if not rec.get("training") or "stats" not in rec:
            return
        if bn.num_batches_tracked is not None:
            counters.append(bn.num_batches_tracked)
        replay.append((rec["momentum"], rec["stats"], rec["count"], bn.running_mean, bn.running_var))

    def put_bn_grads(self, grads, dg, db):
        grads[self.bn.weight], grads[self.bn.bias] = dg, db

    def bn_bwd(self, rec, g, mask_mode, mask_src=None, want_dz=False, **kw):
        """Gradient w.r.t. the raw conv output + BatchNorm parameter gradients (into ``grads``)."""
        out = ops.bn_backward(g, rec["y"], rec["affine"], rec["mean"], rec["invstd"], mask_mode, mask_src,
                              want_dz=want_dz, batch_stats=rec["training"], **kw)
        return out

    def conv_bwd(self, rec, dr, grads, need_dx, dx_out=None, accumulate=False, dilate=True, half=None, side=None):
        """Weight gradient (into ``grads``) and, if asked, the gradient w.r.t. the conv's (staged) input.  ``dilate=False``
        (a strided 1x1): that gradient stays at the conv's output size; ``half``: such a tensor, added at the even pixels."""
        conv = self.conv
        Cout, Cin = conv.out_channels, conv.in_channels
        x = rec["x"]
        B, _, H, W = x.shape
        grads[conv.weight] = _wgrad_aside(side, dr, lambda: ops.conv2d_wgrad(dr, x, Cout, Cin, self.k, self.stride,
                                                                            in_affine=rec["in_affine"]))
        if not need_dx:
            return None
        if (self.k == 3 and self.stride == 1 and half is None and ops.use_wino(B, Cout, Cin, H, W)
                and dr.data_ptr() % 16 == 0 and (dx_out is None or dx_out.data_ptr() % 16 == 0)):
            # fp32 Winograd F(2x2, 3x3) data gradient (ops.CONV3X3_ALGO)
            return ops.conv3x3_wino(dr.contiguous(), self.packed.get_wino(conv.weight, transpose_flip=True), Cin, out=dx_out, accumulate=accumulate)
        cfg, tf = ops.dgrad_plan(self.k, self.stride, B, Cout, Cin, (H, W), dr.shape[-2:], dx_out, accumulate)
        if half is not None and (cfg not in ops.GEMM2_CONFIGS or W % 4):
            dx_out, accumulate, half = ops.dilate2x(half, H, W), True, None     # the form that cannot add it in its epilogue
            cfg, tf = ops.dgrad_plan(self.k, self.stride, B, Cout, Cin, (H, W), dr.shape[-2:], dx_out, accumulate)
        return ops.conv2d_dgrad(dr, self.packed.get(conv.weight, cfg, transpose_flip=tf), Cin, self.k, self.stride,
                                (H, W), cfg, out=dx_out, accumulate=accumulate, dilate=dilate, accum_half=half)


class _StatsPool:
    """Zeroed fp64 pages per forward, sliced per BatchNorm (a few memsets instead of 53).  ``total`` = the sums of one
    copy of every BatchNorm; the high-resolution layers take several copies (``ops.stats_slots``), so the pool grows by
    pages when the first one is used up."""

    PAGE = 1 << 21               # doubles (16 MiB)

    def __init__(self, device, total):
        self.device = device
        self.buf = torch.zeros(max(total, self.PAGE), device=device, dtype=torch.float64)
        self.pos = 0
        self.counters = []       # num_batches_tracked buffers of the BatchNorms this pass went through

    def take(self, n):
        if self.pos + n > self.buf.numel():
            self.buf = torch.zeros(max(n, self.PAGE), device=self.device, dtype=torch.float64)
            self.pos = 0
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out


class _GroupPacked:
    """The packed images of a group's weights one after another (what a grouped launch reads), cached like
    ``ops.PackedConvWeight``."""

    def __init__(self):
        self._cache = {}

    def get(self, weights, config, transpose_flip=False):
        key = (config, transpose_flip)
        stamp = tuple((w.data_ptr(), w._version) for w in weights)
        hit = self._cache.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        # one launch into the previous image's storage (same key = same size): no per-tensor packs, no concatenation
        packed = ops.pack_conv_weights_list([w.detach() for w in weights], config, transpose_flip,
                                            out=hit[1] if hit is not None else None)
        self._cache[key] = (stamp, packed)
        return packed

    def get_wino(self, weights, transpose_flip=False):
        """The groups' Winograd images U = G g G^T one after another (one list-pack launch), cached the same way."""
        key = ("wino", bool(transpose_flip))
        stamp = tuple((w.data_ptr(), w._version) for w in weights)
        hit = self._cache.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        Cout, Cin = weights[0].shape[:2]
        n = (ops.L.lib().spk_conv2d_packed_bytes_wino(Cout, Cin) if transpose_flip else ops.L.lib().spk_conv2d_packed_bytes_wino(Cin, Cout)) // 4
        packed = hit[1] if hit is not None else torch.empty(len(weights) * n, device=weights[0].device, dtype=torch.float32)
        ops.pack_conv_weights_wino_into([w.detach() for w in weights], [packed[i * n:(i + 1) * n] for i in range(len(weights))], transpose_flip)
        self._cache[key] = (stamp, packed)
        return packed


_PENDING_BN = "_pending_bn_grads"      # key (never a Parameter) under which per-image BatchNorm gradients wait for their fold


class _GroupedConvBN:
    """The same conv + BatchNorm position of G trunks as ONE grouped launch (channels of the groups side by side).
    BatchNorm is per channel, so statistics, the folded affine, the residual add, the pools and the BatchNorm backward
    need nothing new: they simply see G*C channels.  ``flat`` = (gamma, beta, running_mean, running_var) of the G
    BatchNorms concatenated, provided per pass by ``GroupedTrunks``."""

    def __init__(self, members, shared_input=False):
        self.convs, self.bns = [m.conv for m in members], [m.bn for m in members]
        self.G = len(members)
        self.uniq = []                       # distinct trunks among the members (members repeat once per image)
        for m in members:
            if not any(m is u for u in self.uniq):
                self.uniq.append(m)
        self.k, self.stride = members[0].k, members[0].stride
        self.shared_input = shared_input
        self.packed = _GroupPacked()
        self.flat = None

    def _weights(self):
        return [c.weight for c in self.convs]

    def fwd(self, x, in_affine, training, stats_pool, keep):
        conv, bn, G = self.convs[0], self.bns[0], self.G
        B, _, H, W = x.shape
        Cin, Cout = conv.in_channels, conv.out_channels
        Ho, Wo = ops.conv_out_size(H, self.k, self.stride), ops.conv_out_size(W, self.k, self.stride)
        cfg = ops.conv2d_pick_config(self.k, self.stride, B, Cin, Cout, Ho, Wo)
        stats = stats_pool.take(ops.stats_slots(cfg, self.k, self.stride, B, Cin, Cout, Ho, Wo) * 2 * G * Cout) if training else None
        y = ops.conv2d_fused(x, self.packed.get(self._weights(), cfg), Cout, self.k, self.stride, in_affine=in_affine,
                             stats=stats, config=cfg, groups=G, shared_input=self.shared_input)
        gamma, beta, rm, rv = self.flat
        if training:
            if bn.momentum is None:
                raise NotImplementedError("grouped encoders: cumulative-average BatchNorm (momentum=None) is not supported")
            # distinct counters only; the pass bumps them by the number of images it carries
            stats_pool.counters.extend(b.num_batches_tracked for b in self.bns[:len(self.uniq)] if b.num_batches_tracked is not None)
            fin = ops.bn_finalize(stats, B * Ho * Wo, gamma, beta, rm, rv, bn.momentum, bn.eps, save=keep)
        else:
            fin = ops.bn_finalize(None, 1, gamma, beta, rm, rv, 0.0, bn.eps, save=keep)
        rec = {"y": y, "affine": (fin[0], fin[1])}
        if keep:
            rec.update(x=x, in_affine=in_affine, mean=fin[2], invstd=fin[3], training=training)
            if training:
                rec.update(stats=stats[:2 * G * Cout], count=B * Ho * Wo, momentum=bn.momentum)   # copy 0 = the totals now
        return rec

    def second_update(self, rec, counters, replay):
        if not rec.get("training") or "stats" not in rec:
            return
        counters.extend(b.num_batches_tracked for b in self.bns[:len(self.uniq)] if b.num_batches_tracked is not None)
        gamma, beta, rm, rv = self.flat
        replay.append((rec["momentum"], rec["stats"], rec["count"], rm, rv))

    def put_bn_grads(self, grads, dg, db):
        if self.G == len(self.uniq):
            for q in range(len(self.uniq)):
                bn = self.bns[q]
                grads[bn.weight], grads[bn.bias] = dg.view(self.G, -1)[q], db.view(self.G, -1)[q]
        else:       # several images per trunk: folded for all layers at once at the end of the pass (flush_bn_grads)
            grads.setdefault(_PENDING_BN, []).append((self, dg, db))

    @staticmethod
    def flush_bn_grads(grads):
        """Sum the per-image BatchNorm gradients of every layer of the pass with ONE concatenation and ONE reduction
        (they were two tiny reductions per layer: 106 launches per pass)."""
        pend = grads.pop(_PENDING_BN, None)
        if not pend:
            return
        imgs = pend[0][0].G // len(pend[0][0].uniq)
        flat = torch.cat([t.view(imgs, -1) for _, dg, db in pend for t in (dg, db)], 1).sum(0)
        off = 0
        for m, dg, db in pend:
            T, n = len(m.uniq), dg.numel() // m.G
            for which in range(2):
                part = flat[off:off + T * n].view(T, n)
                off += T * n
                for q in range(T):
                    bn = m.bns[q]
                    grads[bn.weight if which == 0 else bn.bias] = part[q]

    def bn_bwd(self, rec, g, mask_mode, mask_src=None, want_dz=False, **kw):
        return ops.bn_backward(g, rec["y"], rec["affine"], rec["mean"], rec["invstd"], mask_mode, mask_src,
                               want_dz=want_dz, batch_stats=rec["training"], **kw)

    def conv_bwd(self, rec, dr, grads, need_dx, dx_out=None, accumulate=False, dilate=True, half=None, side=None):
        conv, G = self.convs[0], self.G
        Cout, Cin = conv.out_channels, conv.in_channels
        x = rec["x"]
        B, _, H, W = x.shape
        T = len(self.uniq)
        dw = _wgrad_aside(side, dr, lambda: ops.conv2d_wgrad(dr, x, Cout, Cin, self.k, self.stride, in_affine=rec["in_affine"],
                                                             groups=G, shared_input=self.shared_input, fold=G // T))
        dw = dw.view(T, Cout, Cin, self.k, self.k)
        for q in range(T):
            grads[self.convs[q].weight] = dw[q]
        if not need_dx:
            return None
        if (self.k == 3 and self.stride == 1 and half is None and not self.shared_input and ops.use_wino(B, Cout, Cin, H, W, groups=G)
                and dr.data_ptr() % 16 == 0 and (dx_out is None or dx_out.data_ptr() % 16 == 0)):
            # fp32 Winograd F(2x2, 3x3) data gradient, the G trunks as groups of one launch (ops.CONV3X3_ALGO)
            return ops.conv3x3_wino(dr.contiguous(), self.packed.get_wino(self._weights(), transpose_flip=True), Cin, groups=G,
                                    out=dx_out, accumulate=accumulate)
        cfg, tf = ops.dgrad_plan(self.k, self.stride, B, Cout, Cin, (H, W), dr.shape[-2:], dx_out, accumulate)
        if half is not None and (cfg not in ops.GEMM2_CONFIGS or W % 4):
            dx_out, accumulate, half = ops.dilate2x(half, H, W), True, None     # the form that cannot add it in its epilogue
            cfg, tf = ops.dgrad_plan(self.k, self.stride, B, Cout, Cin, (H, W), dr.shape[-2:], dx_out, accumulate)
        return ops.conv2d_dgrad(dr, self.packed.get(self._weights(), cfg, transpose_flip=tf), Cin, self.k, self.stride,
                                (H, W), cfg, out=dx_out, accumulate=accumulate, groups=G, dilate=dilate, accum_half=half)


class GroupedTrunks:
    """G trunks of one architecture that see the same image -- IRFD's Ei, Ee, Ep (model.py:84-90) -- run as ONE network
    of grouped launches: a third of the launches, three times the work per launch (the 8x8 .. 32x32 layers of a single
    ResNet-50 at batch 8 cannot fill 256 CUs).  Feature order = trunk order: [B, G*2048, 1, 1].  Parameters stay where
    they are (each trunk's own modules, so ``state_dict`` and checkpoints are unchanged); per pass the BatchNorm vectors
    are gathered into flat tensors with four ``torch.cat`` calls and the running statistics written back with two
    ``torch._foreach_copy_`` calls."""

    def __init__(self, trunks, images=1):
        """``images`` = 2: one pass takes TWO images (``group(x_s, x_t)``): the network then has 2*G groups -- every
        trunk once per image -- so BatchNorm statistics stay per (trunk, image) exactly as in two separate passes (they
        are per channel, and the groups are separate channels); parameter gradients of a trunk's two groups add, and the
        running statistics receive the two momentum updates in the reference's order (x_s, then x_t)."""
        self.trunks = list(trunks)
        self.images = int(images)
        self._plan = None
        self.recompute = False

    @property
    def training(self):
        return self.trunks[0].training

    def train(self, mode=True):
        for t in self.trunks:
            t.train(mode)
        return self

    def parameters(self):
        for t in self.trunks:
            yield from t.parameters()

    def _build_plan(self):
        plans = []
        for t in self.trunks:
            t._build_plan()
            plans.append(t._plan)
        rep = self.images
        stem = _GroupedConvBN([p[0] for p in plans] * rep, shared_input=(rep == 1))
        blocks = []
        for parts in zip(*[p[1] for p in plans]):
            blocks.append(tuple(_GroupedConvBN([b[i] for b in parts] * rep) if parts[0][i] is not None else None for i in range(4)))
        self._stats_total = rep * len(self.trunks) * self.trunks[0]._stats_total
        self._plan = (stem, blocks)
        self._members = [stem] + [c for blk in blocks for c in blk if c is not None]

    def _load_flats(self):
        ms = self._members
        cat = lambda get: torch.cat([get(bn).detach() for m in ms for bn in m.bns])
        flats = [cat(lambda bn: bn.weight), cat(lambda bn: bn.bias), cat(lambda bn: bn.running_mean), cat(lambda bn: bn.running_var)]
        pos = 0
        for m in ms:
            n = m.G * m.bns[0].num_features
            m.flat = tuple(f[pos:pos + n] for f in flats)
            pos += n
        self._flat_running = (flats[2], flats[3])

    def _store_running(self, first_image_first=True):
        ms = self._members
        sizes = [bn.num_features for m in ms for bn in m.bns]
        for k, get in ((0, lambda bn: bn.running_mean), (1, lambda bn: bn.running_var)):
            parts = list(self._flat_running[k].split(sizes))
            if self.images == 1:
                torch._foreach_copy_([get(bn) for m in ms for bn in m.bns], parts)
                continue
            # Each group's slice holds (1-m)*r0 + m*mu of ITS image, both computed from the same r0.  Two momentum
            # updates in sequence (image a, then image b) give (1-m)*[(1-m)*r0 + m*mu_a] + m*mu_b
            #   = (1-m)*slice_a + slice_b - (1-m)*r0.
            bufs, a, b, pos = [], [], [], 0
            for m in ms:
                T = len(m.uniq)
                for q in range(T):
                    bufs.append(get(m.bns[q]))
                    s_img, t_img = parts[pos + q], parts[pos + T + q]
                    a.append(s_img if first_image_first else t_img)
                    b.append(t_img if first_image_first else s_img)
                pos += m.G
            keep = 1.0 - ms[0].bns[0].momentum
            torch._foreach_mul_(bufs, -keep)
            torch._foreach_add_(bufs, b)
            torch._foreach_add_(bufs, a, alpha=keep)

    def _run(self, x, keep):
        if self._plan is None:
            self._build_plan()
        self._load_flats()
        out = ResNet50Trunk._run(self, x, keep)
        if self.training:
            self._store_running()
        return out

    def _second_bn_update(self, recs):
        self._load_flats()
        ResNet50Trunk._second_bn_update(self, recs)
        self._store_running(first_image_first=False)     # the checkpoint re-runs the later call (x_t) first

    def _backward(self, recs, dfeat):
        return ResNet50Trunk._backward(self, recs, dfeat)

    def __call__(self, *xs):
        if len(xs) != self.images:
            raise ValueError(f"GroupedTrunks: expected {self.images} image batch(es), got {len(xs)}")
        # one image: every group reads the same three channels; two: each group gets its own copy (3 channels, negligible)
        x = xs[0] if self.images == 1 else torch.cat([x for x in xs for _ in self.trunks], dim=1)
        return TrunkFn.apply(x, self, torch.is_grad_enabled(), *self.parameters())


class TrunkFn(torch.autograd.Function):
    """The trunk under the reference's ``checkpoint(E, x)`` (model.py:84-90).

    ``trunk.recompute`` True: forward keeps the input only, backward re-runs the forward, then back-propagates -- the
    reference's memory-saving schedule.  False (default here: 288 GB of HBM, ~1 GB of activations per B=8 pass): the
    forward keeps the raw conv outputs, backward uses them directly and replays only the checkpoint's visible side
    effect, the second running-statistics update of every BatchNorm.  Gradients and buffers are identical either way;
    the stored form saves one encoder forward per pass."""

    @staticmethod
    def forward(ctx, x, trunk, grad_mode, *params):
        need = grad_mode and any(ctx.needs_input_grad)
        if need and not trunk.recompute:
            y, ctx.recs = trunk._run(x, keep=True)
            ctx.trunk, ctx.training = trunk, trunk.training
            return y
        y = trunk._run(x, keep=False)[0]
        if need:
            ctx.trunk, ctx.training, ctx.recs = trunk, trunk.training, None
            ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dfeat):
        trunk = ctx.trunk
        if ctx.recs is not None:
            recs, ctx.recs = ctx.recs, None
            if ctx.training:
                trunk._second_bn_update(recs)
        else:
            (x,) = ctx.saved_tensors
            was_training = trunk.training
            trunk.train(ctx.training)
            try:
                _, recs = trunk._run(x, keep=True)          # the checkpoint recomputation
            finally:
                trunk.train(was_training)
        grads = trunk._backward(recs, dfeat.contiguous())
        return (None, None, None) + tuple(grads.get(p) for p in trunk.parameters())


class ResNet50Trunk(nn.Sequential):
    recompute = False        # see TrunkFn: keep activations (default) or re-run the forward inside backward

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
        return TrunkFn.apply(x, self, torch.is_grad_enabled(), *self.parameters())

    def _run(self, x, keep):
        """Forward; with ``keep`` also returns what backward needs (raw conv outputs, block outputs, statistics)."""
        if self._plan is None:
            self._build_plan()
        stem, blocks = self._plan
        training = self.training
        x = x.contiguous()
        pool = _StatsPool(x.device, self._stats_total) if training else None
        s = stem.fwd(x, None, training, pool, keep)
        cur = ops.maxpool3x3s2(s["y"], s["affine"][0], s["affine"][1])      # bn1 + relu folded into the pool's loads
        recs = {"stem": s, "blocks": []} if keep else None
        for c1, c2, c3, down in blocks:
            r1 = c1.fwd(cur, None, training, pool, keep)                      # block input is materialised (post-ReLU)
            r2 = c2.fwd(r1["y"], r1["affine"], training, pool, keep)
            r3 = c3.fwd(r2["y"], r2["affine"], training, pool, keep)
            rd = down.fwd(cur, None, training, pool, keep) if down is not None else None
            a3 = r3["affine"]
            if rd is not None:
                out = ops.bn_add_relu(r3["y"], a3[0], a3[1], rd["y"], rd["affine"][0], rd["affine"][1], relu=True)
            else:
                out = ops.bn_add_relu(r3["y"], a3[0], a3[1], cur, None, None, relu=True)
            if keep:
                recs["blocks"].append((r1, r2, r3, rd, out))
            cur = out
        if pool is not None and pool.counters:
            torch._foreach_add_(pool.counters, getattr(self, "images", 1))   # 53 one-element kernels -> one fused launch
        return ops.global_avgpool(cur), recs

    def _second_bn_update(self, recs):
        stem, blocks = self._plan
        counters, replay = [], []
        stem.second_update(recs["stem"], counters, replay)
        for (c1, c2, c3, down), (r1, r2, r3, rd, _) in zip(blocks, recs["blocks"]):
            c1.second_update(r1, counters, replay)
            c2.second_update(r2, counters, replay)
            c3.second_update(r3, counters, replay)
            if down is not None:
                down.second_update(rd, counters, replay)
        for mom in sorted({m for m, *_ in replay}):          # one launch for all BatchNorms (per momentum value: one)
            ops.bn_replay_running([it[1:] for it in replay if it[0] == mom], mom)
        if counters:
            torch._foreach_add_(counters, getattr(self, "images", 1))

    def _backward(self, recs, dfeat):
        """dfeat [B,2048,1,1] -> {parameter: gradient}."""
        side = ops.side_stream(dfeat.device)           # the weight gradients' stream (joined before the gradients are returned)
        if side is not None and torch.cuda.is_current_stream_capturing():
            side = None                                # a stream capture stays on the capturing stream
        stem, blocks = self._plan
        grads = {}
        g, per_plane, g_scale = dfeat.view(dfeat.size(0), -1).contiguous(), True, None
        for (c1, c2, c3, down), (r1, r2, r3, rd, out) in zip(reversed(blocks), reversed(recs["blocks"])):
            if per_plane:                      # last block: gradient of the global average pool, one value per plane
                g_scale = 1.0 / (out.shape[2] * out.shape[3])
            # out = relu(bn3(r3) + identity): mask from the block output; dz also feeds the identity branch
            dr3, dg, db, dz = c3.bn_bwd(r3, g, ops.MASK_TENSOR, mask_src=out, want_dz=True,
                                        g_scale=g_scale if per_plane else 1.0, g_per_plane=per_plane)
            c3.put_bn_grads(grads, dg, db)
            dv2 = c3.conv_bwd(r3, dr3, grads, need_dx=True, side=side)
            dr2, dg, db = c2.bn_bwd(r2, dv2, ops.MASK_RECOMPUTE)
            c2.put_bn_grads(grads, dg, db)
            dv1 = c2.conv_bwd(r2, dr2, grads, need_dx=True, side=side)
            dr1, dg, db = c1.bn_bwd(r1, dv1, ops.MASK_RECOMPUTE)
            c1.put_bn_grads(grads, dg, db)
            if down is None:
                # d(block input) = dz (identity) + conv1's data gradient, summed in the conv epilogue
                g = c1.conv_bwd(r1, dr1, grads, need_dx=True, dx_out=dz, accumulate=True, side=side)
            else:
                drd, dg, db = down.bn_bwd(rd, dz, ops.MASK_NONE)
                down.put_bn_grads(grads, dg, db)
                if down.stride == 2:
                    # the downsample conv reads the even pixels only: its data gradient stays at ITS output size and conv1's
                    # data gradient adds it at the even pixels in its own epilogue -- no dilated copy, no read of one
                    t = down.conv_bwd(rd, drd, grads, need_dx=True, dilate=False, side=side)
                    g = c1.conv_bwd(r1, dr1, grads, need_dx=True, half=t, side=side)
                else:
                    g = down.conv_bwd(rd, drd, grads, need_dx=True, side=side)
                    g = c1.conv_bwd(r1, dr1, grads, need_dx=True, dx_out=g, accumulate=True, side=side)
            per_plane = False
        # stem: max-pool adjoint (bn1+relu re-formed on the fly), BatchNorm backward, 7x7 weight gradient
        s = recs["stem"]
        dv0 = ops.maxpool3x3s2_bwd(s["y"], g, s["affine"][0], s["affine"][1])
        dr0, dg, db = stem.bn_bwd(s, dv0, ops.MASK_RECOMPUTE)
        stem.put_bn_grads(grads, dg, db)
        stem.conv_bwd(s, dr0, grads, need_dx=False, side=side)       # the image itself takes no gradient
        _GroupedConvBN.flush_bn_grads(grads)
        if side is not None:
            torch.cuda.current_stream(dfeat.device).wait_stream(side)     # every weight gradient is complete behind this point
        return grads
