#!/usr/bin/env python3
"""Headline benchmark: generator frames/s at 256^2, batch 8 per GPU (BASELINE.json metric,
configs[1]: StyleGAN synthesis forward only, batch 8, 1x MI355X).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one ``StyleGenerator.forward`` (mapping + 13 style affines + prologue + 12 fused 3x3 conv
launches + toRGB) over one batch of 8 synthetic [6144] latents, inputs and weights resident in HBM,
noise drawn on the device inside the step as the reference does (styleganv1.py:455).  fp32
end to end (exact-f32 MFMA).  N > 1: one rank per GPU, each rank an independent replica on its own batch for
the headline (the forward path has no exchange step -- SURVEY.md 8e); barrier + synchronize on both sides of the
timed region, MAX over ranks, rank 0 prints ONE JSON line.  Either the driver starts the ranks
(``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``: WORLD_SIZE is set) or ``python bench.py
--gpus N`` does it itself: the parent -- before any GPU call -- starts N fresh child ranks through
torch.distributed.run, relays rank 0's line and exits with the children's return code (it never re-executes itself).

Extra objects in the line (headline fields unchanged):
  roofline       -- the dominant kernel (spkconv::conv_kernel, MFMA-bound): algorithmic conv FLOPs per
                    step / conv kernel time per step, timed live with HIP events on the launch stream.
  cpu_baseline   -- the CPU oracle (a port of the reference's algorithm, parity-pinned to it) timed
                    on this box's host cores on the same B=8 workload (rank 0, N=1 only).
  eager          -- the same step WITHOUT the hipGraph: what an unchanged ``model.Gd(x)`` caller gets (one
                    launch list per call, plan.DecoderPlan).
  bf16x3         -- the OPT-IN split-precision speed path on the headline workload (3x3 convs of the >= 16^2 layers on the bf16
                    matrix pipe, operands split hi + lo, three MFMAs per product; ~2e-5 rel-L2 against the reference, asserted in
                    tests/test_bf16x3_gpu.py); roofline against the dense bf16 MFMA peak.  The headline stays exact fp32.
  stylegan2_variant -- the build-defined StyleGAN2 decoder (A11) on the same workload, with its own roofline.
  decoder_512_b4 -- BASELINE config 5: SynthesisNetwork(512), batch 4.
  train_step     -- BASELINE config 3: the IRFD generator step (3 encoders x 2 images + 2 decoder passes,
                    fwd + bwd + clip + Adam) at batch 16.
  d_step         -- the discriminator step of train.py:155-183 at batch 8.
  single_frame   -- BASELINE config 1: one 256^2 frame (B=1) on the GPU and on the CPU port.
(the above on rank 0 at N=1 only; ``--headline-only`` skips them.)
N > 1 adds, on every rank (figures are MAX over ranks), the ONE path with a real exchange step -- BASELINE config 4:
  train_step_dp  -- the IRFD generator step at batch 8 per rank through dp.GradBucketReducer on ``--backend`` (nccl = RCCL
                    over xGMI): ms/step, pairs/s over all ranks, gradient bytes exchanged per step, buckets launched from
                    backward hooks vs by finish(), the backend and world size torch.distributed reports, and the exposed
                    communication = step(N ranks) - the same step inside no_sync() (no exchange).
  d_step_dp      -- the same for the discriminator step (76 MB of gradients, R1 double backward).
  decoder_512_b4 -- BASELINE config 5 per rank (replicas).
"""
from __future__ import annotations

import argparse
import contextlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 8
RES = 256
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 (the headline figures with 2:1 sparsity are not used)
GFLOP_PER_PAIR = 593.5            # IRFD fwd+bwd as the REFERENCE executes it: decoder x3, checkpointed encoders x4
GFLOP_PER_PAIR_EXECUTED = 593.5 - 6 * 10.677   # what this build executes by default: the encoders' activations are stored
                                               # (encoder.TrunkFn), so the checkpoint's second forward (6 x 10.677) never runs


def decoder_conv_flops(batch, resolution=RES):
    """Algorithmic FLOPs (2*MAC) of the 3x3 convs per step -- SURVEY.md 8(d): 56.17 GFLOP/frame at 256^2."""
    total, res, cin = 0, 8, 512
    while res <= resolution:
        cout = min(int(8192 / (2.0 ** (res.bit_length() - 2))), 512)
        total += 2 * 9 * res * res * (cin * cout + cout * cout)
        cin, res = cout, res * 2
    return total * batch


def pair_executed_gflop(ops, frames):
    """EXECUTED matrix GFLOP of one training pair (x_s, x_t) in the generator step as THIS build runs it: the encoders' activations
    are stored (no checkpoint recompute: 3 x 10.677 per encoder pass, 6 passes), the decoder's two frames run forward + data
    gradient + weight gradient (3 x 56.214 each) -- and where a 3x3 layer's forward / data gradient / weight gradient go to the
    Winograd kernels (ops.use_wino / ops.use_wgrad_wino at the step's decoder batch of ``frames``) they execute 16/36 of their
    algorithmic FLOPs."""
    saved, r, cin = 0.0, 8, 512
    while r <= RES:
        cout = min(int(8192 / (2.0 ** (r.bit_length() - 2))), 512)
        for ci, co in ((cin, cout), (cout, cout)):
            fl = 2 * 9 * ci * co * r * r / 1e9                       # per frame
            if ops.use_wino(frames, ci, co, r, r):
                saved += fl * (1 - 16.0 / 36.0)                      # forward
            if ops.use_wino(frames, co, ci, r, r):
                saved += fl * (1 - 16.0 / 36.0)                      # data gradient
            if ops.use_wgrad_wino(frames, ci, co, r, r):
                saved += fl * (1 - 16.0 / 36.0)                      # weight gradient
        cin, r = cout, r * 2
    # the encoders: conv2 of every stride-1 Bottleneck (torchvision ResNet-50: 3 / 3 / 5 / 2 of them at 64 / 128 / 256 / 512 channels and
    # 64^2 / 32^2 / 16^2 / 8^2 pixels for a 256^2 input) -- data gradient and weight gradient on the Winograd kernels where they serve
    # the shape (6 passes as groups of one launch, `frames` / 2 images each); the forward stays on the direct kernel
    saved_enc = 0.0
    for n_blocks, c, r in ((3, 64, 64), (3, 128, 32), (5, 256, 16), (2, 512, 8)):
        fl = n_blocks * 2 * 9 * c * c * r * r / 1e9               # per image and pass
        if ops.use_wino(frames // 2, c, c, r, r, groups=6):
            saved_enc += fl * (1 - 16.0 / 36.0)                       # data gradient
        if ops.use_wgrad_wino(frames // 2, c, 6 * c, r, r):
            saved_enc += fl * (1 - 16.0 / 36.0)                       # weight gradient
    return GFLOP_PER_PAIR_EXECUTED - 2 * saved - 6 * saved_enc


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("SPK_CPU_THREADS", n))))


def cpu_baseline(threads):
    """Oracle (port) on the host cores: 1 warm-up + 9 timed B=8 forwards (about 10 s of CPU work)."""
    from oracle import decoder_ref as R
    from oracle.weights_recipe import recipe_input, recipe_noises
    pkg = importlib.import_module("speak-hack_amd")
    torch.set_num_threads(threads)
    g = pkg.StyleGenerator(6144)
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    feats = recipe_input("bench.cpu.features", (BATCH, 6144))
    noises = recipe_noises("bench.cpu", BATCH, RES)
    times = []
    with torch.no_grad():
        for i in range(10):
            t0 = time.perf_counter()
            R.style_generator(feats, sd, noises)
            times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    return {"value": round(BATCH / med, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle.decoder_ref.style_generator, B={BATCH}, 256^2, fp32, median of 9 after 1 warm-up "
                      f"({med * 1e3:.0f} ms/step)"}


def committed_traffic(which, launches):
    """HBM bytes per conv launch from the committed PMC passes (profiles/conv_traffic.json: FETCH_SIZE / WRITE_SIZE, separate
    rocprofv3 passes over tools/pmc_decoder.py); None when the file does not hold that figure."""
    try:
        with open(os.path.join(ROOT, "profiles", "conv_traffic.json")) as f:
            t = json.load(f)
        if which == "bf16x3":
            return int((t["bf16x3_fetch_bytes_per_step"] + t["bf16x3_write_bytes_per_step"]) / max(1, launches))
        if which == "sg2":
            return int(t["sg2_bytes_per_launch"])
        if which == "f32_direct":
            return int(t["direct_f32_bytes_per_launch"])
        return int(t["bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def plan_conv_flops(plan, L):
    """(algorithmic FLOPs, EXECUTED matrix FLOPs, launches on the Winograd kernel, launches on the direct kernel) of a launch plan's
    3x3 conv launches.  A Winograd F(2x2, 3x3) launch executes 16 multiply-adds per 2x2 output tile and (ci, co) pair where the
    direct form executes 36: its matrix instructions do 16/36 of the algorithmic FLOPs."""
    alg = exe = 0.0
    n_w = n_d = 0
    for kind, d in plan.ops:
        if kind != L.OP_CONV2D:
            continue
        fl = 2.0 * d.kh * d.kw * d.Cin * d.Cout * d.H * d.W * d.B
        alg += fl
        if d.flags & L.CONV_WINOGRAD:
            exe += fl * 16.0 / 36.0
            n_w += 1
        else:
            exe += fl
            n_d += 1
    return alg, exe, n_w, n_d


def event_ms(fn, reps):
    """HIP events on the launch stream around ``reps`` calls of ``fn`` (after one untimed call)."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


_SIDE = None


def capture(fn):
    """``fn`` replayed as one hipGraph (the C ABI neither allocates nor synchronises; the noise draw is graph-safe Philox).
    Warm-up and capture run on ONE side stream: launch plans are per stream, so the capture reuses the warmed plan instead
    of building (and re-packing weights) inside the graph."""
    global _SIDE
    if _SIDE is None:
        _SIDE = torch.cuda.Stream()
    side = _SIDE
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()                                   # warm every cache (plans, packed weights, workspace) before capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = fn()
    torch.cuda.synchronize()
    return g, out


def plan_of(module):
    plans = module.__dict__.get("_plans")
    return next(reversed(plans.values())) if plans else None


def conv_time_ms(module, feats, L):
    """Time of a step's conv launches.  Events between individual launches would sit on the GPU timeline themselves (a
    marker packet with a cache flush: ~30 us per conv when launches are queued back to back).  Instead the step's launch
    list is captured twice -- whole, and with ``kind_mask`` leaving out the conv launches (spk_launch_list's documented
    measurement use: no switch inside the library) -- and both graphs are timed by HIP events around R replays."""
    capture(lambda: module(feats))                # make sure the side stream's plan exists and is the most recent one
    plan = plan_of(module)
    if plan is None:
        return None, None, None
    g_all, _ = capture(lambda: plan.run(feats))
    g_nc, _ = capture(lambda: plan.run(feats, kind_mask=L.ALL_OPS & ~(1 << L.OP_CONV2D)))
    R = 20
    t_nc = event_ms(g_nc.replay, R)
    ms = event_ms(g_all.replay, R) - t_nc
    n = sum(1 for kind, _ in plan.ops if kind == L.OP_CONV2D)
    up_ms = 0.0
    if any(kind == L.OP_UPSAMPLE2X for kind, _ in plan.ops):     # the x2 images the Winograd x2 layers read: their own launches
        g_nu, _ = capture(lambda: plan.run(feats, kind_mask=L.ALL_OPS & ~(1 << L.OP_CONV2D) & ~(1 << L.OP_UPSAMPLE2X)))
        up_ms = t_nc - event_ms(g_nu.replay, R)
    return ms, n, up_ms


def conv_roofline(plan, L, conv_ms, up_ms, kernel, traffic, step_ms, extra=None):
    """The roofline object of a decoder plan's conv launches: achieved = EXECUTED matrix TFLOP/s (what the MFMA pipe does), with the
    algorithmic rate beside it -- the two differ on the Winograd launches by 36/16."""
    alg, exe, n_w, n_d = plan_conv_flops(plan, L)
    ach = exe / (conv_ms * 1e-3) / 1e12
    r = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
         "achieved_is": "EXECUTED matrix FLOPs / conv time: a Winograd F(2x2,3x3) launch executes 16/36 of its algorithmic FLOPs",
         "algorithmic_tflops": round(alg / (conv_ms * 1e-3) / 1e12, 2), "algorithmic_gflop_per_step": round(alg / 1e9, 2),
         "executed_gflop_per_step": round(exe / 1e9, 2), "launches_per_step": n_w + n_d, "launches_winograd": n_w, "launches_direct": n_d,
         "conv_ms_per_step": round(conv_ms, 4), "conv_share_of_step": round(conv_ms / step_ms, 3),
         "upsample_ms_per_step": round(up_ms, 4)}
    if extra:
        r.update(extra)
    return r


def irfd_steps(pkg, dev, which, B, steps, warmup, precision="f32", sg2=False, observe=None):
    """The generator step (config 3) / discriminator step of train.py:150-210 on synthetic pairs -- tools/train_step_bench.py
    in brief.  Returns ms per step."""
    import model as M
    import torch.nn.functional as F
    dp = importlib.import_module("speak-hack_amd.dp")
    T = importlib.import_module("speak-hack_amd.training")
    torch.manual_seed(0)
    net = M.IRFD()
    if sg2:                                    # BASELINE config 3 read literally: the StyleGAN2 decoder behind the encoders
        net.Gd = importlib.import_module("speak-hack_amd.stylegan2").StyleGAN2Generator(6144)
    net = net.to(dev).train()
    for n, p in net.named_parameters():
        p.requires_grad_(n.startswith("D.") == (which == "d"))
    params = [p for p in net.parameters() if p.requires_grad]
    red = dp.GradBucketReducer(params)
    opt = torch.optim.Adam(net.D.parameters() if which == "d" else net.Gd.parameters(), lr=1e-4, betas=(0.5, 0.999))
    x_s, x_t, f_s, f_t = (torch.rand(B, 3, 256, 256, device=dev) * 2 - 1 for _ in range(4))
    bce = lambda pred, label: F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, label))

    def step():
        red.zero_grad()
        if which == "d":
            nz = T.add_instance_noise
            loss = (bce(net.D(nz(x_s)), 0.9) + bce(net.D(nz(x_t)), 0.9)) / 2 + (bce(net.D(nz(f_s)), 0.1) + bce(net.D(nz(f_t)), 0.1)) / 2 \
                + 10.0 * (T.compute_r1_reg(net.D, x_s) + T.compute_r1_reg(net.D, x_t)) / 2
            loss.backward()
            red.finish()
        else:
            out = net(x_s, x_t)
            loss = ((out[0] - x_s) ** 2).mean() + ((out[1] - x_t) ** 2).mean()
            loss.backward()
            red.finish()
            red.clip_(1.0)
        opt.step()

    with pkg.ops.train_conv_precision(precision):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if observe is not None:                # tools/op_callsites.py: the timed steps under a TorchDispatchMode
            with observe:
                for _ in range(steps):
                    step()
        else:
            for _ in range(steps):
                step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    red.remove()
    del net, opt, red
    torch.cuda.empty_cache()
    return ms


def irfd_dp_steps(pkg, dev, which, B, steps, warmup, dist, rank, algo="all_reduce"):
    """BASELINE config 4: the generator (or discriminator) step at ``B`` samples per rank with the gradient exchange of
    train.py:333-338,399-401 (accelerate's DDP) done by dp.GradBucketReducer on the initialised backend.  Times the step
    WITH the exchange, then the same step inside ``no_sync()`` (every rank steps on its own gradients: no collective) --
    the difference is the communication backward could not hide.  All figures are MAX over ranks."""
    import model as M
    import torch.nn.functional as F
    dp = importlib.import_module("speak-hack_amd.dp")
    T = importlib.import_module("speak-hack_amd.training")
    world = dist.get_world_size()
    torch.manual_seed(1234 + rank)             # replicas seeded differently: the reducer's broadcast makes them equal
    net = M.IRFD().to(dev).train()
    for n, p in net.named_parameters():
        p.requires_grad_(n.startswith("D.") == (which == "d"))
    params = [p for p in net.parameters() if p.requires_grad]
    red = dp.GradBucketReducer(params, algo=algo)
    with torch.no_grad():
        for b in net.buffers():                # BatchNorm / spectral-norm buffers, as DDP's broadcast_buffers
            dist.broadcast(b, src=0)
    opt = torch.optim.Adam(net.D.parameters() if which == "d" else net.Gd.parameters(), lr=1e-4, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(10 + rank)            # rank r's shard is seed 10 + r
    x_s, x_t, f_s, f_t = ((torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev) for _ in range(4))
    bce = lambda pred, label: F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, label))

    def step(sync=True):
        red.zero_grad()
        with (contextlib.nullcontext() if sync else red.no_sync()):
            if which == "d":
                nz = T.add_instance_noise
                loss = (bce(net.D(nz(x_s)), 0.9) + bce(net.D(nz(x_t)), 0.9)) / 2 + (bce(net.D(nz(f_s)), 0.1) + bce(net.D(nz(f_t)), 0.1)) / 2 \
                    + 10.0 * (T.compute_r1_reg(net.D, x_s) + T.compute_r1_reg(net.D, x_t)) / 2
            else:
                out = net(x_s, x_t)
                loss = ((out[0] - x_s) ** 2).mean() + ((out[1] - x_t) ** 2).mean()
            loss.backward()
        if sync:
            red.finish()
        if which == "g":
            red.clip_(1.0)
        opt.step()

    def timed(sync):
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(sync)
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps * 1e3

    for _ in range(max(2, warmup)):            # the first finish() re-buckets in autograd-completion order
        step()
    ms = timed(True)
    by_hook, by_finish = list(red.stats["launched_by_hook"]), list(red.stats["launched_by_finish"])
    # (stats are reset by zero_grad(): these are the last timed step's)
    n_buckets, cold = len(red.buckets), [i for i, b in enumerate(red.buckets) if b["cold"]]
    # one more synchronising step with per-bucket stamps (host: launch / wait entry / wait return; stream events at the same
    # points): where an exchange that costs more than its bytes spends the time
    red.profile = True
    torch.cuda.synchronize()
    dist.barrier()
    t_step = time.perf_counter()
    step()
    host_ms = (time.perf_counter() - t_step) * 1e3
    torch.cuda.synchronize()
    timeline = red.timeline()
    red.profile = False
    step(False)
    ms_local = timed(False)                    # last: the replicas drift apart from here on
    res = {"ms_per_step": round(ms, 2), "samples_per_rank": B, "pairs_per_s": round(world * B / ms * 1e3, 1),
           "backend": dist.get_backend(), "world_size": world, "algo": algo, "grad_bytes_per_step": red.bytes_per_step(),
           "buckets": n_buckets, "bucket_bytes": red.bucket_bytes, "buckets_launched_by_hook": len(by_hook),
           "buckets_launched_by_finish": len(by_finish), "cold_buckets": len(cold),
           "ms_per_step_no_exchange": round(ms_local, 2), "exposed_comm_ms": round(ms - ms_local, 2),
           "bucket_timeline_rank0": {"host_ms_of_the_stamped_step": round(host_ms, 2), "buckets": timeline}}
    red.remove()
    del net, opt, red
    torch.cuda.empty_cache()
    return res


def train_iterations(pkg, dev, B, iters=5, warmup_iters=5, precision="f32", dist=None, rank=0, algo="all_reduce"):
    """The reference's REAL iteration (train.py:150-210, config.yaml:18 G_steps = 5): ``training.train_iteration`` over
    ``iters`` consecutive steps starting at a multiple of G_steps -- 5 discriminator steps and 1 generator step WITH the
    adversarial term through model.D, the clip over all parameters, Adam on Gd and on D.  With ``dist``: the data-parallel
    form (both reducers).  Returns (ms per iteration, pairs/s over all ranks)."""
    import model as M
    T = importlib.import_module("speak-hack_amd.training")
    world = dist.get_world_size() if dist is not None else 1
    torch.manual_seed(1234 + rank)
    net = M.IRFD().to(dev).train()
    red_G = red_D = None
    if dist is not None:
        red_G, red_D = T.make_reducers(net, algo=algo)
    opt_G = torch.optim.Adam(net.Gd.parameters(), lr=2e-4, betas=(0.5, 0.999))          # config.yaml:19-20
    opt_D = torch.optim.Adam(net.D.parameters(), lr=5e-5, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(10 + rank)
    batch = {"source_image": (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev),
             "target_image": (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev),
             "emotion_labels_s": torch.randint(0, 8, (B,), generator=g).to(dev),
             "emotion_labels_t": torch.randint(0, 8, (B,), generator=g).to(dev)}
    kw = dict(G_steps=5, r1_weight=1.0, stylegan_loss_weight=0.1, grad_clip_value=1.0, reducer_G=red_G, reducer_D=red_D)   # config.yaml:18,21,36,43

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    with pkg.ops.train_conv_precision(precision):
        for step in range(warmup_iters):
            T.train_iteration(net, batch, opt_G, opt_D, step, **kw)
        sync()
        t0 = time.perf_counter()
        for step in range(iters):
            T.train_iteration(net, batch, opt_G, opt_D, step, **kw)
        sync()
        el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        red_G.remove(), red_D.remove()
    del net, opt_G, opt_D
    torch.cuda.empty_cache()
    ms = el / iters * 1e3
    return ms, world * B / ms * 1e3


def single_frame(pkg, dev, threads):
    """BASELINE config 1: one 256^2 frame from one [1,6144] latent -- hipGraph replay on the GPU, and the CPU port."""
    gen = pkg.StyleGenerator(6144).eval().to(dev)
    f1 = torch.randn(1, 6144, device=dev)
    with torch.no_grad():
        g1, _ = capture(lambda: gen(f1))
        ms = event_ms(g1.replay, 50)
        for _ in range(3):
            gen(f1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            gen(f1)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 20 * 1e3
    res = {"what": "BASELINE config 1: StyleGenerator forward, ONE frame (B=1), 256^2, fp32", "ms_per_frame": round(ms, 4),
           "frames_per_s": round(1e3 / ms, 1), "eager_ms_per_frame": round(eager, 4)}
    if threads:
        from oracle import decoder_ref as R
        torch.set_num_threads(threads)
        cpu = pkg.StyleGenerator(6144)
        sd = {k: v.detach().clone() for k, v in cpu.state_dict().items()}
        x = torch.randn(1, 6144)
        noises = [torch.randn(1, 1, r, r) for r in (4, 8, 8, 16, 16, 32, 32, 64, 64, 128, 128, 256, 256)]
        ts = []
        with torch.no_grad():
            for _ in range(6):
                t0 = time.perf_counter()
                R.style_generator(x, sd, noises)
                ts.append(time.perf_counter() - t0)
        med = sorted(ts[1:])[2]
        res["cpu_port"] = {"ms_per_frame": round(med * 1e3, 1), "frames_per_s": round(1 / med, 2), "cores": threads,
                           "sample": "oracle.decoder_ref.style_generator, B=1, median of 5 after 1 warm-up"}
    del gen, g1
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start N fresh child ranks (one per GPU) through
    torch.distributed.run and relay their output.  Runs BEFORE this process has made any GPU call (importing torch does
    not initialise HIP); the children are new processes, this one is never replaced."""
    env = dict(os.environ)
    if env.get("SPK_BENCH_ONE_DEVICE") == "1":             # the one-GPU rehearsal shares device 0 between the ranks: the
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # only case this script sets it; RCCL's environment is the caller's
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra objects (variant, 512^2, training steps)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path on a box with fewer GPUs than ranks, together with SPK_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--no-graph", action="store_true", help="launch the step's kernels eagerly instead of replaying "
                    "the hipGraph captured from them")
    ap.add_argument("--dp-steps", type=int, default=5, help="timed steps of the data-parallel training steps (N > 1)")
    ap.add_argument("--dp-algo", default="all_reduce", choices=("all_reduce", "rs_ag"), help="the gradient exchange of the "
                    "data-parallel steps: one all-reduce per bucket, or reduce-scatter + all-gather (dp.GradBucketReducer algo)")
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps each; the line reports the median block")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))          # nothing above touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if os.environ.get("SPK_BENCH_ONE_DEVICE") == "1":      # rehearsal: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    pkg = importlib.import_module("speak-hack_amd")
    L = pkg._lib
    L.lib()
    torch.manual_seed(1 + rank)
    gen = pkg.StyleGenerator(6144).eval().to(dev)          # random-init weights of the architecture
    with torch.no_grad():                                   # default init zeroes the noise weights; wake them up
        for n, p in gen.named_parameters():
            if "noise" in n:
                p.normal_(0, 0.1)
    feats = torch.randn(BATCH, 6144, device=dev)            # synthetic latents, resident in HBM

    def eager_step():
        return gen(feats)

    # The step is ~25 launches of 10-700 us each, issued as one launch list per call; captured once into a hipGraph every
    # timed step is ONE graph launch -- the same kernels with the same arguments, and a fresh noise draw per replay, but no
    # dependence on how fast this box's host thread can issue launches.  The eager number is reported beside it.
    graph = None
    with torch.no_grad():
        if not args.no_graph:
            try:
                graph, graph_out = capture(eager_step)
            except Exception as e:                            # capture is an optimisation, never a requirement
                print(f"bench: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()

        def step():
            if graph is not None:
                graph.replay()
                return graph_out
            return eager_step()

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()

        for _ in range(args.warmup):
            step()
        # EXACTLY --steps steps per timed block, barrier + synchronize on both sides, MAX over ranks; --blocks such blocks
        # back to back, the line carries the MEDIAN block (value / ms_per_step) and every block's figure beside it
        block_s = []
        for _ in range(max(1, args.blocks)):
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            block_s.append(el)
        elapsed = sorted(block_s)[len(block_s) // 2]

        # ---- eager: the same step as an unchanged caller gets it (no graph) ----
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_eager = max(10, args.steps // 2)
        for _ in range(n_eager):
            eager_step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / n_eager * 1e3

        # ---- roofline of the dominant kernel ----
        conv_ms, n_launch, headline_up_ms = conv_time_ms(gen, feats, L)
        headline_plan = plan_of(gen)
        roof_how = ("HIP events around 20 replays of the step's hipGraph minus 20 replays of the same launch list captured "
                    "without its conv launches (spk_launch_list kind_mask)")
        if conv_ms is None:
            raise SystemExit("bench: the decoder did not run through its launch plan")

    extras = {}
    if rank == 0 and world == 1 and not args.headline_only:
        with torch.no_grad():
            # ---- the opt-in split-precision path on the headline workload ----
            gen.synthesis.precision = "bf16x3"
            try:
                gb, _ = capture(eager_step)
                bf_ms = event_ms(gb.replay, max(20, args.steps))
                cb_ms, nb, _ = conv_time_ms(gen, feats, L)
                plan_b = plan_of(gen)
                fast = [d for kind, d in plan_b.ops if kind == L.OP_CONV2D and d.flags & L.CONV_BF16X3]
                fl_fast = sum(2 * 9 * d.Cin * d.Cout * d.H * d.W * d.B for d in fast)
                _, exe_b, nw_b, _ = plan_conv_flops(plan_b, L)         # (counts a bf16x3 launch at its algorithmic FLOPs, a Winograd one at 16/36)
            finally:
                gen.synthesis.precision = "f32"
            fl_all = decoder_conv_flops(BATCH)
            extras["bf16x3"] = {
                "what": "OPT-IN: the headline workload with SynthesisNetwork.precision = 'bf16x3' -- the 3x3 convs on which it is the faster "
                        "form (32^2 and up, except the last) run on the bf16 matrix pipe with fp32 operands split into bf16 hi + lo (3 MFMAs per "
                        "product, fp32 accumulate); the 16^2 layers and the last conv (toRGB in its epilogue) stay on the exact fp32 Winograd kernel; "
                        "parity 1.5e-5 rel-L2 vs the reference golden / the oracle at B=8 (tests/test_bf16x3_gpu.py, bound 1e-3)",
                "frames_per_s_per_gpu": round(BATCH / bf_ms * 1e3, 2), "ms_per_step": round(bf_ms, 4),
                "speedup_vs_f32_headline": round((elapsed / args.steps * 1e3) / bf_ms, 3),
                "conv_launches": nb, "conv_launches_on_bf16_pipe": len(fast), "conv_ms_per_step": round(cb_ms, 4),
                "algorithmic_conv_tflops": round(fl_all / (cb_ms * 1e-3) / 1e12, 1),
                "conv_launches_on_f32_winograd": nw_b,
                "roofline": {"bound": "mfma", "kernel": "spkbf::conv3x3_bf16x3_kernel (v_mfma_f32_32x32x16_bf16 x3 per product) + the fp32 Winograd kernel "
                                                        "on the 16^2 layers and the last conv + the direct f32 kernel on the 8^2 layers",
                             "achieved": round((3 * fl_fast + (exe_b - fl_fast)) / (cb_ms * 1e-3) / 1e12, 1),
                             "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (executed: 3 bf16 MFMA FLOPs per algorithmic FLOP on the split layers, the "
                                                                    "f32 launches at their executed FLOPs; priced against the bf16 peak)",
                             "frac": round((3 * fl_fast + (exe_b - fl_fast)) / (cb_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                             "traffic": committed_traffic("bf16x3", nb)}}

            # ---- the headline workload on the DIRECT kernel everywhere (rounds 1-3's arithmetic, an fmaf chain per output) ----
            with pkg.ops.conv3x3_algo("direct"):
                gd, _ = capture(eager_step)
                d_ms = event_ms(gd.replay, max(20, args.steps))
                cd_ms, nd, _ = conv_time_ms(gen, feats, L)
                extras["f32_direct"] = {
                    "what": "the headline workload with ops.conv3x3_algo('direct') / SPK_CONV3X3_ALGO=direct: every 3x3 conv on the direct f32 MFMA "
                            "implicit-GEMM kernel (bilinear x2 folded into staging), no Winograd launch -- the arithmetic of rounds 1-3",
                    "frames_per_s_per_gpu": round(BATCH / d_ms * 1e3, 2), "ms_per_step": round(d_ms, 4),
                    "roofline": conv_roofline(plan_of(gen), L, cd_ms, 0.0, "spkconv::conv_kernel<Cfg,3,3,1,MODE> (f32 MFMA implicit GEMM, fused upsample + "
                                              "epilogue)", committed_traffic("f32_direct", nd), d_ms)}
                del gd
            # ---- the build-defined StyleGAN2 variant (modulated conv + upfirdn2d, A11) on the same workload ----
            sg2 = importlib.import_module("speak-hack_amd.stylegan2")
            gen2 = sg2.StyleGAN2Generator(6144).eval().to(dev)
            for n, p in gen2.named_parameters():
                if n.endswith("noise.weight"):
                    p.fill_(0.1)
            g2, _ = capture(lambda: gen2(feats))
            sg2_ms = event_ms(g2.replay, max(10, args.steps // 2))
            c2_ms, n2, up2_ms = conv_time_ms(gen2, feats, L)
            plan2 = plan_of(gen2)
            t2 = time.perf_counter()
            for _ in range(10):
                gen2(feats)
            torch.cuda.synchronize()
            sg2_eager = (time.perf_counter() - t2) / 10 * 1e3
            gen2.precision = "bf16x3"                     # the same opt-in split-precision path (tests/test_bf16x3_gpu.py)
            try:
                g2b, _ = capture(lambda: gen2(feats))
                sg2_bf_ms = event_ms(g2b.replay, max(10, args.steps // 2))
                del g2b
            finally:
                gen2.precision = "f32"
            extras["stylegan2_variant"] = {
                "what": "speak-hack_amd.stylegan2.StyleGAN2Generator (modulated 3x3 conv + demod -- the >= 16^2 layers on the fp32 Winograd kernel (the 16^2 ones with the contraction in 4 slices), "
                        "their x2 inputs upfirdn2d [1,3,3,1] images written by one launch each; the small layers on the direct kernel with the FIR "
                        "folded into staging --, skip toRGB with the skip upsample + add fused; same channel schedule; parity unpinned by the reference)",
                "frames_per_s_per_gpu": round(BATCH / sg2_ms * 1e3, 2), "ms_per_step": round(sg2_ms, 4), "eager_ms_per_step": round(sg2_eager, 4),
                "bf16x3_opt_in": {"ms_per_step": round(sg2_bf_ms, 4), "frames_per_s_per_gpu": round(BATCH / sg2_bf_ms * 1e3, 2)},
                "roofline": conv_roofline(plan2, L, c2_ms, up2_ms, "spkwino::wino_kernel<MOD> (fp32 Winograd, modulation on the transformed "
                                          "input, demodulation in the epilogue) + spkconv::conv_kernel<Cfg,3,3,1,MODE_(UPSAMPLE_)BATCH_SCALE> (the 4^2 - 16^2 "
                                          "layers)", committed_traffic("sg2", n2), sg2_ms)}
            del gen2, g2
            # ---- config 5: the 512^2 decoder at batch 4 ----
            s512 = pkg.SynthesisNetwork(resolution=512).eval().to(dev)
            for n, p in s512.named_parameters():
                if "noise" in n:
                    p.normal_(0, 0.1)
            w512 = torch.randn(4, 16, 512, device=dev)
            g5, _ = capture(lambda: s512(w512))
            ms5 = event_ms(g5.replay, 20)
            c5_ms, n5, up5_ms = conv_time_ms(s512, w512, L)
            r5 = conv_roofline(plan_of(s512), L, c5_ms, up5_ms, "spkwino::wino_kernel + spkconv::conv_kernel", None, ms5)
            extras["decoder_512_b4"] = {"what": "BASELINE config 5: SynthesisNetwork(resolution=512) forward, batch 4, fp32 (hipGraph replay)",
                                        "ms_per_step": round(ms5, 4), "frames_per_s_per_gpu": round(4 / ms5 * 1e3, 2),
                                        "conv_ms_per_step": round(c5_ms, 4), "conv_tflops": r5["achieved"],
                                        "conv_frac_of_f32_mfma_peak": r5["frac"], "conv_algorithmic_tflops": r5["algorithmic_tflops"],
                                        "conv_tflops_is": "executed matrix FLOPs (Winograd launches: 16/36 of algorithmic)",
                                        "upsample_ms_per_step": r5["upsample_ms_per_step"]}
            del s512, g5
        torch.cuda.empty_cache()
        # ---- the weight-gradient kernel over the decoder's 12 conv layers (what a training step's backward runs), B = 8 ----
        with torch.no_grad():
            ops = pkg.ops
            layers, r, cin = [], 8, 512
            while r <= RES:
                cout = min(int(8192 / (2.0 ** (r.bit_length() - 2))), 512)
                layers += [(cin, cout, r, True), (cout, cout, r, False)]        # (Cin, Cout, output size, reads the x2 of its input)
                cin, r = cout, r * 2
            tens = [(torch.randn(BATCH, co, rr, rr, device=dev), torch.randn(BATCH, ci, rr // 2 if up else rr, rr // 2 if up else rr, device=dev), ci, co, up)
                    for ci, co, rr, up in layers]

            def wg_pass():
                for g_, x_, ci, co, up in tens:
                    ops.conv2d_wgrad(g_, x_, co, ci, 3, 1, upsample=up)
            for _ in range(2):
                wg_pass()
            wg_ms = event_ms(wg_pass, 5)
            wg_fl = sum(2 * 9 * ci * co * g_.shape[-1] * g_.shape[-2] * BATCH for g_, x_, ci, co, up in tens)
            wg_ex = sum(2 * 9 * ci * co * g_.shape[-1] * g_.shape[-2] * BATCH *
                        (16.0 / 36.0 if pkg.ops.use_wgrad_wino(BATCH, ci, co, g_.shape[-2], g_.shape[-1]) else 1.0) for g_, x_, ci, co, up in tens)
            extras["wgrad_decoder_layers"] = {
                "what": "ops.conv2d_wgrad over the decoder's conv layers at batch 8 as a training backward calls it (fp32; Winograd "
                        "F(2x2,3x3) where the kernel serves the layer -- a x2 layer then reads the x2 image, its pass is in the time -- "
                        "the direct kernel elsewhere), kernel + slab reduce",
                "layers": len(tens), "ms": round(wg_ms, 4), "algorithmic_tflops": round(wg_fl / (wg_ms * 1e-3) / 1e12, 1),
                "executed_tflops": round(wg_ex / (wg_ms * 1e-3) / 1e12, 1),
                "frac_of_f32_mfma_peak": round(wg_ex / (wg_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                "frac_is": "EXECUTED matrix FLOPs (16/36 of algorithmic on the Winograd launches) / time"}
            del tens
        torch.cuda.empty_cache()
        # ---- config 3: the IRFD generator step at batch 16; the discriminator step at batch 8 ----
        g_ms = irfd_steps(pkg, dev, "g", 16, steps=5, warmup=4)
        extras["train_step"] = {"what": "BASELINE config 3: IRFD generator step (3 ResNet-50 encoders x 2 images, train-mode BatchNorm, "
                                        "checkpoint semantics; 2 decoder passes; reconstruction loss; fwd + bwd + global-norm clip + Adam on Gd), "
                                        "batch 16, fp32, eager launches",
                                "ms_per_step": round(g_ms, 2), "pairs_per_s": round(16 / g_ms * 1e3, 1),
                                "executed_gflop_per_pair": round(pair_executed_gflop(pkg.ops, 32), 1),
                                "executed_tflops": round(16 / g_ms * 1e3 * pair_executed_gflop(pkg.ops, 32) / 1e3, 1),
                                "frac_of_f32_mfma_peak": round(16 / g_ms * 1e3 * pair_executed_gflop(pkg.ops, 32) / 1e3 / F32_MFMA_PEAK_TFLOPS, 4),
                                "algorithmic_tflops_stored_activations": round(16 / g_ms * 1e3 * GFLOP_PER_PAIR_EXECUTED / 1e3, 1),
                                "algorithmic_tflops_reference_equivalent": round(16 / g_ms * 1e3 * GFLOP_PER_PAIR / 1e3, 1),
                                "flops_note": "frac_of_f32_mfma_peak counts EXECUTED matrix FLOPs: the encoders' activations are stored (the "
                                              "checkpoint's recompute, 6 x 10.677 of the reference's 593.5 GFLOP per pair, is not run: 529.4), and "
                                              "the decoder's 3x3 forward / data-gradient / weight-gradient launches and the encoders' 3x3 data / weight gradients on the Winograd kernels execute 16/36 of their "
                                              "algorithmic FLOPs; the two algorithmic figures price the step at the work of the direct algorithm "
                                              "without / with the reference's recompute"}
        g_bf = irfd_steps(pkg, dev, "g", 16, steps=5, warmup=4, precision="bf16x3")
        extras["train_step"]["bf16x3_opt_in"] = {
            "what": "OPT-IN ops.train_conv_precision('bf16x3'): the decoder's 3x3 convs forward and their data gradients on the bf16 pipe "
                    "(operands split hi + lo, fp32 accumulation), weight gradients and the encoders exact; gradient parity in "
                    "tests/test_bf16x3_gpu.py", "ms_per_step": round(g_bf, 2), "pairs_per_s": round(16 / g_bf * 1e3, 1)}
        g2_ms = irfd_steps(pkg, dev, "g", 16, steps=5, warmup=4, sg2=True)
        extras["stylegan2_variant"]["train_step"] = {
            "what": "BASELINE config 3 with the StyleGAN2 decoder variant as IRFD.Gd: the same generator step (encoders + 2 decoder "
                    "passes, fwd + bwd + clip + Adam), batch 16, fp32; the variant's backward runs fused (weight gradient with the "
                    "modulation / demodulation applied while staging, no materialised up(x), no ATen GEMM)",
            "ms_per_step": round(g2_ms, 2), "pairs_per_s": round(16 / g2_ms * 1e3, 1), "vs_stylegan1_decoder_step": round(g2_ms / g_ms, 3)}
        d_ms = irfd_steps(pkg, dev, "d", 8, steps=5, warmup=4)
        extras["d_step"] = {"what": "discriminator step of train.py:155-183 (4 D fwd+bwd with instance noise + BCE, 2 R1 double backward, "
                                    "Adam on D), batch 8, fp32, eager launches", "ms_per_step": round(d_ms, 2),
                            "pairs_per_s": round(8 / d_ms * 1e3, 1)}
        d_bf = irfd_steps(pkg, dev, "d", 8, steps=5, warmup=4, precision="bf16x3")
        extras["d_step"]["bf16x3_opt_in"] = {"ms_per_step": round(d_bf, 2), "pairs_per_s": round(8 / d_bf * 1e3, 1)}

        it_ms, it_pairs = train_iterations(pkg, dev, 8)
        it_bf, it_bf_pairs = train_iterations(pkg, dev, 8, precision="bf16x3")
        extras["train_iteration"] = {
            "what": "the reference's REAL iteration (train.py:150-210; config.yaml G_steps 5, r1_weight 1, stylegan_loss_weight 0.1, clip 1.0): "
                    "training.train_iteration over 5 consecutive iterations = 5 discriminator steps (instance noise, BCE, R1 double "
                    "backward, Adam on D) + 1 generator step WITH the adversarial term through model.D (fwd + bwd, clip over all "
                    "parameters, Adam on Gd), batch 8, fp32, eager launches",
            "ms_per_iteration": round(it_ms, 2), "pairs_per_s": round(it_pairs, 1),
            "bf16x3_opt_in": {"ms_per_iteration": round(it_bf, 2), "pairs_per_s": round(it_bf_pairs, 1)}}

        extras["single_frame"] = single_frame(pkg, dev, 0 if args.no_cpu_baseline else host_cores())

    line = None
    if rank == 0:
        traffic = None
        try:   # HBM bytes per conv launch from the committed PMC pass of this same command
            with open(os.path.join(ROOT, "profiles", "conv_traffic.json")) as f:
                traffic = json.load(f)["bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "generator frames/sec at 256^2, batch 8/GPU",
            "value": round(world * BATCH * args.steps / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "timed_blocks": {"what": f"{len(block_s)} timed blocks of {args.steps} steps each (barrier + synchronize around every block, max over "
                                     "ranks); value / ms_per_step are the median block's",
                             "ms_per_step": [round(b / args.steps * 1e3, 4) for b in block_s],
                             "frames_per_s_min": round(world * BATCH * args.steps / max(block_s), 2),
                             "frames_per_s_max": round(world * BATCH * args.steps / min(block_s), 2)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "StyleGAN decoder (styleganv1.StyleGenerator) 256^2 forward only, batch 8/GPU, "
                                   "fp32, random-init weights, synthetic [8,6144] latents, device-drawn noise",
                       "global_batch": BATCH * world, "resolution": RES,
                       "parallelism": f"replicas x{world} (no data-path collective)",
                       "launch": "hipGraph replay of the step" if graph is not None else "eager launches",
                       "conv3x3_algo": pkg.ops.CONV3X3_ALGO + " (fp32 Winograd F(2x2,3x3) where the kernel serves the layer, the direct fmaf-chain "
                                       "kernel elsewhere; SPK_CONV3X3_ALGO=direct: the direct kernel everywhere)"},
            "roofline": conv_roofline(headline_plan, L, conv_ms, headline_up_ms, "spkwino::wino_kernel (fp32 Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32, layers >= 16^2; the 16^2 ones run their contraction in 4 slices + the split-K finisher) + "
                                      "spkconv::conv_kernel<Cfg,3,3,1,MODE> (direct f32 MFMA implicit GEMM, the 8^2 layers)", traffic, ms_per_step,
                                      {"measured_by": roof_how}),
            "eager": {"what": "the same step without the hipGraph: one spk_launch_list call per forward (plan.DecoderPlan), as an "
                              "unchanged `model.Gd(x)` caller runs it", "ms_per_step": round(eager_ms, 4),
                      "frames_per_s_per_gpu": round(BATCH / eager_ms * 1e3, 2), "vs_graph": round(eager_ms / ms_per_step, 3)},
        }
    if world > 1 and not args.headline_only:
        # The exchange runs on a transport this build container cannot rehearse (RCCL over xGMI needs N GPUs).  If it
        # stalls, the headline measurement above must still reach the driver: a watchdog thread on every rank prints rank
        # 0's line without the data-parallel objects and ends the process when the deadline passes (a stalled collective
        # blocks inside the runtime, where no Python signal handler runs).
        import threading
        done = threading.Event()
        deadline = float(os.environ.get("SPK_BENCH_DP_DEADLINE_S", "420"))
        flag = os.path.join("/tmp", f"spk_bench_dp_failed_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")   # one node: ranks share /tmp
        if rank == 0 and os.path.exists(flag):
            os.unlink(flag)                  # a stale flag of an earlier job with the same port and launcher pid
        dist.barrier()

        def watchdog():
            # Ends the process when the deadline passes or when ANY rank has flagged a failure.  Rank 0 prints its complete
            # headline line (with the error) first; EVERY rank then leaves with a NON-ZERO code: a stalled or failed exchange
            # is a failed config 4 and the driver's rc must say so (no restart, no re-exec: this process has used the GPU).
            t_end, why = time.monotonic() + deadline, None
            while not done.wait(0.5):
                if os.path.exists(flag):
                    try:
                        why = open(flag).read()[:600] or "a rank failed"
                    except OSError:
                        why = "a rank failed"
                elif time.monotonic() > t_end:
                    why = (f"the data-parallel steps did not finish within {deadline:.0f} s (backend {args.backend}, {world} ranks, "
                           f"algo {args.dp_algo})")
                if why is not None:
                    if rank == 0:
                        line.update(extras)
                        line["train_step_dp"] = {"error": why + "; headline fields are complete; exit code 3"}
                        print(json.dumps(line), flush=True)
                    else:
                        time.sleep(2.0)          # rank 0 reports first
                    os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            if os.environ.get("SPK_BENCH_DP_FAIL_RANK") == str(rank):      # test hook (tests/test_bench_dp_gpu.py): a rank that fails
                raise RuntimeError("injected failure (SPK_BENCH_DP_FAIL_RANK)")
            # ---- config 5 per rank: the 512^2 decoder at batch 4 (replicas) ----
            with torch.no_grad():
                s512 = pkg.SynthesisNetwork(resolution=512).eval().to(dev)
                for n, p in s512.named_parameters():
                    if "noise" in n:
                        p.normal_(0, 0.1)
                w512 = torch.randn(4, 16, 512, device=dev)
                g5, _ = capture(lambda: s512(w512))
                t5 = torch.tensor([event_ms(g5.replay, 20)], device=dev, dtype=torch.float64)
                dist.all_reduce(t5, op=dist.ReduceOp.MAX)
                ms5 = float(t5.item())
                del s512, g5
            torch.cuda.empty_cache()
            dp_g = irfd_dp_steps(pkg, dev, "g", BATCH, args.dp_steps, 2, dist, rank, args.dp_algo)
            dp_d = irfd_dp_steps(pkg, dev, "d", BATCH, args.dp_steps, 2, dist, rank, args.dp_algo)
            it_ms, it_pairs = train_iterations(pkg, dev, BATCH, dist=dist, rank=rank, algo=args.dp_algo)
            if rank == 0:
                extras["decoder_512_b4"] = {"what": "BASELINE config 5: SynthesisNetwork(resolution=512) forward, batch 4 per rank, fp32 "
                                                    "(hipGraph replay, replicas; slowest rank)", "ms_per_step": round(ms5, 4),
                                            "frames_per_s": round(world * 4 / ms5 * 1e3, 2), "frames_per_s_per_gpu": round(4 / ms5 * 1e3, 2)}
                dp_g["what"] = ("BASELINE config 4: IRFD generator step (3 encoders x 2 images, 2 decoder passes, fwd + bwd + global-norm "
                                "clip + Adam on Gd), batch 8 per rank, fp32, gradients of every trained parameter exchanged by "
                                "dp.GradBucketReducer (buckets launched from backward hooks) -- slowest rank")
                dp_g["executed_tflops"] = round(dp_g["pairs_per_s"] * pair_executed_gflop(pkg.ops, 2 * BATCH) / 1e3, 1)
                dp_g["algorithmic_tflops_stored_activations"] = round(dp_g["pairs_per_s"] * GFLOP_PER_PAIR_EXECUTED / 1e3, 1)
                dp_g["algorithmic_tflops_reference_equivalent"] = round(dp_g["pairs_per_s"] * GFLOP_PER_PAIR / 1e3, 1)
                dp_d["what"] = ("discriminator step of train.py:155-183 at batch 8 per rank with the same exchange (76 MB of gradients, R1 "
                                "double backward) -- slowest rank")
                extras["train_step_dp"], extras["d_step_dp"] = dp_g, dp_d
                extras["train_iteration_dp"] = {
                    "what": "the reference's whole iteration (train.py:150-210) data-parallel: training.train_iteration with reducer_G / "
                            "reducer_D over 5 consecutive iterations (5 D steps + 1 G step with the adversarial term; D's share of the "
                            "clip norm exchanged too), batch 8 per rank -- slowest rank",
                    "ms_per_iteration": round(it_ms, 2), "pairs_per_s": round(it_pairs, 1), "algo": args.dp_algo,
                    "backend": dist.get_backend(), "world_size": world}

        except Exception as e:               # a transport error must not cost the headline record -- but it IS a failure
            import traceback
            print(f"bench: data-parallel steps failed on rank {rank}: {type(e).__name__}: {e}\n{traceback.format_exc()}",
                  file=sys.stderr, flush=True)
            try:                             # tell the other ranks' watchdogs (they may sit inside a collective)
                with open(flag, "w") as f:
                    f.write(f"rank {rank}: {type(e).__name__}: {e}"[:600])
            except OSError:
                pass
            if rank == 0:
                line.update(extras)
                line["train_step_dp"] = {"error": f"{type(e).__name__}: {e}"[:600] + "; headline fields are complete; exit code 3"}
                print(json.dumps(line), flush=True)
            else:
                time.sleep(3.0)              # rank 0's watchdog prints the line before the launcher tears the job down
            os._exit(3)                      # never rc 0, never a re-exec
        done.set()
        if rank == 0 and os.path.exists(flag):
            os.unlink(flag)

    if rank == 0:
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_cores())
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
