#!/usr/bin/env python3
"""Headline benchmark: generator frames/s at 256^2, batch 8 per GPU (BASELINE.json metric,
configs[1]: StyleGAN synthesis forward only, batch 8, 1x MI355X).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one ``StyleGenerator.forward`` (mapping + 13 style affines + prologue + 12 fused 3x3 conv
launches + toRGB) over one batch of 8 synthetic [6144] latents, inputs and weights resident in HBM,
noise drawn on the device inside the step as the reference does (styleganv1.py:455).  fp32
end to end (exact-f32 MFMA).  N > 1: launched by torch.distributed.run, one rank per GPU, each rank
an independent replica on its own batch (the forward path has no exchange step -- SURVEY.md 8e);
barrier + synchronize on both sides of the timed region, MAX over ranks, rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the dominant kernel (spkconv::conv_kernel, MFMA-bound): algorithmic conv FLOPs per
                  step / conv kernel time per step, timed live with HIP events on the launch stream.
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm, parity-pinned to it) timed
                  on this box's host cores on the same B=8 workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 8
RES = 256
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def decoder_conv_flops(batch, resolution=RES):
    """Algorithmic FLOPs (2*MAC) of the twelve 3x3 convs per step -- SURVEY.md 8(d): 56.17 GFLOP/frame."""
    total, res, cin = 0, 8, 512
    while res <= resolution:
        cout = min(int(8192 / (2.0 ** (res.bit_length() - 2))), 512)
        total += 2 * 9 * res * res * (cin * cout + cout * cout)
        cin, res = cout, res * 2
    return total * batch


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path on a box with fewer GPUs than ranks, together with SPK_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--no-graph", action="store_true", help="launch the step's kernels one by one instead of replaying "
                    "the hipGraph captured from them")
    args = ap.parse_args()

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
    pkg._lib.lib()
    torch.manual_seed(1 + rank)
    gen = pkg.StyleGenerator(6144).eval().to(dev)          # random-init weights of the architecture
    with torch.no_grad():                                   # default init zeroes the noise weights; wake them up
        for n, p in gen.named_parameters():
            if "noise" in n:
                p.normal_(0, 0.1)
    feats = torch.randn(BATCH, 6144, device=dev)            # synthetic latents, resident in HBM

    def eager_step():
        return gen(feats)

    # The step is ~35 launches of 30-700 us each; captured once into a hipGraph (the C ABI neither allocates nor
    # synchronises, the noise draw is graph-safe Philox), every timed step is ONE graph launch -- the same kernels
    # with the same arguments, and a fresh noise draw per replay, but no dependence on how fast this box's host
    # thread can issue launches.
    graph = None
    if not args.no_graph:
        try:
            with torch.no_grad():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        eager_step()                      # warm every cache (packed weights, workspace) before capture
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    graph_out = eager_step()
            torch.cuda.synchronize()
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

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # ---- roofline of the dominant kernel: time of the step's conv launches, with HIP events on the launch stream ----
        n_launch = 2 * len(gen.synthesis.layers)
        roof_how = None
        if graph is not None:
            # Events between individual launches would sit on the GPU timeline themselves (an event record is a marker
            # packet with a cache flush: ~30 us per conv when launches are queued back to back) and, launch by launch,
            # would also count this box's host latency.  Instead the SAME step is captured a second time without its
            # conv launches, and both graphs are timed by HIP events around R replays: conv time = the difference.
            real_launch = pkg.ops._launch_conv2d
            try:
                pkg.ops._launch_conv2d = lambda desc: None      # measurement only, and only here: the package has no switch
                graph_nc = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_nc):
                    eager_step()
                pkg.ops._launch_conv2d = real_launch
                R = 20

                def timed(g):
                    g.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(R):
                        g.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / R

                conv_ms = timed(graph) - timed(graph_nc)
                roof_how = "HIP events around 20 replays of the step's hipGraph minus 20 replays of the same graph captured without its conv launches"
            except Exception as e:
                print(f"bench: conv-less capture failed ({type(e).__name__}: {e}); timing conv launches with event pairs", file=sys.stderr)
            finally:
                pkg.ops._launch_conv2d = real_launch
        if roof_how is None:
            events, real_launch = [], pkg.ops._launch_conv2d

            def timed_launch(desc):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                real_launch(desc)
                e1.record()
                events.append((e0, e1))

            prof_steps = 5
            try:
                pkg.ops._launch_conv2d = timed_launch
                for _ in range(prof_steps):
                    eager_step()
            finally:
                pkg.ops._launch_conv2d = real_launch
            torch.cuda.synchronize()
            conv_ms = sum(a.elapsed_time(b) for a, b in events) / prof_steps
            n_launch = len(events) // prof_steps
            roof_how = "HIP event pair around every conv launch of 5 eager steps"

        # ---- the build-defined StyleGAN2 variant (modulated conv + upfirdn2d, A11) on the same workload ----
        sg2 = importlib.import_module("speak-hack_amd.stylegan2")
        gen2 = sg2.StyleGAN2Generator(6144).eval().to(dev)
        for n, p in gen2.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1)
        for _ in range(max(2, args.warmup // 2)):
            gen2(feats)
        torch.cuda.synchronize()
        step2 = lambda: gen2(feats)
        if graph is not None:                             # same launch mode as the headline
            try:
                graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph2):
                    gen2(feats)
                step2 = graph2.replay
            except Exception as e:
                print(f"bench: hipGraph capture of the StyleGAN2 variant failed ({e}); eager", file=sys.stderr)
            torch.cuda.synchronize()
        step2()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sg2_steps = max(5, args.steps // 2)
        for _ in range(sg2_steps):
            step2()
        torch.cuda.synchronize()
        sg2_ms = (time.perf_counter() - t1) / sg2_steps * 1e3
        del gen2

    if rank == 0:
        traffic = None
        try:   # HBM bytes per conv launch from the committed PMC pass of this same command
            with open(os.path.join(ROOT, "profiles", "conv_traffic.json")) as f:
                traffic = json.load(f)["bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        flops = decoder_conv_flops(BATCH)
        achieved = flops / (conv_ms * 1e-3) / 1e12
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "generator frames/sec at 256^2, batch 8/GPU",
            "value": round(world * BATCH * args.steps / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "StyleGAN decoder (styleganv1.StyleGenerator) 256^2 forward only, batch 8/GPU, "
                                   "fp32, random-init weights, synthetic [8,6144] latents, device-drawn noise",
                       "global_batch": BATCH * world, "resolution": RES,
                       "parallelism": f"replicas x{world} (no data-path collective)",
                       "launch": "hipGraph replay of the step" if graph is not None else "eager launches"},
            "roofline": {"bound": "mfma", "kernel": "spkconv::conv_kernel<Cfg,3,3,1,MODE> (f32 MFMA implicit GEMM, fused upsample + epilogue)",
                         "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "launches_per_step": n_launch, "conv_ms_per_step": round(conv_ms, 4),
                         "conv_share_of_step": round(conv_ms / ms_per_step, 3),
                         "algorithmic_gflop_per_step": round(flops / 1e9, 2), "measured_by": roof_how},
        }
        line["stylegan2_variant"] = {"what": "speak-hack_amd.stylegan2.StyleGAN2Generator (modulated 3x3 conv + demod, upfirdn2d "
                                             "[1,3,3,1], skip toRGB; same channel schedule; parity unpinned by the reference)",
                                     "frames_per_s_per_gpu": round(BATCH / sg2_ms * 1e3, 2), "ms_per_step": round(sg2_ms, 4)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_cores())
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
