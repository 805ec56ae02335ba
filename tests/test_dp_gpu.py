"""BASELINE config 4 on the HIP path: the REAL IRFD generator step (three ResNet-50 encoders, train-mode BatchNorm,
two decoder passes, reconstruction loss, backward) on two ranks through ``dp.GradBucketReducer``.

A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so both ranks share GPU 0 and
exchange over ``gloo`` -- the reducer, its hooks, the bucket order and the HIP gradients are exactly what runs over
RCCL; only the transport differs.  The ranks are FRESH child processes (``python -m torch.distributed.run``); this
process never re-executes itself."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(tmp_path, *extra):
    assert torch.cuda.is_available()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "workers", "dp_irfd_worker.py"), "--out", str(tmp_path),
           "--backend", "gloo", "--one-device", "--steps", "2", *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return (torch.load(tmp_path / "rank0.pt", weights_only=False), torch.load(tmp_path / "rank1.pt", weights_only=False))


def _check_exchange(r0, r1, never_reached=()):
    """Reduced gradients are bit-identical on both ranks and equal the mean of the two shards' own gradients; collectives
    went out in bucket order, each bucket once."""
    for step in (0, 1):
        a, b = r0["steps"][step], r1["steps"][step]
        assert a["loss"] != b["loss"]                                  # different shards ...
        assert abs(a["loss"] - a["loss_local"]) <= 1e-6 * abs(a["loss"])   # ... the exchange does not touch the forward
        assert a["norm"] == b["norm"]
        for name, ga, gb, la, lb in zip(r0["names"], a["reduced"], b["reduced"], a["local"], b["local"]):
            if name.startswith(never_reached):                         # no gradient on any rank: zeros after the exchange
                assert la is None and lb is None and float(ga.abs().max()) == 0.0 and float(gb.abs().max()) == 0.0
                continue
            assert la is not None and lb is not None, name
            assert torch.equal(ga, gb), name                           # ranks hold bit-identical gradients
            mean = (la.double() + lb.double()) / 2                     # what DDP's averaging leaves: mean of the shard gradients
            err = float((ga.double() - mean).norm() / mean.norm().clamp_min(1e-30))
            assert err < 1e-6, (name, err)                             # one fp32 add + one fp32 multiply per element
        assert a["by_hook"] + [i for i in a["by_finish"]] == list(range(a["n_buckets"])), (a["by_hook"], a["by_finish"])
        assert a["by_hook"] == b["by_hook"] and a["by_finish"] == b["by_finish"]
        assert a["rebuilt_after"]


def test_irfd_generator_step_two_ranks_through_the_reducer(tmp_path):
    """BASELINE config 4's per-rank batch: 8 samples per rank."""
    r0, r1 = _run_ranks(tmp_path, "--batch", "8")
    assert r0["names"] == r1["names"] and len(r0["names"]) == 3 * 159 + 83 + 2
    assert r0["bytes_per_step"] == sum(t.numel() * 4 for t in r0["steps"][0]["reduced"] if t is not None)
    assert abs(r0["bytes_per_step"] - (115.7e6 - 19.1e6) * 4) < 4e6          # everything but D: ~386 MB of gradients
    _check_exchange(r0, r1, never_reached=("Cm.",))                     # Cm: never reached by the reconstruction loss
    s1 = r0["steps"][1]
    # step 1 runs on buckets rebuilt in autograd-completion order: the only bucket left for finish() is the cold one
    # (Cm: no gradient in a reconstruction-only step) -- every other bucket's all-reduce went out from a hook, i.e. while
    # backward was still running
    assert s1["cold"] == [s1["n_buckets"] - 1] and set(s1["cold_names"]) == {"Cm.weight", "Cm.bias"}
    assert s1["by_hook"] == list(range(s1["n_buckets"] - 1)) and s1["by_finish"] == [s1["n_buckets"] - 1]
    assert s1["n_buckets"] >= 12                                       # 386 MB in 32 MiB buckets


def test_discriminator_step_two_ranks_through_the_reducer(tmp_path):
    """The D step's exchange (train.py:155-183): 4 D forwards with instance noise, BCE, the R1 double backward; 76 MB of
    spectral-norm ``weight_orig`` / bias gradients at 8 samples per rank."""
    r0, r1 = _run_ranks(tmp_path, "--batch", "8", "--which", "d")
    assert r0["names"] == r1["names"] and all(n.startswith("D.") for n in r0["names"])
    assert abs(r0["bytes_per_step"] - 19.11e6 * 4) < 1e6
    _check_exchange(r0, r1)
    s1 = r0["steps"][1]
    assert s1["cold"] == [] and s1["by_hook"] == list(range(s1["n_buckets"])) and s1["by_finish"] == []
    assert s1["n_buckets"] >= 3


def test_generator_step_with_gradient_accumulation_two_ranks(tmp_path):
    """``gradient_accumulation_steps = 2`` (train.py:152,335) on the HIP path: two micro-batches of 2 samples per rank, the
    first accumulates locally (hooks disarmed), the second exchanges the sums."""
    r0, r1 = _run_ranks(tmp_path, "--batch", "4", "--accum", "2")
    _check_exchange(r0, r1, never_reached=("Cm.",))
    s1 = r0["steps"][1]
    assert s1["by_hook"] == list(range(s1["n_buckets"] - 1)) and s1["by_finish"] == [s1["n_buckets"] - 1]


@pytest.mark.parametrize("algo", ["all_reduce", "rs_ag"])
def test_training_iteration_two_ranks_equals_the_single_process_iteration(tmp_path, algo):
    """``training.train_iteration(..., reducer_G, reducer_D)`` -- the reference's whole iteration (train.py:150-210) under
    data parallelism -- on two ranks of the REAL model on the HIP path: three iterations (D, G | D | D, G).  The replicas
    end bit-identical (parameters AND every gradient, D's included: the global-norm clip of train.py:208 runs over both
    reducers' exchanged buffers), and equal ONE process stepping on the mean of the two shards' gradients, with each rank's
    own RNG streams and BatchNorm buffers played back (tests/dp_emulation.py)."""
    import importlib
    from dp_emulation import emulate
    sys.path.insert(0, os.path.join(ROOT, "tests", "workers"))
    W = importlib.import_module("dp_iteration_worker")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "workers", "dp_iteration_worker.py"), "--out", str(tmp_path),
           "--backend", "gloo", "--one-device", "--batch", "2", "--algo", algo]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    r0, r1 = (torch.load(tmp_path / f"rank{i}.pt", weights_only=False) for i in (0, 1))
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
        g0, g1 = r0["grads"][k], r1["grads"][k]
        assert (g0 is None) == (g1 is None) and (g0 is None or torch.equal(g0, g1)), k
    assert [x["loss_G"] is not None for x in r0["log"]] == [True, False, True]
    assert r0["log"][0]["loss_D"] != r1["log"][0]["loss_D"]                      # different shards, different draws
    assert any(not torch.equal(r0["buffers"][k], r1["buffers"][k]) for k in r0["buffers"] if k.endswith("running_mean"))
    for k in r0["buffers"]:
        if k.endswith(("weight_u", "weight_v")):
            assert torch.equal(r0["buffers"][k], r1["buffers"][k]), k
    # ---- the same three iterations in ONE process on the mean of the shards' gradients ----
    import model as M
    dev = torch.device("cuda:0")
    net = M.IRFD().to(dev).train()
    net.load_state_dict(torch.load(tmp_path / "init.pt", map_location=dev))
    init = {k: v.detach().clone() for k, v in net.named_parameters()}
    opt_G, opt_D = W.optimizers(net)
    shards = [W.shard(r, 2, dev) for r in (0, 1)]
    log, states = emulate(net, shards, opt_G, opt_D, W.STEPS, seeds=(W.SEED0, W.SEED0 + 1), device=dev, **W.KW)
    for it, rec in enumerate(log):
        for r, rr in enumerate((r0, r1)):
            assert abs(rec["loss_D"][r] - rr["log"][it]["loss_D"]) <= 2e-4 * max(1.0, abs(rec["loss_D"][r])), (it, r, rec, rr["log"][it])
            if rr["log"][it]["loss_G"] is not None:
                assert abs(rec["loss_G"][r] - rr["log"][it]["loss_G"]) <= 2e-4 * max(1.0, abs(rec["loss_G"][r])), (it, r, rec, rr["log"][it])
    # gradients as they stand after the last iteration (encoders / Cm: accumulated over both G steps and clipped twice; Gd:
    # the last G step's; D: the last D step's mean plus the mean of the G step's contribution, clipped)
    # Criterion.  The two runs execute the same kernels on the same operands except for ONE number: the clip coefficient (a
    # multi-tensor norm over flat buffers here, torch's per-tensor norm there) differs in its last bits, so from the first G step on
    # every parameter differs by ~1e-7 relative.  That cannot move a gradient -- except through a LeakyReLU mask whose pre-activation
    # lies within 1e-7 of zero: among the ~1e7 activations of a D pass about one does.  A flipped mask at a layer with P = batch x
    # pixels positions changes the gradient field behind it over that position's receptive field, i.e. every weight gradient
    # UPSTREAM of it by ~1/P of its norm -- with 2 samples P is 128 at the 8^2 block and 32 at the 4^2 one.  Measured (deterministic,
    # the same in every run of one build): median 4.8e-6, p90 8.0e-6 over the 594 tensors, and ~20 tensors of D between 2e-3 and 4e-3.
    # So: the bulk must agree to rounding, and no tensor may be off by more than one flip at the deepest layer explains (1/32).
    errs = {}
    for k, p in net.named_parameters():
        g = r0["grads"][k]
        assert (p.grad is None) == (g is None), k
        if g is not None and float(g.norm()) > 0:
            errs[k] = rel_l2(p.grad.cpu(), g)
    vals = sorted(errs.values())
    print(f"gradient rel-L2 against the emulation over {len(vals)} tensors: median {vals[len(vals) // 2]:.2e}, p90 {vals[int(0.9 * len(vals))]:.2e}, "
          f"p99 {vals[int(0.99 * len(vals))]:.2e}, max {vals[-1]:.2e}")
    assert vals[len(vals) // 2] < 2e-5, vals[len(vals) // 2]                       # median: rounding
    assert vals[int(0.9 * len(vals))] < 2e-4, vals[int(0.9 * len(vals))]           # 90 % of the tensors: rounding
    assert vals[-1] < 1.0 / 32, max(errs.items(), key=lambda t: t[1])              # the rest: at most one flipped mask's worth
    # the Adam updates: compare what the steps did to each stepped tensor (delta from the initial weights)
    for k, p in net.named_parameters():
        if k.startswith(("Gd.", "D.")):
            d_ref, d_got = (p.detach() - init[k]).cpu(), r0["params"][k] - init[k].cpu()
            if float(d_ref.norm()) > 0:
                assert rel_l2(d_got, d_ref) < 2e-2, (k, rel_l2(d_got, d_ref))
        else:
            assert torch.equal(p.detach().cpu(), r0["params"][k]), k                # encoders / Cm: never stepped (train.py:346-347)
    tl = r0["timeline_G"] + r0["timeline_D"]
    assert tl and all(t["wait_end_ms"] >= t["wait_begin_ms"] >= t["launch_ms"] >= 0 for t in tl)
