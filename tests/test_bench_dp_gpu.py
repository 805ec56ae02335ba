"""``python bench.py --gpus 2`` as a plain command (no launcher, no WORLD_SIZE): the parent must start fresh child ranks
itself and the line must carry the data-parallel training steps (BASELINE config 4) with a real gradient exchange.

Rehearsal on the one-GPU box: both ranks on GPU 0 (``SPK_BENCH_ONE_DEVICE=1``) exchanging over gloo -- RCCL refuses two
ranks on one device; on an 8-GPU node the same command with the default ``--backend nccl`` runs RCCL over xGMI."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks_and_times_the_gradient_exchange():
    assert torch.cuda.is_available()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SPK_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "2",
                        "--dp-steps", "2", "--dp-algo", "rs_ag"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 16 and line["value"] > 0
    assert "roofline" in line and "cpu_baseline" not in line      # the CPU leg is rank 0 at N=1 only
    assert len(line["timed_blocks"]["ms_per_step"]) == 3
    for key, nbytes in (("train_step_dp", (115.7e6 - 19.1e6) * 4), ("d_step_dp", 19.11e6 * 4)):
        d = line[key]
        assert d["backend"] == "gloo" and d["world_size"] == 2 and d["samples_per_rank"] == 8 and d["algo"] == "rs_ag"
        tl = d["bucket_timeline_rank0"]["buckets"]
        assert len(tl) == d["buckets"] and sum(t["bytes"] for t in tl) == d["grad_bytes_per_step"]
        assert all(t["wait_end_ms"] >= t["wait_begin_ms"] >= t["launch_ms"] and "gpu_done_ms" in t for t in tl)
        assert abs(d["grad_bytes_per_step"] - nbytes) < 4e6
        assert d["buckets_launched_by_hook"] + d["buckets_launched_by_finish"] == d["buckets"]
        assert d["buckets_launched_by_hook"] >= d["buckets"] - d["cold_buckets"] >= 1     # every hot bucket went out during backward
        assert d["ms_per_step"] > 0 and d["ms_per_step_no_exchange"] > 0
        assert abs(d["exposed_comm_ms"] - (d["ms_per_step"] - d["ms_per_step_no_exchange"])) < 0.02
    assert line["decoder_512_b4"]["frames_per_s"] > 0
    it = line["train_iteration_dp"]
    assert it["world_size"] == 2 and it["ms_per_iteration"] > 0 and it["algo"] == "rs_ag"
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):                                        # keep the rehearsal line (copied to profiles/ by hand)
        with open(os.path.join(out, "bench_dp_rehearsal_2ranks_gloo.json"), "w") as f:
            f.write(lines[0] + "\n")


def test_a_failing_rank_makes_the_whole_bench_exit_non_zero_and_keeps_the_headline():
    """A data-parallel leg that fails (here: rank 1 raises before its first collective, rank 0 is left waiting in one) must
    not be reported as rc 0: rank 0 still prints its complete headline line, with the error, and every rank exits non-zero."""
    assert torch.cuda.is_available()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SPK_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", SPK_BENCH_DP_FAIL_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--blocks", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["n_gpus"] == 2 and "roofline" in line
    assert "injected failure" in line["train_step_dp"]["error"] and "exit code 3" in line["train_step_dp"]["error"]
