"""bench.py's launcher logic, without a GPU: ``python bench.py --gpus N`` (no WORLD_SIZE) must start N fresh ranks
through torch.distributed.run BEFORE any GPU call and pass their return code on."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_parent_launches_fresh_ranks_before_touching_the_gpu(monkeypatch):
    bench = _bench()
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        seen["cuda_initialised"] = torch.cuda.is_initialized()
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.delenv("SPK_BENCH_ONE_DEVICE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    try:
        bench.main()
        raise AssertionError("main() must exit with the children's return code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], os.path.join(ROOT, "bench.py"))
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert "HSA_ENABLE_IPC_MODE_LEGACY" not in seen["env"]        # RCCL ranks get the caller's environment, untouched
    assert not seen["cuda_initialised"]
    # the one-device rehearsal (all ranks on GPU 0 over gloo) is the only case that sets the IPC mode itself
    monkeypatch.setenv("SPK_BENCH_ONE_DEVICE", "1")
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_rank_process_does_not_relaunch(monkeypatch):
    """With WORLD_SIZE set (the driver's torch.distributed.run form) main() goes straight on -- and, here, stops at the
    missing device instead of spawning anything."""
    bench = _bench()
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not spawn")))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    if torch.cuda.is_available():
        return
    try:
        bench.main()
        raise AssertionError("no device: main() must stop")
    except SystemExit as e:
        assert "HIP device" in str(e.code)
