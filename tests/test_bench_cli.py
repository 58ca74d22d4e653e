"""bench.py command line (no GPU): `--gpus N` never lets one process stand for N ranks."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_n_without_torchrun_spawns_n_ranks(monkeypatch):
    """`python bench.py --gpus 2` (the driver's form, no WORLD_SIZE): the parent starts 2 worker
    ranks through torch.distributed.run before touching a GPU and exits with their status."""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()

    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    import torch
    monkeypatch.setattr(torch.cuda, "is_available",
                        lambda: pytest.fail("the parent must not touch the GPU"))
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert e.value.code == 7                      # the children's status
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    k = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[k + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_world_size_mismatch_is_refused(monkeypatch):
    """Under a launcher, --gpus must equal the number of ranks actually running."""
    import torch
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    import tensorflowraytrace_amd.distributed as tdist
    monkeypatch.setattr(tdist, "init_from_env", lambda backend=None: (0, 1, 0))
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"])
    assert "--gpus 8" in str(e.value.code)


def test_single_gpu_needs_a_gpu(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(SystemExit) as e:
        bench.main([])
    assert "no CPU fallback" in str(e.value.code)
