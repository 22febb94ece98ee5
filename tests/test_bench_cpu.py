"""bench.py's own launcher (`python bench.py --gpus N` without a torchrun environment): the parent starts the N ranks as a
child `python -m torch.distributed.run ...`, makes no GPU call itself, relays the output and exits with the child's code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_ranks_builds_the_contract_command_line(monkeypatch):
    import bench
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, timeout=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    assert "torch" not in bench.__dict__                    # the launcher module level never imports torch (no GPU call in the parent)
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert rc == 7                                           # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]          # the same flags reach every rank
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_gpus2_without_torchrun_spawns_two_ranks_and_returns_their_failure():
    """No GPU in the build container: both ranks start (the rendezvous works), refuse to run without an MI355X, and the
    launcher exits non-zero with their message -- it does not raise `launch with: python -m torch.distributed.run ...`."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():                            # on a GPU box: one card cannot host two RCCL ranks -- clear message
        assert r.returncode != 0 and "RCCL needs one GPU per rank" in r.stderr, r.stderr[-2000:]
    else:
        assert r.returncode != 0 and "needs an MI355X" in r.stderr, r.stderr[-2000:]
    assert "launching" in r.stderr and "torch.distributed.run" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())     # no JSON line from a failed run


def test_bench_n1_needs_no_launcher(monkeypatch):
    """--gpus 1 never goes through the launcher (the N = 1 path is unchanged)."""
    import bench
    called = []
    monkeypatch.setattr(bench, "launch_ranks", lambda *a, **k: called.append(a) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import torch
    if torch.cuda.is_available():
        return
    try:
        bench.main()
    except SystemExit as e:
        assert "needs an MI355X" in str(e)
    assert not called
