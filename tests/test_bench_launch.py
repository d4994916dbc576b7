"""bench.py must be startable as the driver starts it — `python bench.py --gpus N` with no
launcher — and bring up its own N ranks (fresh children under torch.distributed.run, before any
GPU call).  No GPU here: `--selftest-launch` makes the ranks rendezvous over gloo, push synthetic
packed keys through the product's exchange function (host.allreduce_min_keys) and print the one
JSON line, which exercises exactly the launch path the N > 1 bench takes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                        "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    return env


@pytest.mark.parametrize("n", [2, 3])
def test_bench_self_launches_its_ranks(built, n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1",
                          "--warmup", "0", "--selftest-launch"],
                         cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["selftest"] == "launch" and d["n_gpus"] == n and d["ok"] is True


def test_bench_propagates_a_failing_rank(built):
    """Without a GPU the real N > 1 bench cannot run: the ranks exit with the product's 'needs a HIP
    device' message and the parent must hand the non-zero code back (never a silent success)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("checks the no-device error path")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs a HIP device" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_self_launched_ranks_die_with_the_parent(built):
    """A driver that times out sends SIGTERM to `python bench.py --gpus N`: the launcher and its ranks (their own
    process group) must go down with it, not linger holding GPUs."""
    import signal
    import time
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                             "--selftest-launch", "--selftest-sleep", "60"],
                            cwd=ROOT, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(8)                                     # ranks are up and sleeping
    import psutil
    kids = psutil.Process(proc.pid).children(recursive=True)
    assert len(kids) >= 3                             # the launcher + 2 ranks
    proc.send_signal(signal.SIGTERM)
    proc.wait(timeout=60)
    gone, alive = psutil.wait_procs(kids, timeout=30)
    assert not alive, [p.cmdline()[:4] for p in alive]
