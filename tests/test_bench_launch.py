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
    # the per-rank attribution block of the N > 1 line (bench.rank_report, the function the measured path uses)
    rk = d["ranks"]
    for f in ("step_ms", "search_ms", "exchange_ms"):
        assert len(rk[f]) == n and all(v >= 0 for v in rk[f])
    assert rk["step_ms_min"] == min(rk["step_ms"]) and rk["step_ms_max"] == max(rk["step_ms"])
    assert rk["slowest_rank"] == n - 1                  # the selftest's rank r "searches" for 10 (r + 1) ms
    assert rk["rank0_vs_slowest_ms"] == pytest.approx(rk["step_ms"][n - 1] - rk["step_ms"][0], abs=1e-3)
    # the exchange time contains the wait for the slowest rank: rank 0 waits ~10 (n - 1) ms longer than the last one
    assert rk["exchange_ms"][0] > rk["exchange_ms"][n - 1]


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


def test_self_launched_ranks_die_when_the_parent_is_killed(built):
    """SIGKILL cannot be forwarded (the -k stage of `timeout -k`): the launcher child carries PR_SET_PDEATHSIG, gets
    SIGTERM from the kernel when bench.py dies and takes its ranks down."""
    import signal
    import time
    import psutil
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                             "--selftest-launch", "--selftest-sleep", "120"],
                            cwd=ROOT, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(8)
    kids = psutil.Process(proc.pid).children(recursive=True)
    assert len(kids) >= 3
    proc.send_signal(signal.SIGKILL)
    proc.wait(timeout=30)
    gone, alive = psutil.wait_procs(kids, timeout=60)
    assert not alive, [p.cmdline()[:4] for p in alive]


def test_sighup_is_forwarded_too(built):
    import signal
    import time
    import psutil
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                             "--selftest-launch", "--selftest-sleep", "120"],
                            cwd=ROOT, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(8)
    kids = psutil.Process(proc.pid).children(recursive=True)
    proc.send_signal(signal.SIGHUP)
    assert proc.wait(timeout=60) != 0
    gone, alive = psutil.wait_procs(kids, timeout=30)
    assert not alive, [p.cmdline()[:4] for p in alive]


@pytest.mark.parametrize("where", ["init", "exchange"])
def test_rendezvous_is_bounded_and_names_the_step(built, where):
    """A rank that never reaches the process-group set-up / never joins the first collective: the waiting ranks give
    up after --rendezvous-timeout, say WHICH step on stderr, and the whole job exits non-zero without a line."""
    import time
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--selftest-launch", "--selftest-hang", where, "--rendezvous-timeout", "6"],
                         cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=240)
    assert out.returncode != 0
    assert time.monotonic() - t0 < 200
    assert "did not complete within 6 s" in out.stderr, out.stderr[-2000:]
    step = "init_process_group" if where == "init" else "first min all-reduce"
    assert step in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_parity_flags_turn_a_wrong_answer_into_a_failure():
    sys.path.insert(0, ROOT)
    import bench
    line = {"value": 1.0, "cpu_baseline": {"matches_gpu_indices": True, "kdtree": {"matches_gpu_indices": True}},
            "also": {"c3x": {"same_indices_as_c3": True}, "c2": {"unprofiled": {"same_indices": True}}},
            "exchange": {"matches_torch_all_reduce": True}}
    assert bench.parity_flags(line) == []
    line["also"]["c2"]["unprofiled"]["same_indices"] = False
    line["cpu_baseline"]["kdtree"]["matches_gpu_indices"] = False
    assert sorted(bench.parity_flags(line)) == ["also.c2.unprofiled.same_indices", "cpu_baseline.kdtree.matches_gpu_indices"]
