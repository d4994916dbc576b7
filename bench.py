#!/usr/bin/env python3
"""bench.py — throughput of the nearest-neighbour hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 needs one rank per GPU: when bench.py is started WITHOUT a
launcher (no RANK / WORLD_SIZE in the environment) it starts the N ranks itself — fresh child
processes under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ...`, before this process has made any GPU call — relays rank 0's JSON line and exits
with the launcher's code.  Started by that launcher directly it is one of the ranks.

Metric (BASELINE.json): query-point-pairs/s = m * n_total / t, plus the achieved fraction of
the roofline of the dominant kernel.

Workload (default `c3`, the configuration the metric is quoted on): 65536 queries x 1048576
refs x 128-D fp32, synthetic uniform [0,1) clouds from the in-repo counter-based generator
(seed 1000; queries stream then refs stream, echoing the reference driver main.cu:27-34, :54).
With N GPUs the refs are sharded 1048576 per GPU (C4 at N = 8: 65536 x 8388608 x 128) — weak
scaling; queries are replicated; the only exchange is ONE min all-reduce of m packed
(distance, index) keys: ncclAllReduce(uint64, min) issued by the library (nns_comm_allreduce_min,
the same call site as nns_search_f32_multi's), torch.distributed's all_reduce if that
communicator cannot be built.

A step = one pass of the hot path over the batch, inputs already resident in HBM:
  K2 ref pre-pass (centre + MFMA tile image + norms)  -> nns_index_refresh
  K2 query pre-pass, K3 fp32 MFMA filter, K5 finalize (prove or re-rank), exact
  re-rank of ambiguous queries                         -> nns_index_search
  [N > 1] all-reduce(min) of the keys over RCCL
  unpack keys -> int32 indices                         -> nns_keys_unpack (N = 1: nns_index_search_indices
                                                          does search + unpack; the 3-D exact kernel in one launch)
(the reference times alloc + H2D + D2H too, main.cu:73-75; the PCIe-inclusive figure of the
whole-call drop-in is reported in DESIGN.md, never here).

At N = 1 the default run also measures, after the headline and outside its timed region, the
other single-GPU configurations of BASELINE.json (C1, C2, C5) with the same fields: `also`.

At N > 1 rank 0's line additionally says where a step's time went: `ranks` (per rank, gathered outside the
timed region: wall ms per step up to the rank's own device idle, HIP-event ms of search and of the exchange,
the slowest rank and rank 0's gap to it) and `exchange.exchange_ms`.  Every rendezvous step of a launched rank
(process group, RCCL unique id, communicator, first all-reduce) runs under --rendezvous-timeout: a rank that
waits longer names the step on stderr and exits non-zero.  Every cross-check of the line (`matches_*`, `same_*`,
`verified_*`) is enforced: if one is false the line carries "parity_ok": false and the exit code is 3.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# the host driver only supports dmabuf IPC: RCCL across processes needs this (set before HIP starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA, spec (155 measured)
PEAK_F32_VALU_TFLOPS = 157.3   # fp32 vector peak (FMA-counted)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0  # bf16 dense MFMA, spec

WORKLOADS = {
    # name: (m, n_per_gpu, k, dtype)
    "c3": (65536, 1048576, 128, "f32"),     # the configuration the metric is quoted on
    "c1": (1024, 4096, 3, "f32"),           # BASELINE config 1: the reference's own CPU-runnable case
    "c2": (4096, 65536, 3, "f32"),          # per-pair exact kernel, no MFMA
    "c5": (131072, 2097152, 256, "bf16"),   # bf16 points, fp32 accumulate
    "c3s": (8192, 131072, 128, "f32"),      # quick look, not a reported config
    # EXTRA, not a BASELINE configuration and never the default: C3's fp32 points through the opt-in bf16
    # MFMA filter (NNS_FILTER_BF16) + exact fp32 re-rank — same result bits, reduced-precision GEMM
    "c3x": (65536, 1048576, 128, "f32"),
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _die_with_parent():
    """preexec of the launcher child (nothing has touched the GPU there): SIGTERM when this process dies, however
    it dies — a SIGKILL (the -k stage of `timeout -k`) cannot be forwarded, the kernel delivers this instead."""
    try:
        import ctypes
        PR_SET_PDEATHSIG = 1
        ctypes.CDLL(None, use_errno=True).prctl(PR_SET_PDEATHSIG, 15, 0, 0, 0)
    except Exception:
        pass


def self_launch(n, grace_s=20.0):
    """Started without a launcher and asked for N > 1 GPUs: become the launcher.  Nothing in this
    process has touched the GPU yet (importing torch does not), so the ranks are clean children."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")        # what torchrun would set, without its warning
    # stdout / stderr inherited: rank 0's JSON line is ours.  The launcher and its ranks get their own session, so
    # that a signal to this process takes exactly them down with it: TERM / INT / HUP are forwarded to the group;
    # if the group has not gone `grace_s` seconds after that (a rank stuck in a GPU call ignores SIGTERM) it is
    # SIGKILLed and this process exits non-zero; and if THIS process is SIGKILLed the launcher gets SIGTERM from
    # the kernel (PR_SET_PDEATHSIG) and takes its ranks down itself.
    import signal
    proc = subprocess.Popen(cmd, env=env, start_new_session=True, preexec_fn=_die_with_parent)
    state = {"sig": None, "t": None}

    def _forward(signum, _frame):
        if state["sig"] is None:
            state["sig"], state["t"] = signum, time.monotonic()
        try:
            os.killpg(proc.pid, signum)
        except ProcessLookupError:
            pass

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, _forward)
    while True:
        try:
            rc = proc.wait(timeout=0.5)
            break
        except subprocess.TimeoutExpired:
            if state["t"] is not None and time.monotonic() - state["t"] > grace_s:
                sys.stderr.write(f"bench.py: ranks still alive {grace_s:.0f} s after signal {state['sig']}: SIGKILL\n")
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.wait()
                raise SystemExit(128 + int(state["sig"]))
    if state["sig"] is not None and rc == 0:
        rc = 128 + int(state["sig"])
    raise SystemExit(rc)


class Deadline:
    """Bound one rendezvous step of a launched rank (process-group set-up, communicator creation, the first
    collective): if it has not completed after `seconds`, say WHICH step on stderr and leave with a non-zero code
    — the launcher then takes the other ranks down.  (A hung collective cannot be cancelled from inside; os._exit
    from the watchdog thread works while the main thread sits in a C call.  Nothing is re-executed.)"""

    def __init__(self, step, seconds, rank=0):
        self.step, self.seconds, self.rank, self.timer = step, seconds, rank, None

    def _fire(self):
        sys.stderr.write(f"bench.py rank {self.rank}: rendezvous step '{self.step}' did not complete within "
                         f"{self.seconds:.0f} s: giving up (exit 4)\n")
        sys.stderr.flush()
        os._exit(4)

    def __enter__(self):
        import threading
        if self.seconds and self.seconds > 0:
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


def rank_report(dist, local, rehearse_or_cpu, dev=None):
    """Outside the timed region: gather every rank's own numbers (all_gather_object) and condense them for rank 0's
    line — per-rank step time (first step start -> own device idle, before the closing barrier), its min / max /
    slowest rank, rank 0 against the slowest, and the device time of search and exchange per rank.  `local` is
    this rank's {"step_ms", "search_ms", "exchange_ms"}."""
    world = dist.get_world_size()
    got = [None] * world
    dist.all_gather_object(got, local)
    steps = [float(g["step_ms"]) for g in got]
    slow = int(np.argmax(steps))
    return {"step_ms": [round(v, 4) for v in steps], "step_ms_min": round(min(steps), 4), "step_ms_max": round(max(steps), 4),
            "slowest_rank": slow, "rank0_vs_slowest_ms": round(steps[slow] - steps[0], 4),
            "search_ms": [round(float(g["search_ms"]), 4) for g in got],
            "exchange_ms": [round(float(g["exchange_ms"]), 4) for g in got],
            "note": "per rank: wall ms per step up to the rank's own device idle (before the closing barrier); "
                    "search = K2 + filter + K5 + re-rank (HIP events inside the library); exchange = the min "
                    "all-reduce between two events on the launch stream, which includes waiting for the slowest rank"}


def cpu_baseline(orc, q_h, r_h, idx_gpu, target_s=12.0, full=False):
    """V0 on the host, single thread, on a bounded sample: `s` of the randomly chosen queries handed in
    (baseline_sample) against ALL of this GPU's refs (`full`: every one of them).  Uses the reference's own V0
    binary when it was built (oracle/_ref), else our restatement.  Also cross-checks the GPU indices."""
    n = r_h.shape[0]
    use_ref = orc.have_reference()

    def run(s):
        t0 = time.perf_counter()
        if use_ref:
            idx = orc.v0_reference(q_h[:s], r_h)
        else:
            idx, _ = orc.v0_search(q_h[:s], r_h)
        return time.perf_counter() - t0, idx

    if full:
        s = q_h.shape[0]
    else:
        t1, _ = run(1)
        s = int(max(1, min(q_h.shape[0], target_s / max(t1, 1e-6))))
    t, idx = run(s)
    ok = bool(np.array_equal(idx, idx_gpu[:s]))
    out = {"value": s * n / t, "unit": "pairs/s", "cores": 1,
           "kind": "reference" if use_ref else "port",
           "sample": f"{'all' if full and s == q_h.shape[0] else 'random'} {s} queries x all {n} refs of the workload, {t:.1f} s, V0 single thread"
                     f"{' (reference core.cu:11-54 built -O2 -ffp-contract=off)' if use_ref else ''}",
           "matches_gpu_indices": ok}
    # all host cores (OpenMP over queries, our restatement) for scale
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))           # the GPU box's CPU share for one GPU
    if cores > 1:
        s2 = int(min(q_h.shape[0], max(cores, s * cores // 2)))
        t0 = time.perf_counter()
        idx2, _ = orc.v0_search(q_h[:s2], r_h, threads=cores)
        t2 = time.perf_counter() - t0
        out["all_cores"] = {"value": s2 * n / t2, "cores": cores, "kind": "port",
                            "matches_gpu_indices": bool(np.array_equal(idx2, idx_gpu[:s2]))}
    # low-dimensional workloads: the reference's other working algorithm families (V10's CPU k-d
    # tree, core.cu:1060-1163, which itself bails out above 16 dimensions; V12's CPU octree,
    # core.cu:1454-1659, 3-D only) as further comparators; oracle/kdtree.c and oracle/octree.c are
    # exact with V0's semantics (equivalent pairs/s = m * n / t, build included)
    if q_h.shape[1] <= 16 and np.isfinite(r_h).all() and np.isfinite(q_h).all():
        trees = [("kdtree", orc.kdtree_search)]
        if q_h.shape[1] == 3 and hasattr(orc, "octree_search"):
            trees.append(("octree", orc.octree_search))
        for name, fn in trees:
            t0 = time.perf_counter()
            idx3, _ = fn(q_h, r_h, threads=cores)
            t3 = time.perf_counter() - t0
            out[name] = {"value": q_h.shape[0] * n / t3, "unit": "pairs/s (equivalent: m*n / wall time incl. build)",
                         "cores": cores, "kind": "port", "queries": int(q_h.shape[0]), "seconds": t3,
                         "matches_gpu_indices": bool(np.array_equal(idx3, idx_gpu[:q_h.shape[0]]))}
    return out


def baseline_sample(q, idx, cap=4096, seed=1000):
    """Up to `cap` RANDOMLY chosen queries of the workload (fixed seed; all of them when m <= cap) and the GPU's
    indices for them: what the CPU baseline times and cross-checks."""
    m = q.shape[0]
    if m <= cap:
        return q.float().cpu().numpy(), idx.cpu().numpy()
    sel = np.random.default_rng(seed).choice(m, cap, replace=False)      # (unsorted: any prefix of it is random too)
    sel_t = torch.from_numpy(sel).to(q.device)
    return q[sel_t].float().cpu().numpy(), idx[sel_t].cpu().numpy()


def parity_flags(node, path=""):
    """Every cross-check the line carries (`matches_*`, `same_*`, `verified_*` booleans, at any depth) that is
    False: the run then prints its line and exits non-zero."""
    bad = []
    if isinstance(node, dict):
        for k_, v in node.items():
            p_ = f"{path}.{k_}" if path else k_
            if isinstance(v, bool) and (k_.startswith("matches_") or k_.startswith("same_") or k_.startswith("verified_")):
                if not v:
                    bad.append(p_)
            else:
                bad += parity_flags(v, p_)
    elif isinstance(node, (list, tuple)):
        for i, v in enumerate(node):
            bad += parity_flags(v, f"{path}[{i}]")
    return bad


def load_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed PMC run (profiles/traffic.json,
    produced by tools/pmc_traffic.py from separate rocprofv3 --pmc passes over this same command,
    FETCH_SIZE x2-corrected per MI355X_MICROARCH.md).  NOT measured inside this run: counters cannot
    be read from within the process; `traffic_source` in the line says so.  (None, None) if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            d = json.load(f)
        v = d.get(kernel_key)
        return (v, "profiles/traffic.json[%s]: rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE per launch, collected "
                   "in separate passes over this command, not in this run" % kernel_key) if v is not None else (None, None)
    except Exception:
        return None, None


class Ctx:
    """What one rank needs to run workloads."""

    def __init__(self, pkg, dev, rank, world, dist=None, comm=None, rehearse=False, rendezvous_s=0.0):
        self.pkg, self.dev, self.rank, self.world = pkg, dev, rank, world
        self.dist, self.comm, self.rehearse, self.rendezvous_s = dist, comm, rehearse, rendezvous_s

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()


def run_workload(ctx, name, steps, warmup, seed=1000):
    """W untimed + K timed steps of workload `name`; returns (report dict, idx, q, r, keys_local_fn)."""
    pkg, dev, rank, world, dist = ctx.pkg, ctx.dev, ctx.rank, ctx.world, ctx.dist
    m, n_local, k, dtype = WORKLOADS[name]
    n_total = n_local * world

    # synthetic clouds, generated on the device (same bits as the CPU generator)
    q = torch.empty((m, k), dtype=torch.float32, device=dev)
    r = torch.empty((n_local, k), dtype=torch.float32, device=dev)
    pkg.fill_uniform(q, seed, 0)
    beg = rank * n_local                      # contiguous ref shard (core.cu:781-791)
    pkg.fill_uniform(r, seed, m * k + beg * k)
    if dtype == "bf16":                       # C5: the same clouds rounded to bf16 (RNE)
        q = q.to(torch.bfloat16)
        r = r.to(torch.bfloat16)
        torch.cuda.empty_cache()
    keys = torch.empty(m, dtype=torch.int64, device=dev)
    mixed = name == "c3x"
    ix = pkg.Index(r, index_base=beg, path="auto", profile=True, filter_bf16=mixed)

    idx_buf = torch.empty(m, dtype=torch.int32, device=dev)
    # N > 1: HIP events around the exchange of every timed step (on the launch stream = torch's current stream)
    ex_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] \
        if dist is not None else []
    ex_i = [-1]                               # index of the timed step being run (-1: warm-up)

    def step():
        ix.refresh()                          # K2 on refs
        if dist is None:                      # one shard: keys + unpacked indices from the search itself
            return ix.search_indices(q, keys, idx_buf)
        ix.search_keys(q, keys)               # K2 queries, K3 filter, K5, re-rank
        if ex_i[0] >= 0:
            ex_ev[ex_i[0]][0].record()
        pkg.allreduce_min_keys(keys, comm=ctx.comm)   # MINLOC-style exchange: one min all-reduce
        if ex_i[0] >= 0:
            ex_ev[ex_i[0]][1].record()
        return pkg.keys_unpack(keys)

    # (N > 1: the FIRST collective of the run happens here — or below if there is no warm-up — under a deadline)
    with Deadline("first min all-reduce of the packed keys", ctx.rendezvous_s if dist is not None else 0, rank):
        for _ in range(warmup):
            step()
        if dist is not None:
            if warmup == 0:
                pkg.allreduce_min_keys(keys.clone().fill_(0), comm=ctx.comm)
            torch.cuda.synchronize()
    stage = {"filter_ms": 0.0, "exact_ms": 0.0, "prep_refs_ms": 0.0, "prep_queries_ms": 0.0,
             "finalize_ms": 0.0, "rerank_ms": 0.0}
    ix.stats()                                # drop the warm-up steps' event sets
    ctx.barrier()
    t0 = time.perf_counter()
    done = 0
    idx = None
    for i in range(steps):
        ex_i[0] = i
        idx = step()
        # HIP events of every step's kernels are recorded on the launch stream inside the
        # library; read (= one device sync) every 32 steps at most: its ring of event sets
        if (i + 1) % 32 == 0 and i + 1 < steps:
            st = ix.stats()
            for f in stage:
                stage[f] += st[f] * 32
            done += 32
    local_elapsed = None
    if dist is not None:
        torch.cuda.synchronize()              # this rank's own device is idle: its own time, before the barrier
        local_elapsed = time.perf_counter() - t0
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    ex_i[0] = -1
    st = ix.stats()                           # averages over the steps not read yet
    for f in stage:
        stage[f] += st[f] * (steps - done)
    amb = st["ambiguous"]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if ctx.rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for f in stage:
        stage[f] /= max(1, steps)

    ms_per_step = elapsed / steps * 1e3
    value = m * float(n_total) / (elapsed / steps)
    mfma_path = st["path"] == 2
    bf_ops = dtype == "bf16" or mixed
    if mfma_path:
        kern_ms = stage["filter_ms"]
        flops = 2.0 * m * n_local * k       # algorithmic: 2*k flop per pair
        achieved = flops / (kern_ms * 1e-3) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if bf_ops else PEAK_F32_MFMA_TFLOPS
        traffic, tsrc = load_traffic(f"filter_{'bf16' if bf_ops else 'f32'}_{name}")
        roof = {"bound": "mfma", "kernel": f"filter_kernel<{'OpBF16' if bf_ops else 'OpF32'}> (kt={st['k_tile']})",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "kernel_ms": kern_ms, "flop_per_pair": 2 * k, "traffic": traffic}
    else:
        kern_ms = stage["exact_ms"]
        flops = 3.0 * k * m * n_local       # sub, mul, add per dim
        achieved = flops / (kern_ms * 1e-3) / 1e12
        traffic, tsrc = load_traffic(f"exact_{name}")
        # V0's sub, mul, add (core.cu:41-42) must stay three roundings — nothing may fuse — so the kernel retires ONE flop
        # per lane-instruction where the 157.3 TF vector peak counts an FMA's two: the reachable ceiling is half of it
        # (k <= 3 from 2^27 pairs — launch_k1a's rule — runs as K1f: k FMAs per pair on prepared refs + V0's own
        #  arithmetic on the 32 refs of each query's two best chunks; the line keeps V0's 3k flop per pair as the
        #  algorithmic count and the non-FMA ceiling as the yardstick of the kernel it replaced)
        k1f = k <= 3 and m >= 64 and m * n_local >= 2 ** 27
        roof = {"bound": "valu", "kernel": "lowdim_filter_kernel (K1f: VALU FMA filter + V0 re-rank)" if k1f else "exact_lane_query_kernel",
                "achieved": achieved,
                "peak": PEAK_F32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_VALU_TFLOPS,
                "ceiling": PEAK_F32_VALU_TFLOPS / 2, "ceiling_frac": achieved / (PEAK_F32_VALU_TFLOPS / 2),
                "ceiling_note": "non-FMA fp32 VALU rate (one flop per lane-instruction): V0's un-contracted arithmetic cannot use FMAs"
                                + ("; K1f evaluates V0's arithmetic only on the filter's candidates" if k1f else ""),
                "kernel_ms": kern_ms, "flop_per_pair": 3 * k, "traffic": traffic,
                "hbm_GBs_algorithmic": ((m + n_local) * k * 4 + 4 * m) / (kern_ms * 1e-3) / 1e9}
    if tsrc:
        roof["traffic_source"] = tsrc
    ranks = None
    if dist is not None:
        ex_ms = float(np.mean([a.elapsed_time(b) for a, b in ex_ev])) if ex_ev else 0.0
        ranks = rank_report(dist, {"step_ms": local_elapsed / steps * 1e3,
                                   "search_ms": stage["prep_refs_ms"] + (stage["prep_queries_ms"] + stage["filter_ms"]
                                                                        + stage["finalize_ms"] + stage["rerank_ms"]
                                                                        if mfma_path else stage["exact_ms"]),
                                   "exchange_ms": ex_ms}, ctx.rehearse)
    out = {
        "metric": "query-point-pairs/s", "value": value, "unit": "pairs/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16 points, f32 accumulate" if dtype == "bf16"
        else ("f32 points, bf16 filter operands, f32 accumulate, exact f32 re-rank (opt-in extra)" if mixed else "f32"),
        "data": "synthetic",
        "config": {"workload": f"{name}: {m} queries x {n_total} refs x {k}-D {dtype}"
                               f" ({n_local} refs per GPU x {world} GPU)",
                   "m": m, "n": n_total, "k": k, "refs_per_gpu": n_local,
                   "path": "mfma-filter+exact-rerank" if mfma_path else "exact-valu",
                   "ambiguous_queries": amb, "stage_ms": {f: round(v, 4) for f, v in stage.items()}},
        "roofline": roof,
    }
    if ranks is not None:
        out["ranks"] = ranks
    return out, idx, q, r, ix, keys


def selftest_launch(args):
    """CPU-only check of the N > 1 launch plumbing (tests/test_bench_launch.py): the ranks
    rendezvous over gloo, exchange synthetic packed keys with the product's exchange function and
    rank 0 prints one JSON line.  No GPU, no search: NOT a measurement."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    from datetime import timedelta
    with Deadline("init_process_group(gloo)", args.rendezvous_timeout, rank):
        if args.selftest_hang == "init" and rank == world - 1:
            time.sleep(3600)                  # (tests: a rank that never reaches the rendezvous)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world,
                                timeout=timedelta(seconds=max(5.0, args.rendezvous_timeout)))
    pkg = graft.load_package()
    m, n_total = 1000, 1000 * world
    beg, cnt = pkg.shard_range(n_total, world, rank)
    # rank r "finds" distance (i * 7 + r * 3) % 11 at local index i % cnt for query i
    i = torch.arange(m, dtype=torch.int64)
    d = ((i * 7 + rank * 3) % 11).to(torch.float32)
    keys = (d.view(torch.int32).to(torch.int64) << 32) | (beg + (i % cnt))
    t_begin = time.perf_counter()
    time.sleep(0.01 * (rank + 1))             # the "search": rank r takes 10 (r + 1) ms, so the last rank is the slowest
    t_search = time.perf_counter()
    with Deadline("first min all-reduce of the packed keys", args.rendezvous_timeout, rank):
        if args.selftest_hang == "exchange" and rank == world - 1:
            time.sleep(3600)                  # (tests: a rank that never joins the collective)
        pkg.allreduce_min_keys(keys)
    t_end = time.perf_counter()
    # every rank can compute the expected merge
    want = None
    for rr in range(world):
        b, c = pkg.shard_range(n_total, world, rr)
        kk = ((((i * 7 + rr * 3) % 11).to(torch.float32)).view(torch.int32).to(torch.int64) << 32) | (b + (i % c))
        want = kk if want is None else torch.minimum(want, kk)
    ok = torch.tensor([1 if torch.equal(keys, want) else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if args.selftest_sleep > 0:
        time.sleep(args.selftest_sleep)       # (tests/test_bench_launch.py: signal forwarding)
    # the per-rank attribution block of the N > 1 line, through the same function the measured path uses
    ranks = rank_report(dist, {"step_ms": (t_search - t_begin) * 1e3, "search_ms": (t_search - t_begin) * 1e3,
                               "exchange_ms": (t_end - t_search) * 1e3}, True)
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "ok": bool(ok.item()), "ranks": ranks,
                          "note": "gloo rendezvous + key exchange only: not a measurement"}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the C1 / C2 / C5 block of the default N = 1 run")
    # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (NOT a measurement):
    # every rank on cuda:0, keys exchanged through gloo on the host instead of RCCL
    ap.add_argument("--rehearse-one-gpu", action="store_true")
    ap.add_argument("--verify", action="store_true", help="rank 0 checks the merged indices against the unsharded search")
    ap.add_argument("--exchange", default="auto", choices=("auto", "library", "torch"),
                    help="N > 1: who issues the RCCL min all-reduce (auto: the library, torch.distributed if that fails)")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--selftest-sleep", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--selftest-hang", default="", choices=("", "init", "exchange"), help=argparse.SUPPRESS)
    ap.add_argument("--rendezvous-timeout", type=float, default=240.0,
                    help="N > 1: seconds a rank waits in process-group set-up, communicator creation or the first "
                         "collective before it names the step on stderr and exits non-zero (0: unbounded)")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        self_launch(args.gpus)                # never returns
    if args.selftest_launch:
        return selftest_launch(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    if args.rehearse_one_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible); "
                         "one rank per GPU (use --rehearse-one-gpu to rehearse the code path on fewer)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if launched:                              # one rank per GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        from datetime import timedelta
        pg_to = timedelta(seconds=max(30.0, args.rendezvous_timeout)) if args.rendezvous_timeout > 0 else None
        with Deadline("torch.distributed.init_process_group", args.rendezvous_timeout, rank):
            if args.rehearse_one_gpu:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world, **({"timeout": pg_to} if pg_to else {}))
            else:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev,
                                        **({"timeout": pg_to} if pg_to else {}))

    pkg = graft.load_package()

    # ---- the exchange of the N > 1 form ------------------------------------------------------------
    comm, exch = None, None
    if dist is not None:
        exch = {"impl": "torch.distributed all_reduce(int64, MIN) "
                        + ("over gloo (rehearsal)" if args.rehearse_one_gpu else "over RCCL")}
        if not args.rehearse_one_gpu and args.exchange != "torch":
            # rank 0 draws RCCL's unique id, torch.distributed carries its 128 bytes, every rank joins
            uid = [None]
            if rank == 0:
                try:
                    uid[0] = pkg.comm_unique_id()
                except pkg.NNSError as e:
                    exch["library_comm_error"] = str(e)
            with Deadline("broadcast of the RCCL unique id (first torch.distributed collective)", args.rendezvous_timeout, rank):
                dist.broadcast_object_list(uid, src=0, device=dev)
            t_comm = time.perf_counter()
            if uid[0] is not None:
                try:
                    with Deadline("nns_comm_create (ncclCommInitRank)", args.rendezvous_timeout, rank):
                        comm = pkg.Comm(uid[0], world, rank, local_rank)
                except pkg.NNSError as e:
                    comm = None
                    exch["library_comm_error"] = str(e)
            t_comm = time.perf_counter() - t_comm
            ok = torch.tensor([1 if comm is not None else 0], device=dev)
            with Deadline("agreement on the communicator (all_reduce)", args.rendezvous_timeout, rank):
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None
            if comm is not None:
                exch = {"impl": "nns_comm_allreduce_min: ncclAllReduce(ncclUint64, ncclMin) issued by libnns_mi355x "
                                "(the call site nns_search_f32_multi uses)", "rccl_ranks": comm.size(),
                        "comm_create_s": round(t_comm, 3)}
            elif args.exchange == "library":
                raise SystemExit(f"--exchange library: {exch.get('library_comm_error', 'communicator not built')}")

    ctx = Ctx(pkg, dev, rank, world, dist, comm, args.rehearse_one_gpu, args.rendezvous_timeout)
    out, idx, q, r, ix, keys = run_workload(ctx, args.workload, args.steps, args.warmup)
    m, n_local, k, dtype = WORKLOADS[args.workload]
    n_total = n_local * world

    if dist is not None:
        # outside the timed region: both exchange routes must give the same bits
        loc = ix.search_keys(q).clone()
        a = loc.clone()
        pkg.allreduce_min_keys(a, comm=comm)
        b = loc.clone()
        pkg.allreduce_min_keys(b)             # torch.distributed
        torch.cuda.synchronize()
        same = torch.tensor([1 if torch.equal(a, b) else 0], device="cpu" if args.rehearse_one_gpu else dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        exch["matches_torch_all_reduce"] = bool(int(same.item()))
        # a rank's own keys must survive the merge somewhere (sanity: shards really differ)
        won = torch.tensor([int((a == loc).sum().item())], device="cpu" if args.rehearse_one_gpu else dev)
        dist.all_reduce(won, op=dist.ReduceOp.SUM)
        exch["queries_won_sum_over_ranks"] = int(won.item())   # >= m (== m without cross-shard exact ties)

    if rank == 0:
        if exch is not None:
            if "ranks" in out:                 # the attribution block: where a step's time went, per rank
                exch["exchange_ms"] = {"max_over_ranks": max(out["ranks"]["exchange_ms"]),
                                       "min_over_ranks": min(out["ranks"]["exchange_ms"]),
                                       "payload_bytes": 8 * m,
                                       "note": "HIP events around the all-reduce on the launch stream, mean over the timed "
                                               "steps; the minimum over ranks is the slowest rank's own cost of the collective, "
                                               "the others' excess is time spent waiting for it"}
            out["exchange"] = exch
        if args.rehearse_one_gpu:
            out["rehearsal"] = "all ranks on one GPU, gloo exchange: NOT a measurement"
        if args.verify:
            # the merged answer must equal one unsharded search over all n_total refs
            r_all = torch.empty((n_total, k), dtype=torch.float32, device=dev)
            pkg.fill_uniform(r_all, 1000, m * k)
            if dtype == "bf16":
                r_all = r_all.to(torch.bfloat16)
            ix_all = pkg.Index(r_all)
            want = ix_all.search(q)
            torch.cuda.synchronize()
            out["verified_vs_unsharded"] = bool(torch.equal(want, idx))
            ix_all.close()
            del r_all
        orc = None
        if world == 1 and not args.no_cpu_baseline:
            orc = graft.load_oracle()          # cpu_baseline leg only
            qs, is_ = baseline_sample(q, idx)
            out["cpu_baseline"] = cpu_baseline(orc, qs, r.float().cpu().numpy(), is_)
    ix.close()
    del q, r, keys, ix
    torch.cuda.empty_cache()

    # ---- the other single-GPU configurations of BASELINE.json, same fields, same run ------------------
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_also:
        also = {}
        idx_c3 = idx.clone()                  # the headline run's answers (for the c3x entry)
        for name, (st_, wu_) in (("c2", (200, 20)), ("c5", (5, 1)), ("c1", (200, 20)), ("c3x", (10, 2))):
            o, idx2, q2, r2, ix2, keys2 = run_workload(ctx, name, st_, wu_)
            ent = {"value": o["value"], "unit": "pairs/s", "ms_per_step": o["ms_per_step"], "steps": st_, "warmup": wu_,
                   "dtype": o["dtype"], "config": o["config"], "roofline": o["roofline"]}
            if name == "c3x":
                # the opt-in NNS_FILTER_BF16 form of the HEADLINE workload: fp32 points, bf16 filter operands, the
                # same exact fp32 re-rank — reported beside the headline, never instead of it
                ent["note"] = ("opt-in extra, not the headline: same inputs as c3, filter on bf16 operands "
                               "(wider tau, more candidates), answers re-ranked with V0's fp32 arithmetic")
                ent["same_indices_as_c3"] = bool(torch.equal(idx2, idx_c3))
            if name in ("c1", "c2"):
                # launch-bound shapes: the same 200 back-to-back searches WITHOUT the per-launch HIP events of the
                # profiled loop above (an event record between two launches keeps the next kernel's head from
                # overlapping the previous one's tail) — SURVEY 8(d) defines C2's time this way
                ix_np = pkg.Index(r2, path="auto", profile=False)
                idx_np = torch.empty(o["config"]["m"], dtype=torch.int32, device=dev)
                for _ in range(wu_):
                    ix_np.search_indices(q2, keys2, idx_np)
                torch.cuda.synchronize()
                t_np = time.perf_counter()
                for _ in range(st_):
                    ix_np.search_indices(q2, keys2, idx_np)
                torch.cuda.synchronize()
                dt_np = (time.perf_counter() - t_np) / st_
                ent["unprofiled"] = {"ms_per_step": dt_np * 1e3, "value": o["config"]["m"] * float(o["config"]["n"]) / dt_np,
                                     "unit": "pairs/s", "same_indices": bool(torch.equal(idx_np, idx2)),
                                     "note": "no HIP events between launches; not the line's value"}
                ix_np.close()
            if name != "c3x" and orc is not None:
                full = name == "c1"            # C1 is the reference's CPU-runnable case: V0 over the whole problem
                qs, is_ = baseline_sample(q2, idx2)
                ent["cpu_baseline"] = cpu_baseline(orc, qs, r2.float().cpu().numpy(), is_, target_s=4.0, full=full)
            also[name] = ent
            ix2.close()
            del q2, r2, keys2, ix2
            torch.cuda.empty_cache()
            pkg.trim()
        out["also"] = also

    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:                             # last thing on stdout (RCCL prints a banner at init)
        bad = parity_flags(out)
        out["parity_ok"] = not bad
        if bad:
            out["parity_failed"] = bad
        print(json.dumps(out), flush=True)
        if bad:                               # a wrong answer must not look like a successful run
            sys.stderr.write("bench.py: cross-checks FAILED: " + ", ".join(bad) + "\n")
            raise SystemExit(3)


if __name__ == "__main__":
    main()
