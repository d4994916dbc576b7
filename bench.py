#!/usr/bin/env python3
"""bench.py — throughput of the nearest-neighbour hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): query-point-pairs/s = m * n_total / t, plus the achieved
fraction of the fp32-MFMA roofline of the dominant kernel.

Workload (default `c3`, the configuration the metric is quoted on): 65536 queries x
1048576 refs x 128-D fp32, synthetic uniform [0,1) clouds from the in-repo
counter-based generator (seed 1000; queries stream then refs stream, echoing the
reference driver main.cu:27-34, :54).  With N GPUs the refs are sharded 1048576 per
GPU (C4 at N = 8: 65536 x 8388608 x 128) — weak scaling; queries are replicated; the
only exchange is ONE min all-reduce of m packed (distance, index) keys (RCCL).

A step = one pass of the hot path over the batch, inputs already resident in HBM:
  K2 ref pre-pass (centre + MFMA tile image + norms)  -> nns_index_refresh
  K2 query pre-pass, K3 fp32 MFMA filter, K5 finalize (prove or re-rank), exact
  re-rank of ambiguous queries                         -> nns_index_search
  [N > 1] all-reduce(min) of the keys over RCCL
  unpack keys -> int32 indices                         -> nns_keys_unpack
(the reference times alloc + H2D + D2H too, main.cu:73-75; the PCIe-inclusive figure
of the whole-call drop-in is reported in DESIGN.md, never here).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
    "c2": (4096, 65536, 3, "f32"),          # per-pair exact kernel, no MFMA
    "c5": (131072, 2097152, 256, "bf16"),   # bf16 points, fp32 accumulate
    "c3s": (8192, 131072, 128, "f32"),      # quick look, not a reported config
    # EXTRA, not a BASELINE configuration and never the default: C3's fp32 points through the opt-in bf16
    # MFMA filter (NNS_FILTER_BF16) + exact fp32 re-rank — same result bits, reduced-precision GEMM
    "c3x": (65536, 1048576, 128, "f32"),
}


def cpu_baseline(orc, q_h, r_dev, idx_gpu, target_s=12.0):
    """V0 on the host, single thread, on a bounded sample: the first `s` queries of the
    workload against ALL of this GPU's refs.  Uses the reference's own V0 binary when it
    was built (oracle/_ref), else our restatement.  Also cross-checks the GPU indices."""
    r_h = r_dev.cpu().numpy()
    n = r_h.shape[0]
    use_ref = orc.have_reference()

    def run(s):
        t0 = time.perf_counter()
        if use_ref:
            idx = orc.v0_reference(q_h[:s], r_h)
        else:
            idx, _ = orc.v0_search(q_h[:s], r_h)
        return time.perf_counter() - t0, idx

    t1, _ = run(1)
    s = int(max(1, min(q_h.shape[0], target_s / max(t1, 1e-6))))
    t, idx = run(s)
    ok = bool(np.array_equal(idx, idx_gpu[:s]))
    out = {"value": s * n / t, "unit": "pairs/s", "cores": 1,
           "kind": "reference" if use_ref else "port",
           "sample": f"first {s} queries x all {n} refs of the workload, {t:.1f} s, V0 single thread"
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
    # low-dimensional workloads: the reference's other working algorithm family (V10's CPU k-d
    # tree, core.cu:1060-1163, which itself bails out above 16 dimensions) as a second comparator;
    # oracle/kdtree.c is exact with V0's semantics (equivalent pairs/s = m * n / t, build included)
    if q_h.shape[1] <= 16 and np.isfinite(r_h).all() and np.isfinite(q_h).all():
        t0 = time.perf_counter()
        idx3, _ = orc.kdtree_search(q_h, r_h, threads=cores)
        t3 = time.perf_counter() - t0
        out["kdtree"] = {"value": q_h.shape[0] * n / t3, "unit": "pairs/s (equivalent: m*n / wall time incl. build)",
                         "cores": cores, "kind": "port", "queries": int(q_h.shape[0]), "seconds": t3,
                         "matches_gpu_indices": bool(np.array_equal(idx3, idx_gpu[:q_h.shape[0]]))}
    return out


def load_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed PMC run
    (profiles/traffic.json, produced by tools/pmc_traffic.py from rocprofv3 --pmc
    passes, FETCH_SIZE x2-corrected per MI355X_MICROARCH.md); None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(kernel_key)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (NOT a measurement):
    # every rank on cuda:0, keys exchanged through gloo on the host instead of RCCL
    ap.add_argument("--rehearse-one-gpu", action="store_true")
    ap.add_argument("--verify", action="store_true", help="rank 0 checks the merged indices against the unsharded search")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run: one rank per GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.rehearse_one_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    pkg = graft.load_package()
    m, n_local, k, dtype = WORKLOADS[args.workload]
    path = "auto"
    n_total = n_local * world
    seed = 1000

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
    mixed = args.workload == "c3x"
    ix = pkg.Index(r, index_base=beg, path=path, profile=True, filter_bf16=mixed)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ix.refresh()                          # K2 on refs
        ix.search_keys(q, keys)               # K2 queries, K3 filter, K5, re-rank
        if dist is not None:
            pkg.allreduce_min_keys(keys)      # MINLOC-style exchange: one min all-reduce
        return pkg.keys_unpack(keys)

    for _ in range(args.warmup):
        step()
    stage = {"filter_ms": 0.0, "exact_ms": 0.0, "prep_refs_ms": 0.0, "prep_queries_ms": 0.0,
             "finalize_ms": 0.0, "rerank_ms": 0.0}
    amb = 0
    ix.stats()                                # drop the warm-up steps' event sets
    barrier()
    t0 = time.perf_counter()
    done = 0
    for i in range(args.steps):
        idx = step()
        # HIP events of every step's kernels are recorded on the launch stream inside the
        # library; read (= one device sync) every 32 steps at most: its ring of event sets
        if (i + 1) % 32 == 0 and i + 1 < args.steps:
            st = ix.stats()
            for f in stage:
                stage[f] += st[f] * 32
            done += 32
    barrier()
    elapsed = time.perf_counter() - t0
    st = ix.stats()                           # averages over the steps not read yet
    for f in stage:
        stage[f] += st[f] * (args.steps - done)
    amb = st["ambiguous"]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for f in stage:
        stage[f] /= max(1, args.steps)

    ms_per_step = elapsed / args.steps * 1e3
    value = m * float(n_total) / (elapsed / args.steps)

    if rank == 0:
        mfma_path = st["path"] == 2
        if mfma_path:
            kern_ms = stage["filter_ms"]
            flops = 2.0 * m * n_local * k       # algorithmic: 2*k flop per pair
            achieved = flops / (kern_ms * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS if (dtype == "bf16" or mixed) else PEAK_F32_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": f"filter_kernel<{'OpBF16' if (dtype == 'bf16' or mixed) else 'OpF32'}>",
                    "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "kernel_ms": kern_ms, "flop_per_pair": 2 * k,
                    "traffic": load_traffic(f"filter_f32_{args.workload}")}
        else:
            kern_ms = stage["exact_ms"]
            flops = 3.0 * k * m * n_local       # sub, mul, add per dim
            achieved = flops / (kern_ms * 1e-3) / 1e12
            roof = {"bound": "valu", "kernel": "exact_lane_query_kernel", "achieved": achieved,
                    "peak": PEAK_F32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_VALU_TFLOPS,
                    "kernel_ms": kern_ms, "flop_per_pair": 3 * k,
                    "traffic": load_traffic(f"exact_{args.workload}")}
        out = {
            "metric": "query-point-pairs/s", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 points, f32 accumulate" if dtype == "bf16"
            else ("f32 points, bf16 filter operands, f32 accumulate, exact f32 re-rank (opt-in extra)" if mixed else "f32"),
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {m} queries x {n_total} refs x {k}-D {dtype}"
                                   f" ({n_local} refs per GPU x {world} GPU)",
                       "m": m, "n": n_total, "k": k, "refs_per_gpu": n_local,
                       "path": "mfma-filter+exact-rerank" if mfma_path else "exact-valu",
                       "ambiguous_queries": amb, "stage_ms": {f: round(v, 4) for f, v in stage.items()}},
            "roofline": roof,
        }
        if args.rehearse_one_gpu:
            out["rehearsal"] = "all ranks on one GPU, gloo exchange: NOT a measurement"
        if args.verify:
            # the merged answer must equal one unsharded search over all n_total refs
            r_all = torch.empty((n_total, k), dtype=torch.float32, device=dev)
            pkg.fill_uniform(r_all, seed, m * k)
            if dtype == "bf16":
                r_all = r_all.to(torch.bfloat16)
            ix_all = pkg.Index(r_all)
            want = ix_all.search(q)
            torch.cuda.synchronize()
            out["verified_vs_unsharded"] = bool(torch.equal(want, idx))
            ix_all.close()
            del r_all
        if world == 1 and not args.no_cpu_baseline:
            orc = graft.load_oracle()          # cpu_baseline leg only
            out["cpu_baseline"] = cpu_baseline(orc, q[:4096].float().cpu().numpy(), r.float(), idx.cpu().numpy())
        print(json.dumps(out), flush=True)
    ix.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
