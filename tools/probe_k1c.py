#!/usr/bin/env python3
"""Tuning probe: the HBM-bound few-query shapes (K1c / K1b): HIP-event time per search (library events) and
back-to-back wall time; GB/s = n * k * 4 / t."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
SHAPES = [(3, 1, 65536), (16, 1, 65536), (3, 1, 1048576), (16, 1, 1048576), (16, 4, 1048576), (32, 1, 1048576),
          (128, 1, 1048576), (16, 1, 4194304), (3, 1, 16777216)]
tag = os.environ.get("NNS_LIB_PATH", "product").split("libnns_")[-1]
for (k, m, n) in SHAPES:
    q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
    keys = torch.empty(m, dtype=torch.int64, device="cuda")
    ixp = pkg.Index(r, profile=True)
    for _ in range(5): ixp.search_keys(q, keys)
    ixp.stats()
    for _ in range(30): ixp.search_keys(q, keys)
    ev_us = ixp.stats()["exact_ms"] * 1e3
    ixp.close()
    ix = pkg.Index(r)
    for _ in range(5): ix.search_keys(q, keys)
    torch.cuda.synchronize(); reps = 300; t0 = time.perf_counter()
    for _ in range(reps): ix.search_keys(q, keys)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    ix.close()
    b = n * k * 4
    print(f"{tag:14s} k={k:3d} m={m} n={n:9d}: events {ev_us:8.1f} us ({b / ev_us / 1e6:7.2f} TB/s)   back-to-back {dt * 1e6:8.1f} us ({b / dt / 1e12:6.2f} TB/s)", flush=True)
