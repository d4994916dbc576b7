#!/usr/bin/env python3
"""Tuning probe: back-to-back C2 searches (no host sync between launches)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
m, n, k = 4096, 65536, 3
q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
pkg.fill_uniform(q, 1000, 0); pkg.fill_uniform(r, 1000, m * k)
ix = pkg.Index(r)
keys = torch.empty(m, dtype=torch.int64, device="cuda")
for reps in (1, 10, 100, 1000, 1000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        ix.search_keys(q, keys)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{reps:5d} back-to-back searches: {dt / reps * 1e6:8.1f} us each -> {m * n / (dt / reps):.3e} pairs/s", flush=True)
