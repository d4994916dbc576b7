#!/usr/bin/env python3
"""Tuning probe: device-resident search time for the reference driver's shapes (main.cu:38-51)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
SHAPES = [(3, 1, 1024), (16, 1, 1024), (3, 1, 65536), (16, 1, 65536), (3, 1024, 1024), (16, 1024, 1024),
          (3, 1024, 65536), (16, 1024, 65536), (3, 1024, 1048576), (16, 1024, 1048576),
          (3, 1, 1048576), (16, 1, 1048576), (128, 1, 1048576), (128, 64, 1048576), (128, 1024, 1048576)]
for (k, m, n) in SHAPES:
    q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
    ix = pkg.Index(r)
    keys = torch.empty(m, dtype=torch.int64, device="cuda")
    for _ in range(3): ix.search_keys(q, keys)
    torch.cuda.synchronize(); reps = 50; t0 = time.perf_counter()
    for _ in range(reps): ix.search_keys(q, keys)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    hbm = (n * k * 4 + m * k * 4) / dt / 1e9
    print(f"k={k:3d} m={m:5d} n={n:8d}: {dt * 1e6:9.1f} us  {m * n / dt:.3e} pairs/s  input-stream {hbm:7.1f} GB/s  path={ix.stats()['path']}", flush=True)
    ix.close()
