#!/usr/bin/env python3
"""Tuning probe: exact K1a vs MFMA filter around the small-problem crossover (k = 8 / 16)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
for (k, m, n) in [(16, 1024, 16384), (16, 1024, 32768), (16, 1024, 65536), (16, 1024, 131072), (16, 4096, 8192), (16, 4096, 16384),
                  (16, 2048, 16384), (8, 4096, 8192), (8, 4096, 16384), (8, 1024, 65536), (16, 16384, 4096), (16, 65536, 1024)]:
    q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
    res = []
    for path in ("exact", "mfma", "auto"):
        ix = pkg.Index(r, path=path); keys = torch.empty(m, dtype=torch.int64, device="cuda")
        for _ in range(5): ix.search_keys(q, keys)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): ix.search_keys(q, keys)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 100 * 1e6); ix.close()
    print(f"k={k:2d} m={m:5d} n={n:6d} pairs=2^{(m*n).bit_length()-1}: exact {res[0]:7.1f} us  mfma {res[1]:7.1f} us  auto {res[2]:7.1f} us", flush=True)
