#!/usr/bin/env python3
"""Tuning probe: K1f (VALU filter + re-rank) against K1a around its size threshold.  Run once per library
(NNS_LIB_PATH: the product, and a build with EXACT=-DNNS_K1F_OFF or -DNNS_K1F_MIN_PAIRS=1)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
k = 3
for (m, n) in [(256, 65536), (512, 65536), (1024, 16384), (1024, 65536), (2048, 65536), (4096, 16384), (4096, 65536), (1024, 262144),
               (2048, 262144), (8192, 65536), (16384, 16384), (1024, 1048576), (65536, 65536)]:
    q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
    ix = pkg.Index(r)
    keys = torch.empty(m, dtype=torch.int64, device="cuda")
    for _ in range(5): ix.search_keys(q, keys)
    torch.cuda.synchronize(); reps = 200; t0 = time.perf_counter()
    for _ in range(reps): ix.search_keys(q, keys)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"k={k} m={m:6d} n={n:8d} pairs=2^{(m * n).bit_length() - 1}: {dt * 1e6:9.1f} us", flush=True)
    ix.close()
