#!/usr/bin/env python3
"""Tuning probe: whole-call time (host arrays in, indices out) of mid-size problems (uploads of 12 MiB .. 512 MiB)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
lib = pkg.lib
pkg.warmup()
rng = np.random.default_rng(0)
for (k, m, n) in [(3, 1024, 1048576), (16, 1024, 1048576), (3, 4096, 4194304), (16, 4096, 1048576), (128, 1024, 262144), (128, 4096, 524288), (128, 16384, 1048576)]:
    q = rng.random((m, k), dtype=np.float32); r = rng.random((n, k), dtype=np.float32)
    idx = np.empty(m, np.int32)
    for _ in range(2):
        lib.nns_search_f32_ex(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data, None, 1, 0, 0)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        lib.nns_search_f32_ex(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data, None, 1, 0, 0)
        ts.append(time.perf_counter() - t0)
    print(f"k={k:3d} m={m:5d} n={n:7d} refs {r.nbytes / 2**20:6.0f} MiB: whole call median {sorted(ts)[3] * 1e3:8.3f} ms  min {min(ts) * 1e3:8.3f} ms", flush=True)
