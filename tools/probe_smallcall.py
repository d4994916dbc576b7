#!/usr/bin/env python3
"""Tuning probe: whole-call time (host arrays in, indices out) of small problems through the pinned-scratch path
and through the plain path (NNS_PROFILE keeps a call off the scratch path), by input size."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
host = sys.modules[pkg.__name__ + ".host"]
lib = pkg.lib
pkg.warmup()
rng = np.random.default_rng(0)
for (k, m, n) in [(3, 1, 1024), (16, 1024, 1024), (3, 1024, 65536), (16, 64, 8192), (16, 64, 16384), (16, 64, 24576), (16, 64, 32000), (128, 256, 3500)]:
    q = rng.random((m, k), dtype=np.float32); r = rng.random((n, k), dtype=np.float32)
    idx = np.empty(m, np.int32)
    res = []
    for flags in (0, host.NNS_PROFILE):
        for _ in range(3):
            lib.nns_search_f32_ex(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data, None, 1, flags, 0)
        t0 = time.perf_counter()
        for _ in range(50):
            lib.nns_search_f32_ex(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data, None, 1, flags, 0)
        res.append((time.perf_counter() - t0) / 50 * 1e6)
    print(f"k={k:3d} m={m:5d} n={n:6d} inputs {(q.nbytes + r.nbytes) / 1024:7.0f} KiB: scratch path {res[0]:7.1f} us   plain path {res[1]:7.1f} us", flush=True)
