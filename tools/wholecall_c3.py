#!/usr/bin/env python3
"""PCIe-inclusive whole-call timing of the drop-in entry point at C3 (never the bench value)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
m, n, k = 65536, 1048576, 128
q = orc.rng_uniform(m * k, 1000, 0).reshape(m, k); r = orc.rng_uniform(n * k, 1000, m * k).reshape(n, k)
ref = None
for rep in range(4):
    t0 = time.perf_counter(); idx = pkg.cudaCall(k, m, n, q, r); dt = time.perf_counter() - t0
    print(f"whole call nns_search_f32 (malloc + H2D of {(q.nbytes + r.nbytes) / 2**20:.0f} MiB pageable + search + D2H + free): {dt * 1e3:.1f} ms -> {m * n / dt:.3e} pairs/s")
    if ref is None: ref = idx
    assert np.array_equal(ref, idx)
idx2 = pkg.search(q, r, shards=2)   # the non-pipelined path
print("pipelined == sharded path:", bool(np.array_equal(idx, idx2)))
