#!/usr/bin/env python3
"""Tuning probe: per-stage HIP-event times of the shapes the short-stream work targets
(1024 x 1 M x 16 / x 128: the reference driver's samples; 16384 x 65536 x 128; C3 for no-regression)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
SHAPES = [(16, 1024, 1048576), (128, 1024, 1048576), (128, 16384, 65536), (3, 1024, 1048576), (64, 1024, 1048576),
          (256, 1024, 1048576), (128, 4096, 1048576), (16, 65536, 1048576)]
if "--c3" in sys.argv:
    SHAPES.append((128, 65536, 1048576))
for (k, m, n) in SHAPES:
    q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
    ix = pkg.Index(r, profile=True)
    keys = torch.empty(m, dtype=torch.int64, device="cuda")
    for _ in range(3): ix.search_keys(q, keys)
    ix.stats()
    for _ in range(20): ix.search_keys(q, keys)
    st = ix.stats()
    fl = 2.0 * m * n * k / (st["filter_ms"] * 1e-3) / 1e12 if st["filter_ms"] > 0 else 0.0
    print(f"k={k:3d} m={m:5d} n={n:8d}: total {st['total_ms']*1e3:9.1f} us  qprep {st['prep_queries_ms']*1e3:7.1f}  filter {st['filter_ms']*1e3:9.1f}"
          f" ({fl:6.1f} TF algorithmic, kt={st['k_tile']}, splits={st['splits']})  final {st['finalize_ms']*1e3:7.1f}  rerank {st['rerank_ms']*1e3:6.1f}"
          f"  exact {st['exact_ms']*1e3:8.1f}  amb={st['ambiguous']} multi={st['multi_candidate']}", flush=True)
    ix.close()
