#!/usr/bin/env python3
"""Tuning probe: per-stage HIP-event times for one shape: probe_stages.py k m n"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
k, m, n = (int(x) for x in sys.argv[1:4])
q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
ix = pkg.Index(r, profile=True)
for _ in range(3):
    ix.search_keys(q); torch.cuda.synchronize()
    print({a: round(b, 4) if isinstance(b, float) else b for a, b in ix.stats().items()})
