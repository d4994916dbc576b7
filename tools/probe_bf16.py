#!/usr/bin/env python3
"""Tuning probe: bf16 filter time on one shape: probe_bf16.py m n k"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
m, n, k = (int(x) for x in sys.argv[1:4])
q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
q = q.to(torch.bfloat16); r = r.to(torch.bfloat16)
ix = pkg.Index(r, profile=True)
for _ in range(3):
    ix.search_keys(q); torch.cuda.synchronize(); st = ix.stats()
tf = 2.0 * m * n * k / (st["filter_ms"] * 1e-3) / 1e12
print(f"bf16 {m}x{n}x{k}: k_tile {st['k_tile']} filter {st['filter_ms']:.2f} ms = {tf:.0f} TFLOP/s ({tf / 2500 * 100:.1f} % of 2.5 PF), total {st['total_ms']:.2f} ms, ambiguous {st['ambiguous']}")
