#!/usr/bin/env python3
"""Tuning probe: timeline of one K1a launch at C2 (a library built with EXACT=-DNNS_K1A_STAMPS prints it)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
m, n, k = [int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (4096, 65536, 3))]
q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
pkg.fill_uniform(q, 1000, 0); pkg.fill_uniform(r, 1000, m * k)
ix = pkg.Index(r)
keys = torch.empty(m, dtype=torch.int64, device="cuda")
os.environ.pop("NNS_K1A_STAMPS", None)
for _ in range(20):
    ix.search_keys(q, keys)
torch.cuda.synchronize()
os.environ["NNS_K1A_STAMPS"] = "1"
for _ in range(2):
    ix.search_keys(q, keys)
torch.cuda.synchronize()
