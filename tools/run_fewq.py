#!/usr/bin/env python3
"""One few-query (HBM-bound) search shape, R times back to back — the thing to put behind rocprofv3.
usage: run_fewq.py k m n [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
k, m, n = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
ix = pkg.Index(r)
keys = torch.empty(m, dtype=torch.int64, device="cuda")
for _ in range(reps): ix.search_keys(q, keys)
torch.cuda.synchronize()
ix.close()
