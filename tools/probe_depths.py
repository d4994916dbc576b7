#!/usr/bin/env python3
"""Filter throughput at every tile depth the product instantiates (the numbers DESIGN.md section 4 quotes):
65536 x 1048576 at k = 16 / 32 / 64 / 128 / 256 (fp32 tiles), bf16 points at k = 128 / 256 / 512, fp32 points
through the bf16-operand tile at k = 512.  HIP-event time of the filter kernel, algorithmic 2*k flop per pair."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
m, n = 65536, 1048576
CASES = [(16, "f32"), (32, "f32"), (64, "f32"), (128, "f32"), (256, "f32"), (512, "f32"), (1024, "f32"),
         (128, "bf16"), (256, "bf16"), (512, "bf16"), (1024, "bf16")]
if "--deep" in sys.argv:
    CASES = [(1024, "f32"), (1024, "bf16"), (600, "bf16"), (640, "bf16"), (768, "bf16"), (700, "bf16"), (600, "f32")]
if "--k16" in sys.argv:     # the shallow fp32 tiles only
    CASES = [(16, "f32"), (32, "f32")]
if "--nw4" in sys.argv:     # the one-wave-per-SIMD tiles only
    CASES = [(512, "bf16"), (1024, "bf16")]
if "--k512" in sys.argv:
    CASES = [(512, "bf16"), (512, "f32"), (300, "bf16"), (384, "bf16"), (300, "f32")]
for k, dt in CASES:
    q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1000, 0); pkg.fill_uniform(r, 1000, m * k)
    if dt == "bf16":
        q = q.to(torch.bfloat16); r = r.to(torch.bfloat16); torch.cuda.empty_cache()
    ix = pkg.Index(r, profile=True)
    keys = torch.empty(m, dtype=torch.int64, device="cuda")
    for _ in range(2): ix.search_keys(q, keys)
    ix.stats()
    for _ in range(5): ix.search_keys(q, keys)
    st = ix.stats()
    tf = 2.0 * m * n * k / (st["filter_ms"] * 1e-3) / 1e12
    peak = 157.3 if (dt == "f32" and k <= 256) else 2500.0
    print(f"{dt:4s} points k={k:3d}: tile kt={st['k_tile']:3d}  filter {st['filter_ms']:8.2f} ms = {tf:7.1f} TFLOP/s algorithmic = "
          f"{tf / peak * 100:5.1f} % of {peak:.1f}  (total {st['total_ms']:.2f} ms, ambiguous {st['ambiguous']}, near-ties {st['multi_candidate']})", flush=True)
    ix.close()
    if k > 512 and "--deep" in sys.argv:   # the same through the exact VALU kernels, on 1/16 of the queries
        ms = m // 16
        ixe = pkg.Index(r, path="exact", profile=True)
        ke = torch.empty(ms, dtype=torch.int64, device="cuda")
        ixe.search_keys(q[:ms], ke); ixe.stats(); ixe.search_keys(q[:ms], ke)
        ste = ixe.stats()
        print(f"     exact VALU scan of {ms} queries: {ste['exact_ms']:.1f} ms -> {ste['exact_ms'] * 16:.0f} ms for all {m}; "
              f"keys equal: {bool(torch.equal(ke, keys[:ms]))}", flush=True)
        ixe.close()
    del q, r, keys; torch.cuda.empty_cache(); pkg.trim()
