#!/usr/bin/env python3
"""Tuning probe (not part of the product): time each filter variant on a shape."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=65536)
    ap.add_argument("--n", type=int, default=1048576)
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--variants", default="0", help="labels only (ablations are compile-time: -DNNS_FILTER_ABLATE)")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    pkg = graft.load_package()
    q = torch.empty((a.m, a.k), dtype=torch.float32, device="cuda")
    r = torch.empty((a.n, a.k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1000, 0)
    pkg.fill_uniform(r, 1000, a.m * a.k)
    flops = 2.0 * a.m * a.n * a.k
    ref_keys = None
    for v in [int(x) for x in a.variants.split(",")]:
        ix = pkg.Index(r, path="mfma", profile=True)
        best = None
        for _ in range(a.reps):
            keys = ix.search_keys(q)
            torch.cuda.synchronize()
            st = ix.stats()
            if best is None or st["filter_ms"] < best["filter_ms"]:
                best = st
        same = True if ref_keys is None else bool(torch.equal(ref_keys, keys))
        if ref_keys is None:
            ref_keys = keys.clone()
        tf = flops / (best["filter_ms"] * 1e-3) / 1e12
        print(f"run {v}: filter {best['filter_ms']:.2f} ms = {tf:.1f} TF ({tf / 157.3 * 100:.1f}% of 157.3) "
              f"| total {best['total_ms']:.2f} ms prepR {best['prep_refs_ms']:.2f} prepQ {best['prep_queries_ms']:.2f} "
              f"final {best['finalize_ms']:.2f} rerank {best['rerank_ms']:.2f} amb {best['ambiguous']} "
              f"splits {best['splits']} keys_equal_v0 {same}", flush=True)
        ix.close()


if __name__ == "__main__":
    main()
