#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box (not a pytest: run it for as long as you like).

    python tools/fuzz_parity.py --seconds 240 --seed 1

Random shapes (m, n, k), data families (uniform, clustered, duplicates, huge / tiny scale, offset
clouds, planted NaN / INF), dtypes (fp32, bf16), paths (auto / mfma / exact) and shard counts; every
result is compared index-for-index and distance-bit-for-bit with the V0 oracle.  Prints one line
per failure and a summary; exit status 1 if anything differed.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def make_cloud(rng, m, n, k, family):
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    if family == "clustered":
        c = rng.random((8, k), dtype=np.float32) * 4
        r = (c[rng.integers(0, 8, n)] + rng.normal(0, 0.05, (n, k))).astype(np.float32)
        q = (c[rng.integers(0, 8, m)] + rng.normal(0, 0.05, (m, k))).astype(np.float32)
    elif family == "duplicates":
        base = r[: max(1, n // 7)]
        r = base[rng.integers(0, base.shape[0], n)].copy()
        q[: m // 2] = r[rng.integers(0, n, m // 2)]
    elif family == "huge":
        q *= np.float32(3e5)
        r *= np.float32(3e5)
    elif family == "tiny":
        q *= np.float32(1e-12)
        r *= np.float32(1e-12)
    elif family == "offset":
        q += np.float32(1000.0)
        r += np.float32(1000.0)
    elif family == "specials" and n > 4:
        r[rng.integers(0, n), rng.integers(0, k)] = np.nan
        r[rng.integers(0, n), rng.integers(0, k)] = np.inf
        if m > 2:
            q[rng.integers(0, m), rng.integers(0, k)] = np.inf
    return q, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-pairs", type=float, default=3e8)
    ap.add_argument("--lowdim", action="store_true", help="only shapes of K1f's domain: k <= 3, mostly >= 2^27 pairs, fp32")
    a = ap.parse_args()
    pkg, orc = graft.load_package(), graft.load_oracle()
    rng = np.random.default_rng(a.seed)
    families = ["uniform", "clustered", "duplicates", "huge", "tiny", "offset", "specials"]
    t0 = time.time()
    cases = fails = 0
    last = t0
    while time.time() - t0 < a.seconds:
        k = int(rng.choice([1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 64, 65, 100, 128, 129, 200, 256, 257, 300, 384, 385, 400, 512, 513, 600, 640, 641, 700,
                            768, 769, 777, 1024, 1025]))
        m = int(rng.choice([1, 2, 3, 4, 5, 31, 63, 64, 65, 255, 512, 513, 1000, 2049]))
        n = int(rng.choice([1, 5, 31, 64, 65, 257, 1000, 4097, 20001, 70000, 150000]))
        while float(m) * n * k > a.max_pairs * 64:
            n = max(1, n // 2)
        # now and then a whole call big enough for the chunked, overlapped upload (>= 8 MiB of refs, search >= 0.5 ms)
        if rng.integers(0, 40) == 0:
            k, m, n = [(3, 4096, 700000), (16, 4096, 600000), (128, 4096, 70000)][int(rng.integers(0, 3))]
            n += int(rng.integers(0, 1000))
        if a.lowdim:
            k = int(rng.choice([1, 2, 3, 3]))
            m = int(rng.choice([513, 1000, 2049, 4096, 5000]))
            n = int(rng.choice([30000, 70000, 150000, 400000])) + int(rng.integers(0, 40))
        fam = str(rng.choice(families))
        bf16 = bool(rng.integers(0, 3) == 0) and fam not in ("huge",) and not a.lowdim
        # ("mfma_perref": the long-stream record form of the filter forced at any size, NNS_RECORDS_PER_REF)
        path = str(rng.choice(["auto", "auto", "mfma", "mfma_perref", "exact"]))
        if path.startswith("mfma") and k > (1024 if bf16 else 256):
            path = "auto"
        shards = int(rng.choice([1, 1, 2, 3]))
        # opt-in bf16 filter on fp32 points (NNS_FILTER_BF16): same answers required
        mixed = (not bf16) and k <= 1024 and path != "exact" and bool(rng.integers(0, 2))
        q, r = make_cloud(rng, m, n, k, fam)
        with np.errstate(all="ignore"):
            if bf16:
                qw, rw = orc.round_bf16(q), orc.round_bf16(r)
                want_idx, want_dist = orc.v0_search(qw, rw, threads=16)
                idx, dist = pkg.search_bf16(pkg.to_bf16_bits(q), pkg.to_bf16_bits(r), return_distances=True,
                                            shards=shards, path=path)
            else:
                want_idx, want_dist = orc.v0_search(q, r, threads=16)
                idx, dist = pkg.search(q, r, return_distances=True, shards=shards, path=path, filter_bf16=mixed)
        cases += 1
        ok = np.array_equal(idx, want_idx) and np.array_equal(dist.view(np.uint32), want_dist.view(np.uint32))
        if not ok:
            fails += 1
            bad = np.nonzero(idx != want_idx)[0]
            print(f"FAIL m={m} n={n} k={k} family={fam} bf16={bf16} mixed={mixed} path={path} shards={shards}: "
                  f"{bad.size} index mismatches (first {bad[:4]}), "
                  f"{int((dist.view(np.uint32) != want_dist.view(np.uint32)).sum())} distance mismatches", flush=True)
        if time.time() - last > 30:
            last = time.time()
            print(f"... {cases} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz_parity: {cases} cases, {fails} failures in {time.time() - t0:.0f} s (seed {a.seed})")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
