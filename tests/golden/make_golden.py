#!/usr/bin/env python3
"""Generate the committed golden vectors for the nearest-neighbour hot path.

Run in the BUILD CONTAINER only (needs /root/reference to have been compiled into
oracle/_ref/libv0ref.so by oracle/build_ref.sh).  Every expected output below is
produced by the reference's OWN V0 (core.cu:11-54), never by our code:

  golden_recipe.npz   the reference driver's ten samples (main.cu:38-51) drawn
                      with its data recipe (srand(1000), glibc rand(), queries then
                      refs: main.cu:10-13, 27-34, 54, 64).  Inputs are stored for the
                      6 small samples; the 1024x65536 and 1024x1048576 samples store the
                      expected indices plus an FNV-1a digest of the inputs (regenerated
                      from the glibc stream at test time).
  golden_cases.npz    adversarial cases (exact ties from duplicated refs, NaN / INF
                      rows, near-ties, clusters far from the origin, ragged sizes,
                      m = 1, n = 1, 128-D clouds) with inputs and expected indices.

The reference cannot travel to the GPU box; these fixtures (data only) do.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as graft  # noqa: E402

SAMPLES = [(3, 1, 1024), (16, 1, 1024), (3, 1, 65536), (16, 1, 65536),
           (3, 1024, 1024), (16, 1024, 1024), (3, 1024, 65536), (16, 1024, 65536),
           (3, 1024, 1048576), (16, 1024, 1048576)]      # all ten of main.cu:38-51
STORE_INPUT_LIMIT = 1 << 20   # floats


def adversarial_cases(rng):
    cases = {}

    def add(name, q, r):
        cases[name] = (np.ascontiguousarray(q, np.float32), np.ascontiguousarray(r, np.float32))

    # exact ties: every ref appears twice -> lowest index must win
    r = rng.random((300, 3), dtype=np.float32)
    add("ties_dup_k3", rng.random((64, 3), dtype=np.float32), np.concatenate([r, r]))
    r = rng.random((257, 128), dtype=np.float32)
    add("ties_dup_k128", rng.random((70, 128), dtype=np.float32), np.concatenate([r, r, r]))
    # all refs identical: index 0 everywhere
    add("all_same", rng.random((33, 16), dtype=np.float32), np.tile(rng.random((1, 16), dtype=np.float32), (100, 1)))
    # queries ARE refs (distance 0, and duplicates of them later in the list)
    r = rng.random((500, 128), dtype=np.float32)
    r[400:450] = r[10:60]
    add("self_match_k128", r[:200].copy(), r)
    # NaN / INF handling: V0 never selects a NaN or INF distance; all-bad row -> 0
    r = rng.random((130, 4), dtype=np.float32)
    r[0, 1] = np.nan
    r[5, 0] = np.inf
    r[17, 2] = -np.inf
    q = rng.random((20, 4), dtype=np.float32)
    q[3, 0] = np.nan            # every distance NaN -> index 0
    q[4, 1] = np.inf            # every distance INF/NaN -> index 0
    add("nan_inf_k4", q, r)
    r = rng.random((200, 128), dtype=np.float32)
    r[7, 100] = np.nan
    r[150, 5] = np.inf
    q = rng.random((40, 128), dtype=np.float32)
    q[9, 64] = np.nan
    add("nan_inf_k128", q, r)
    # near ties: refs that differ from each other by 1 ulp in one coordinate
    base = rng.random((1, 128), dtype=np.float32)
    r = np.tile(base, (96, 1))
    for j in range(96):
        r[j, j % 128] = np.nextafter(r[j, j % 128], np.float32(2.0 if j % 2 else -1.0))
    add("near_ties_k128", base + rng.normal(0, 1e-3, (50, 128)).astype(np.float32), r)
    # clusters far from the origin (centring by the mean matters for the GEMM form)
    c = rng.random((8, 64), dtype=np.float32) * 1000.0 + 5000.0
    r = (c[rng.integers(0, 8, 1500)] + rng.normal(0, 0.05, (1500, 64))).astype(np.float32)
    q = (c[rng.integers(0, 8, 100)] + rng.normal(0, 0.05, (100, 64))).astype(np.float32)
    add("far_clusters_k64", q, r)
    # ragged shapes
    add("m1_k128", rng.random((1, 128), dtype=np.float32), rng.random((1000, 128), dtype=np.float32))
    add("n1_k128", rng.random((17, 128), dtype=np.float32), rng.random((1, 128), dtype=np.float32))
    add("n1_k3", rng.random((17, 3), dtype=np.float32), rng.random((1, 3), dtype=np.float32))
    add("k1", rng.random((100, 1), dtype=np.float32), rng.random((999, 1), dtype=np.float32))
    add("k5_odd", rng.random((77, 5), dtype=np.float32), rng.random((1234, 5), dtype=np.float32))
    add("k100_pad", rng.random((65, 100), dtype=np.float32), rng.random((1030, 100), dtype=np.float32))
    add("k32", rng.random((513, 32), dtype=np.float32), rng.random((2049, 32), dtype=np.float32))
    add("k200_big", rng.random((9, 200), dtype=np.float32), rng.random((700, 200), dtype=np.float32))
    add("uniform_k128", rng.random((600, 128), dtype=np.float32), rng.random((3000, 128), dtype=np.float32))
    # negative / mixed-sign, large dynamic range
    add("mixed_sign_k16", rng.normal(0, 100, (90, 16)).astype(np.float32),
        rng.normal(0, 100, (2000, 16)).astype(np.float32))
    return cases


def main():
    orc = graft.load_oracle()
    if not orc.have_reference():
        raise SystemExit("oracle/_ref/libv0ref.so missing: run `make -C oracle` where /root/reference exists")
    out = {}
    for i, (k, m, n, q, r) in enumerate(orc.ref_recipe(SAMPLES, seed=1000)):
        idx = orc.v0_reference(q, r)
        out[f"s{i}_shape"] = np.array([k, m, n], np.int64)
        out[f"s{i}_idx"] = idx
        out[f"s{i}_input_fnv"] = np.array([orc.fnv1a64(q), orc.fnv1a64(r)], np.uint64)
        if q.size + r.size <= STORE_INPUT_LIMIT:
            out[f"s{i}_q"] = q
            out[f"s{i}_r"] = r
        print(f"recipe sample {i}: k={k} m={m} n={n} first idx {idx[:6].tolist()} fnv {orc.fnv1a64(idx):016x}")
    np.savez_compressed(os.path.join(HERE, "golden_recipe.npz"), **out)

    rng = np.random.default_rng(20261004)
    out = {}
    for name, (q, r) in adversarial_cases(rng).items():
        with np.errstate(all="ignore"):
            idx = orc.v0_reference(q, r)
        out[f"{name}__q"] = q
        out[f"{name}__r"] = r
        out[f"{name}__idx"] = idx
        print(f"case {name}: q{q.shape} r{r.shape} first idx {idx[:6].tolist()}")
    np.savez_compressed(os.path.join(HERE, "golden_cases.npz"), **out)


if __name__ == "__main__":
    main()
