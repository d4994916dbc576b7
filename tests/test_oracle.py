"""CPU tests: the oracle (our restatement of V0) against the golden vectors and,
where present, against the reference's own V0 binary.  No GPU."""
import numpy as np
import pytest

SAMPLES = [(3, 1, 1024), (16, 1, 1024), (3, 1, 65536), (16, 1, 65536),
           (3, 1024, 1024), (16, 1024, 1024), (3, 1024, 65536), (16, 1024, 65536),
           (3, 1024, 1048576), (16, 1024, 1048576)]

# first indices of the reference's V0 on its own srand(1000) stream
SURVEY_FIRST = {0: [537], 1: [733], 2: [59136], 3: [41906],
                4: [179, 754, 847, 432, 640, 8], 5: [134, 274, 9, 27, 690, 797],
                6: [12922, 16487, 38271, 16165, 14710, 26533],
                7: [12389, 51866, 40272, 61522, 33809, 52264]}


def _cases(golden_dir):
    z = np.load(f"{golden_dir}/golden_cases.npz")
    names = sorted({k.split("__")[0] for k in z.files})
    return z, names


def test_oracle_matches_golden_recipe_inputs_stored(orc, golden_dir):
    """Stored-input samples of the reference driver's table: oracle == reference V0."""
    z = np.load(f"{golden_dir}/golden_recipe.npz")
    checked = 0
    for i in range(8):
        if f"s{i}_q" not in z.files:
            continue
        q, r, want = z[f"s{i}_q"], z[f"s{i}_r"], z[f"s{i}_idx"]
        assert tuple(z[f"s{i}_shape"]) == SAMPLES[i]
        idx, _ = orc.v0_search(q, r)
        assert np.array_equal(idx, want)
        assert idx[:len(SURVEY_FIRST[i])].tolist() == SURVEY_FIRST[i]
        checked += 1
    assert checked >= 6


def test_oracle_matches_golden_recipe_glibc_stream(orc, golden_dir):
    """All 8 samples regenerated from the glibc rand() stream (main.cu:10-13, 54, 64);
    skipped if this libc's rand() is not the one the fixtures were drawn with."""
    z = np.load(f"{golden_dir}/golden_recipe.npz")
    for i, (k, m, n, q, r) in enumerate(orc.ref_recipe(SAMPLES, seed=1000)):
        fnv = z[f"s{i}_input_fnv"]
        if orc.fnv1a64(q) != int(fnv[0]) or orc.fnv1a64(r) != int(fnv[1]):
            pytest.skip("libc rand() stream differs from the fixture's")
        if m * n > (1 << 26) // 4:
            idx, _ = orc.v0_search(q, r, threads=8)
        else:
            idx, _ = orc.v0_search(q, r)
        assert np.array_equal(idx, z[f"s{i}_idx"]), f"sample {i}"
        if i in SURVEY_FIRST:
            assert idx[:len(SURVEY_FIRST[i])].tolist() == SURVEY_FIRST[i]


def test_oracle_matches_golden_adversarial(orc, golden_dir):
    z, names = _cases(golden_dir)
    assert len(names) >= 15
    for name in names:
        with np.errstate(all="ignore"):
            idx, _ = orc.v0_search(z[f"{name}__q"], z[f"{name}__r"])
        assert np.array_equal(idx, z[f"{name}__idx"]), name


def test_oracle_vs_reference_binary_random(orc):
    """Index-for-index against the reference's own V0 (only where oracle/_ref exists)."""
    if not orc.have_reference():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(7)
    for (k, m, n) in [(3, 50, 2000), (16, 20, 3000), (128, 8, 4000), (1, 9, 100), (7, 33, 513)]:
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        idx, dist = orc.v0_search(q, r)
        assert np.array_equal(idx, orc.v0_reference(q, r))
        # the distance we report is V0's minSum: recompute pairwise, bit for bit
        for i in range(m):
            assert dist[i].view(np.uint32) == orc.pair_distance(q[i], r[idx[i]]).view(np.uint32)


def test_oracle_tie_nan_rules(orc):
    """lowest index wins exact ties; NaN/INF never selected; empty -> 0."""
    r = np.array([[1.0, 1.0], [0.5, 0.5], [0.5, 0.5], [np.nan, 0.0]], np.float32)
    q = np.array([[0.5, 0.5], [np.nan, 0.0], [np.inf, 0.0]], np.float32)
    with np.errstate(all="ignore"):
        idx, dist = orc.v0_search(q, r)
    assert idx.tolist() == [1, 0, 0]
    assert dist[0] == 0.0 and np.isinf(dist[1]) and np.isinf(dist[2])


def test_oracle_sharded_merge_equals_unsharded(orc):
    rng = np.random.default_rng(3)
    q = rng.random((40, 8), dtype=np.float32)
    r = rng.random((1001, 8), dtype=np.float32)
    r[900] = r[10]
    q[0] = r[10]
    want, wd = orc.v0_search(q, r)
    for shards in (2, 3, 4, 8, 1001):
        idx, d = orc.v0_search_sharded(q, r, shards)
        assert np.array_equal(idx, want) and np.array_equal(d.view(np.uint32), wd.view(np.uint32))


def test_oracle_omp_equals_serial(orc):
    rng = np.random.default_rng(4)
    q = rng.random((64, 16), dtype=np.float32)
    r = rng.random((3000, 16), dtype=np.float32)
    a, ad = orc.v0_search(q, r)
    b, bd = orc.v0_search(q, r, threads=4)
    assert np.array_equal(a, b) and np.array_equal(ad.view(np.uint32), bd.view(np.uint32))


def test_rng_reproducible_and_uniform(orc):
    a = orc.rng_uniform(100000, seed=1000)
    b = orc.rng_uniform(100000, seed=1000)
    c = orc.rng_uniform(50000, seed=1000, offset=50000)
    assert np.array_equal(a, b) and np.array_equal(a[50000:], c)
    assert a.min() >= 0.0 and a.max() < 1.0 and abs(a.mean() - 0.5) < 0.01
    assert not np.array_equal(a, orc.rng_uniform(100000, seed=1001))


def test_bf16_rounding(orc):
    x = np.array([1.0, 1.00390625, 1.005859375, 3.14159274, -2.7182817, 0.0, np.inf], np.float32)
    y = orc.round_bf16(x)
    assert (y.view(np.uint32) & 0xFFFF == 0).all()
    assert y[0] == 1.0 and y[1] == 1.0 and y[2] == np.float32(1.0078125)   # ties-to-even, round up
    assert np.isnan(orc.round_bf16(np.array([np.nan], np.float32))[0])


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 16])
def test_kdtree_comparator_equals_v0(k):
    """oracle/kdtree.c (the reference's V10 family as an additional CPU comparator): the same indices
    and distance bits as V0 on random, clustered and duplicate-heavy clouds, ties included."""
    import __graft_entry__ as graft
    orc = graft.load_oracle()
    rng = np.random.default_rng(100 + k)
    for m, n in ((300, 5000), (64, 37), (1000, 20000)):
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        r[n // 2:n // 2 + n // 10] = r[:n // 10]                      # exact duplicates: lowest index wins
        q[:m // 4] = r[rng.integers(0, n, m // 4)]                   # exact hits (distance +0, several refs tie)
        if k >= 2:                                                   # a lattice: many equidistant refs
            r[-200:] = np.round(r[-200:] * 4) / 4
            q[-50:] = np.round(q[-50:] * 4) / 4 + np.float32(0.125)
        want_idx, want_dist = orc.v0_search(q, r, threads=4)
        got_idx, got_dist = orc.kdtree_search(q, r, threads=4)
        assert np.array_equal(got_idx, want_idx)
        assert np.array_equal(got_dist.view(np.uint32), want_dist.view(np.uint32))


def test_octree_comparator_equals_v0(orc):
    """oracle/octree.c (the reference's V12 family — CPU octree, 3-D only — as an additional CPU
    comparator, with V12's point-indexing bug fixed): the same indices and distance bits as V0 on
    uniform, clustered, duplicated (lowest index wins), degenerate (all points on a plane / one point)
    and offset clouds."""
    rng = np.random.default_rng(3112)
    cases = []
    q = rng.random((500, 3), dtype=np.float32)
    r = rng.random((20000, 3), dtype=np.float32)
    cases.append((q, r))
    c = rng.random((6, 3), dtype=np.float32) * 50
    cases.append(((c[rng.integers(0, 6, 300)] + rng.normal(0, 0.2, (300, 3))).astype(np.float32),
                  (c[rng.integers(0, 6, 9000)] + rng.normal(0, 0.2, (9000, 3))).astype(np.float32)))
    base = rng.random((700, 3), dtype=np.float32)
    rd = np.concatenate([base, base[:300], base])            # exact duplicates
    cases.append((base[:200].copy(), rd))
    flat = rng.random((4000, 3), dtype=np.float32)
    flat[:, 2] = 0.25                                        # all refs in one plane
    cases.append((rng.random((100, 3), dtype=np.float32), flat))
    cases.append((rng.random((10, 3), dtype=np.float32), rng.random((1, 3), dtype=np.float32)))
    cases.append(((1000.0 + rng.random((100, 3))).astype(np.float32), (1000.0 + rng.random((5000, 3))).astype(np.float32)))
    cases.append((rng.random((50, 3), dtype=np.float32), np.tile(rng.random((1, 3), dtype=np.float32), (100, 1))))
    for q, r in cases:
        want_idx, want_dist = orc.v0_search(q, r, threads=4)
        got_idx, got_dist = orc.octree_search(q, r, threads=4)
        assert np.array_equal(got_idx, want_idx)
        assert np.array_equal(got_dist.view(np.uint32), want_dist.view(np.uint32))
    with pytest.raises(RuntimeError):
        orc.octree_search(rng.random((4, 2), dtype=np.float32), rng.random((9, 2), dtype=np.float32))
