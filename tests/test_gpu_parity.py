"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against
the oracle (V0 restatement) and the committed golden vectors.  Bar: indices
bit-exact; distances bit-equal (tolerance 0 ulp: the returned distance IS V0's
fp32 minSum, recomputed with V0's arithmetic on the device)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu



def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check(pkg, orc, q, r, paths=("auto",), shards=(1,), want=None):
    with np.errstate(all="ignore"):
        want_idx, want_dist = orc.v0_search(q, r, threads=8)
    if want is not None:
        assert np.array_equal(want_idx, want), "oracle disagrees with the golden vector"
    # the MFMA filter has two candidate-record forms — per (lane, ref tile) on short ref streams, per score on long
    # ones; "mfma" drives whatever the planner picks for the size (small tests: tiles), "mfma_perref" forces the other
    if "mfma" in paths and "mfma_perref" not in paths:
        paths = tuple(paths) + ("mfma_perref",)
    for path in paths:
        if path.startswith("mfma") and q.shape[1] > 256:
            continue
        for s in shards:
            idx, dist = pkg.search(q, r, return_distances=True, shards=s, path=path)
            bad = np.nonzero(idx != want_idx)[0]
            assert bad.size == 0, f"path={path} shards={s}: {bad.size} index mismatches, first at query {bad[:5]}"
            assert np.array_equal(_bits(dist), _bits(want_dist)), f"path={path} shards={s}: distance bits differ"


def test_native_library_loaded_on_gpu(pkg):
    """The tests must run the HIP extension, not a fallback."""
    assert torch.cuda.is_available()
    assert pkg.device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libnns_mi355x.so" in f.read()


def test_mfma_is_an_fmaf_chain(pkg, orc):
    """v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain, bit for bit (assumed by tau)."""
    rng = np.random.default_rng(1)
    for kt in (64, 128):
        a = (rng.random((32, kt), dtype=np.float32) - 0.5) * 4
        b = (rng.random((32, kt), dtype=np.float32) - 0.5) * 4
        c0 = rng.random(32, dtype=np.float32) * 10
        out = pkg.selftest_mfma(a, b, c0)
        # k order of the kernel: step s contracts k = 8(s>>2) + (s&3) then the same + 4
        order = []
        for s in range(kt // 2):
            base = 8 * (s >> 2) + (s & 3)
            order += [base, base + 4]
        order = np.array(order)
        for i in range(32):
            for j in range(0, 32, 5):
                want = orc.fmaf_chain(a[i, order], b[j, order], float(c0[i]))
                assert out[i, j].view(np.uint32) == want.view(np.uint32), (kt, i, j)


def _bf16_model_cases(rng, kt):
    """(name, a[32][kt], b[32][kt]) operand families for the accumulate-model test: a = refs (x -2),
    b = queries.  fp32 values; the test rounds them to bf16."""
    u01 = lambda: rng.random((32, kt), dtype=np.float32)                       # noqa: E731
    cases = [("uniform", u01() * -2.0, u01()),
             ("same_sign_big", (1.0 + u01()) * -2.0, 1.0 + u01()),             # every product negative: no cancellation
             ("same_sign_pos", (1.0 + u01()) * 2.0, 1.0 + u01()),
             ("cancelling", (u01() - 0.5) * -2.0, u01() - 0.5),               # products of both signs, sum ~ 0
             ("centred_small", (u01() - 0.5) * -2e-3, (u01() - 0.5) * 1e-3)]
    e = rng.integers(-12, 12, (32, kt))
    cases.append(("wide_exponents", (np.ldexp(1.0 + u01(), e) * -2.0).astype(np.float32),
                  np.ldexp(1.0 + u01(), rng.integers(-12, 12, (32, kt))).astype(np.float32)))
    # one huge term first / last among tiny ones: tests the order-independence assumption
    big_first = u01() * 1e-3
    big_first[:, 0] = 300.0
    big_last = u01() * 1e-3
    big_last[:, -1] = 300.0
    cases.append(("big_first", big_first * -2.0, big_first.copy()))
    cases.append(("big_last", big_last * -2.0, big_last.copy()))
    return cases


@pytest.mark.parametrize("shape", [2, 1])
@pytest.mark.parametrize("kt", [128, 256, 384, 512, 640, 768, 1024])
def test_bf16_mfma_error_model(pkg, orc, shape, kt):
    """The bf16 MFMA's accumulation (fp32 accumulate, order and rounding undocumented) against fp64 on
    bf16-representable operands, at EVERY depth the product runs bf16 tiles (128 / 256 / 512 / 1024), for both
    MFMA shapes (2 = v_mfma_f32_16x16x32_bf16 with the operand / result lane mapping K2's images and the
    filter's epilogue assume — a wrong mapping shows up as O(1) errors here; 1 = 32x32x16), several seeds
    and adversarial magnitudes.  The proof margin tau assumes 2u per add (nns_internal.h tau_consts,
    mode 1): the test FAILS if the hardware error of any output exceeds 1/4 of that e3 bound."""
    u = 2.0 ** -24
    worst = 0.0
    for seed in (2, 3, 5, 8):
        rng = np.random.default_rng(seed * 1000 + kt + shape)
        for name, a32, b32 in _bf16_model_cases(rng, kt):
            a = orc.round_bf16(a32)
            b = orc.round_bf16(b32)
            a64, b64 = a.astype(np.float64), b.astype(np.float64)
            c0 = ((a64 / 2) ** 2).sum(1).astype(np.float32)           # the refs' squared norms (a = -2 y)
            out = pkg.selftest_mfma(a, b, c0, bf16=shape)
            exact = c0.astype(np.float64)[:, None] + a64 @ b64.T
            err = np.abs(out.astype(np.float64) - exact)                # [ref i][query j]
            # the model's bound for query j: e3 = (c0_tau / (2 + c1)) / 1.001 with X^2 = |x_j|^2, Y^2 = max |y|^2
            y2max = float(((a64 / 2) ** 2).sum(1).max())
            for j in range(32):
                x2 = float((b64[j] ** 2).sum())
                c0t, c1t, _ = pkg.tau_consts(kt, x2, y2max, 1)
                e3 = c0t / (2.0 + c1t) / 1.001
                ratio = err[:, j].max() / e3
                worst = max(worst, ratio)
                assert ratio <= 0.25, (name, seed, kt, shape, j, err[:, j].max(), e3)
            # and elementwise inside the plain 2u-per-add bound on the actual magnitudes
            mag = np.abs(c0.astype(np.float64))[:, None] + np.abs(a64) @ np.abs(b64).T
            assert (err <= 2 * (kt + kt // 16 + 2) * u * mag).all(), (name, seed, kt, shape)
    print(f"bf16 MFMA shape {shape} kt {kt}: worst error / e3 bound = {worst:.4f}")


@pytest.mark.parametrize("kt", [128, 256, 384, 512, 640, 768, 1024])
def test_bf16_operand_rounding_model_mode2(pkg, orc, kt):
    """tau mode 2 (fp32 points, operands rounded to bf16, NNS_FILTER_BF16): the filter's score error
    against the UNROUNDED fp32 values — rounding bound 2^-6 (1 + 2^-8) |x'||y'| plus the accumulate
    model — must hold, also on data placed at bf16 rounding midpoints with every product of the same
    sign (where the rounding bound is nearly attained; random data sits ~1/sqrt(k) below it)."""
    for seed in (1, 4):
        rng = np.random.default_rng(seed * 77 + kt)
        fams = [("uniform", rng.random((32, kt), dtype=np.float32) - 0.5, rng.random((32, kt), dtype=np.float32) - 0.5)]
        # just below a rounding midpoint: 1 + 2^-8 - 2^-20 rounds DOWN to 1 (relative error ~2^-8), same sign
        mid = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20)
        sc = np.ldexp(np.float32(1.0), rng.integers(-3, 3, (32, 1))).astype(np.float32)
        fams.append(("midpoints_same_sign", np.full((32, kt), mid, np.float32) * sc, np.full((32, kt), mid, np.float32)))
        fams.append(("midpoints_up", np.full((32, kt), np.float32(1.0 + 2.0 ** -8 + 2.0 ** -20)) * sc,
                     np.full((32, kt), np.float32(1.0 + 2.0 ** -8 + 2.0 ** -20))))
        for name, y, x in fams:
            a_true = (-2.0 * y).astype(np.float32)                      # exact scaling
            a, b = orc.round_bf16(a_true), orc.round_bf16(x)            # what K2 writes (RNE)
            c0 = (y.astype(np.float64) ** 2).sum(1).astype(np.float32)
            out = pkg.selftest_mfma(a, b, c0, bf16=2 if kt < 512 else 1)
            exact = c0.astype(np.float64)[:, None] + a_true.astype(np.float64) @ x.astype(np.float64).T
            err = np.abs(out.astype(np.float64) - exact)
            y2max = float((y.astype(np.float64) ** 2).sum(1).max())
            for j in range(32):
                x2 = float((x[j].astype(np.float64) ** 2).sum())
                c0t, c1t, _ = pkg.tau_consts(kt, x2, y2max, 2)
                bound = c0t / (2.0 + c1t) / 1.001                       # e3 + e2 of mode 2
                assert err[:, j].max() <= bound, (name, seed, kt, j, err[:, j].max(), bound)
                if name == "uniform":
                    assert err[:, j].max() <= 0.25 * bound


def test_golden_recipe_samples(pkg, orc, golden_dir):
    """The reference driver's own samples (main.cu:38-47) on its srand(1000) stream."""
    z = np.load(f"{golden_dir}/golden_recipe.npz")
    for i in range(8):
        if f"s{i}_q" not in z.files:
            continue
        k, m, n = (int(v) for v in z[f"s{i}_shape"])
        q, r, want = z[f"s{i}_q"], z[f"s{i}_r"], z[f"s{i}_idx"]
        got = pkg.cudaCall(k, m, n, q, r)
        assert np.array_equal(got, want), f"sample {i} (k={k} m={m} n={n})"
        _check(pkg, orc, q, r, paths=("auto", "exact"), shards=(1, 3), want=want)


def test_golden_recipe_glibc_big_samples(pkg, orc, golden_dir):
    """Samples 6 - 9 (1024 x 65536 and 1024 x 1048576) regenerated from the glibc stream."""
    z = np.load(f"{golden_dir}/golden_recipe.npz")
    samples = [tuple(int(v) for v in z[f"s{i}_shape"]) for i in range(10)]
    for i, (k, m, n, q, r) in enumerate(orc.ref_recipe(samples, seed=1000)):
        fnv = z[f"s{i}_input_fnv"]
        if orc.fnv1a64(q) != int(fnv[0]) or orc.fnv1a64(r) != int(fnv[1]):
            pytest.skip("libc rand() stream differs from the fixture's")
        if i < 6:
            continue
        got = pkg.cudaCall(k, m, n, q, r)
        assert np.array_equal(got, z[f"s{i}_idx"]), f"sample {i}"


def test_golden_adversarial_all_paths(pkg, orc, golden_dir):
    z = np.load(f"{golden_dir}/golden_cases.npz")
    names = sorted({k.split("__")[0] for k in z.files})
    for name in names:
        q, r, want = z[f"{name}__q"], z[f"{name}__r"], z[f"{name}__idx"]
        _check(pkg, orc, q, r, paths=("auto", "exact", "mfma"), shards=(1, 2, 5), want=want)


@pytest.mark.parametrize("shape", [(700, 20001, 128), (33, 4097, 64), (1, 130, 128), (513, 63, 100),
                                   (2049, 777, 37), (300, 66000, 128)])
def test_filter_ragged_shapes(pkg, orc, shape):
    """The MFMA filter on ragged m / n / k (padding of queries, refs and dims) vs the oracle."""
    m, n, k = shape
    rng = np.random.default_rng(100 + m)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    _check(pkg, orc, q, r, paths=("mfma",), shards=(1, 2))


def test_filter_sorted_refs_ring_lists(pkg, orc):
    """Refs ordered so that every ref is a new record for the query (monotone approach): the
    per-lane candidate lists wrap around (old records fall above the shrinking threshold and
    are overwritten) — parity must hold without needing the exact scan for every query."""
    rng = np.random.default_rng(77)
    k, n = 64, 40000      # 256 ref-range splits -> ~78 refs per lane per split > the list capacity
    base = rng.random((1, k), dtype=np.float32)
    direction = rng.normal(0, 1, (1, k)).astype(np.float32)
    direction /= np.linalg.norm(direction)
    steps = np.linspace(30.0, 0.5, n, dtype=np.float32)[:, None]      # marching towards the query
    r = (base + steps * direction).astype(np.float32)
    q = (base + rng.normal(0, 1e-3, (40, k))).astype(np.float32)
    _check(pkg, orc, q, r, paths=("mfma",), shards=(1, 3))
    for path in ("mfma", "mfma_perref"):
        ix = pkg.Index(torch.from_numpy(r).cuda(), path=path)
        ix.search(torch.from_numpy(q).cuda())
        st = ix.stats()
        assert st["path"] == 2 and st["ambiguous"] < q.shape[0], (path, st)
        ix.close()


def test_filter_true_list_overflow_falls_back(pkg, orc):
    """80000 exact duplicates of the nearest ref: every copy is a live candidate,
    the ring lists genuinely overflow, and the exact scan must return the LOWEST duplicate."""
    rng = np.random.default_rng(78)
    k = 128
    base = rng.random((2000, k), dtype=np.float32)
    hot = rng.random((1, k), dtype=np.float32)
    r = np.concatenate([base[:700], np.repeat(hot, 80000, axis=0), base[700:]])
    q = np.concatenate([hot + rng.normal(0, 1e-4, (24, k)).astype(np.float32), rng.random((40, k), dtype=np.float32)])
    _check(pkg, orc, q, r, paths=("mfma",), shards=(1, 2))
    # every record form must notice that it cannot hold all live candidates and hand the query to the exact scan: a
    # record per score (80000 live candidates against 64-entry rings), and the short-stream form of 32 x 32 tiles (a
    # lane's two best tiles: here ~10 tiles of its stream hold copies of the nearest ref)
    assert pkg.plan_filter(k, q.shape[0], r.shape[0])["tile_rec"] == 2
    for path in ("mfma_perref", "mfma"):
        ix = pkg.Index(torch.from_numpy(r).cuda(), path=path)
        idx = ix.search(torch.from_numpy(q).cuda())
        st = ix.stats()
        assert st["ambiguous"] >= 24, (path, st)
        assert (idx[:24].cpu().numpy() == 700).all()
        ix.close()
    # a record per (lane, ref tile) behind the threshold test — the short-stream form of the 16 x 16 bf16 tiles —
    # overflows only when more than 64 TILES of one stream hold live candidates: 600000 duplicates over 256 streams
    k = 32
    base = orc.round_bf16(rng.random((2000, k), dtype=np.float32))
    hot = orc.round_bf16(rng.random((1, k), dtype=np.float32))
    r = np.concatenate([base[:700], np.repeat(hot, 600000, axis=0), base[700:]])
    q = orc.round_bf16(np.concatenate([hot + rng.normal(0, 1e-2, (24, k)).astype(np.float32), rng.random((40, k), dtype=np.float32)]))
    assert pkg.plan_filter(k, q.shape[0], r.shape[0], bf16=True)["tile_rec"] == 1
    ix = pkg.Index(torch.from_numpy(r).cuda().to(torch.bfloat16), path="mfma")
    idx = ix.search(torch.from_numpy(q).cuda().to(torch.bfloat16))
    st = ix.stats()
    want_idx, _ = orc.v0_search(q, r, threads=8)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert st["ambiguous"] >= 20 and (want_idx[:24] == 700).sum() >= 20, st
    ix.close()


def test_low_dim_shapes_exact_path(pkg, orc):
    rng = np.random.default_rng(9)
    for (m, n, k) in [(4096, 8192, 3), (64, 100000, 3), (1000, 5000, 16), (63, 777, 2), (5, 50000, 8),
                      (1, 65536, 16), (300, 3000, 7),
                      # a handful of refs: one wave per workgroup must still finish BOTH query slots of a lane
                      (1000, 1, 8), (1000, 5, 3), (513, 31, 16), (2049, 64, 4), (200, 33, 1)]:
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        _check(pkg, orc, q, r, paths=("auto",), shards=(1, 4))


def test_near_duplicate_refs_are_reranked(pkg, orc):
    """1-ulp twins of the true neighbours: their filter scores are within tau, so the
    exact re-rank of the candidate lists must decide (lowest index on exact ties)."""
    rng = np.random.default_rng(21)
    base = rng.random((3000, 128), dtype=np.float32)
    dup = base[:1000].copy()
    dup[:, 7] = np.nextafter(dup[:, 7], np.float32(2.0))      # 1-ulp twins of 1000 refs
    r = np.concatenate([base, dup])
    q = base[:256] + rng.normal(0, 1e-4, (256, 128)).astype(np.float32)
    _check(pkg, orc, q, r, paths=("mfma",), shards=(1, 2))
    ix = pkg.Index(torch.from_numpy(r).cuda(), path="mfma")
    ix.search(torch.from_numpy(q).cuda())
    st = ix.stats()
    assert st["path"] == 2, st       # near-ties are resolved from the candidate lists
    ix.close()


def test_device_api_keys_and_shards(pkg, orc):
    """Split API: per-shard keys with index_base, merged with nns_keys_min == V0."""
    rng = np.random.default_rng(33)
    q = rng.random((500, 128), dtype=np.float32)
    r = rng.random((9000, 128), dtype=np.float32)
    r[8000] = r[5]
    q[0] = r[5]
    want_idx, want_dist = orc.v0_search(q, r, threads=8)
    qd = torch.from_numpy(q).cuda()
    rd = torch.from_numpy(r).cuda()
    shards = 4
    keys = None
    for s in range(shards):
        beg, cnt = pkg.shard_range(r.shape[0], shards, s)
        ix = pkg.Index(rd[beg:beg + cnt], index_base=beg)
        ks = ix.search_keys(qd)
        if keys is None:
            keys = ks
        else:
            pkg.keys_min(keys, ks)
        torch.cuda.synchronize()
        ix.close()
    idx, dist = pkg.keys_unpack(keys, return_distances=True)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(_bits(dist.cpu().numpy()), _bits(want_dist))
    # keys are plain (dist bits << 32) | idx
    kk = keys.cpu().numpy()
    assert np.array_equal((kk >> 32).astype(np.uint32), _bits(want_dist))


def test_fill_uniform_matches_oracle_bits(pkg, orc):
    t = torch.empty(100003, dtype=torch.float32, device="cuda")
    pkg.fill_uniform(t, seed=1000, offset=12345)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), orc.rng_uniform(100003, 1000, 12345))


def test_determinism_bitwise_repeatable(pkg):
    q = torch.empty((2048, 128), dtype=torch.float32, device="cuda")
    r = torch.empty((50000, 128), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1)
    pkg.fill_uniform(r, 2)
    ix = pkg.Index(r)
    a = ix.search_keys(q).clone()
    for _ in range(3):
        b = ix.search_keys(q)
        assert torch.equal(a, b)
    ix.close()


def test_errors_are_status_codes(pkg):
    q = np.zeros((4, 300), np.float32)
    with pytest.raises(pkg.NNSError) as e:
        pkg.search(q, q, path="mfma")          # k > 256 not tiled on the MFMA path
    assert e.value.status == 5


@pytest.mark.timeout(900)
def test_headline_shape_properties(pkg, orc):
    """BASELINE C3 (65536 x 1048576 x 128) at full size: sampled queries against the
    oracle, planted exact matches, shard invariance of the keys, distances bit-equal to
    V0's pair arithmetic."""
    m, n, k = 65536, 1048576, 128
    q = torch.empty((m, k), dtype=torch.float32, device="cuda")
    r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1000, 0)
    pkg.fill_uniform(r, 1000, m * k)
    # plant: query i*97 is an exact copy of ref p(i) -> distance +0, index p(i)
    planted_q = torch.arange(0, 512, device="cuda") * 97
    planted_r = (torch.arange(0, 512, device="cuda") * 2039 + 17) % n
    q[planted_q] = r[planted_r]
    ix = pkg.Index(r)
    keys = ix.search_keys(q)
    idx, dist = pkg.keys_unpack(keys, return_distances=True)
    torch.cuda.synchronize()
    st = ix.stats()
    assert st["path"] == 2 and st["nonfinite"] == 0
    assert st["ambiguous"] < m // 20, st
    idx_h, dist_h = idx.cpu().numpy(), dist.cpu().numpy()
    assert np.array_equal(idx_h[planted_q.cpu().numpy()], planted_r.cpu().numpy().astype(np.int32))
    assert (dist_h[planted_q.cpu().numpy()] == 0).all()
    assert idx_h.min() >= 0 and idx_h.max() < n
    # SURVEY 8d's gate: >= 1024 RANDOM queries + EVERY query whose filter margin was below tau (K5 chose
    # among several candidates: nns_index_near_ties), against the oracle over ALL refs (queries sent to
    # the exact scan — none at this shape — get V0's arithmetic over all refs by construction)
    rh = r.cpu().numpy()
    near = ix.near_ties()
    assert near.size == st["multi_candidate"] and near.size < m // 4, (near.size, st)
    sel = np.unique(np.concatenate([np.random.default_rng(0).choice(m, 1024, replace=False), near]))
    print(f"C3: {near.size} near-tie queries, {st['ambiguous']} exact-scan queries, checking {sel.size} queries in full")
    qh = q[torch.from_numpy(sel).cuda()].cpu().numpy()
    want_idx, want_dist = orc.v0_search(qh, rh, threads=16)
    assert np.array_equal(idx_h[sel], want_idx)
    assert np.array_equal(_bits(dist_h[sel]), _bits(want_dist))
    # every returned distance is V0's arithmetic on (q_i, r_idx_i): check a slice on the host
    for i in range(0, m, 4099):
        assert _bits(dist_h[i:i + 1])[0] == orc.pair_distance(q[i].cpu().numpy(), rh[idx_h[i]]).view(np.uint32)
    ix.close()
    # shard invariance: 2 shards merged == unsharded keys, bit for bit
    half = n // 2
    k0 = pkg.Index(r[:half], index_base=0)
    a = k0.search_keys(q).clone()
    torch.cuda.synchronize()
    k0.close()
    k1 = pkg.Index(r[half:], index_base=half)
    b = k1.search_keys(q)
    pkg.keys_min(a, b)
    torch.cuda.synchronize()
    k1.close()
    assert torch.equal(a, keys)


# ---------------------------------------------------------------------------
# bf16 points (config C5): oracle = V0 arithmetic on the bf16 values widened to fp32
# ---------------------------------------------------------------------------
def _check_bf16(pkg, orc, q, r, paths=("auto",), shards=(1,)):
    qb, rb = pkg.to_bf16_bits(q), pkg.to_bf16_bits(r)
    qw, rw = orc.round_bf16(q), orc.round_bf16(r)
    assert np.array_equal((qw.view(np.uint32) >> 16).astype(np.uint16), qb)   # same rounding as the oracle
    with np.errstate(all="ignore"):
        want_idx, want_dist = orc.v0_search(qw, rw, threads=8)
    if "mfma" in paths and "mfma_perref" not in paths:   # both candidate-record forms (see _check)
        paths = tuple(paths) + ("mfma_perref",)
    for path in paths:
        if path.startswith("mfma") and q.shape[1] > 1024:
            continue
        for s in shards:
            idx, dist = pkg.search_bf16(qb, rb, return_distances=True, shards=s, path=path)
            bad = np.nonzero(idx != want_idx)[0]
            assert bad.size == 0, f"bf16 path={path} shards={s}: {bad.size} mismatches, first {bad[:5]}"
            assert np.array_equal(_bits(dist), _bits(want_dist)), f"bf16 path={path} shards={s}: distance bits"


@pytest.mark.parametrize("shape", [(300, 5000, 256), (700, 20001, 256), (33, 4097, 128), (1, 130, 256),
                                   (513, 63, 200), (64, 3000, 40), (100, 2000, 16), (50, 999, 300)])
def test_bf16_random_shapes(pkg, orc, shape):
    m, n, k = shape
    rng = np.random.default_rng(500 + m)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    _check_bf16(pkg, orc, q, r, paths=("auto", "exact", "mfma"), shards=(1, 2))


def test_bf16_ties_and_specials(pkg, orc):
    rng = np.random.default_rng(8)
    r = rng.random((400, 256), dtype=np.float32)
    r = np.concatenate([r, r])                 # bf16 rounding also creates many exact ties
    q = r[:100] + rng.normal(0, 1e-2, (100, 256)).astype(np.float32)
    _check_bf16(pkg, orc, q, r, paths=("auto", "exact"), shards=(1, 3))
    r2 = rng.random((300, 64), dtype=np.float32)
    r2[5, 3] = np.nan
    r2[9, 1] = np.inf
    q2 = rng.random((20, 64), dtype=np.float32)
    q2[2, 0] = np.nan
    _check_bf16(pkg, orc, q2, r2, paths=("auto", "exact"), shards=(1, 2))


def test_bf16_device_api(pkg, orc):
    rng = np.random.default_rng(9)
    q = rng.random((600, 256), dtype=np.float32)
    r = rng.random((30000, 256), dtype=np.float32)
    want_idx, want_dist = orc.v0_search(orc.round_bf16(q), orc.round_bf16(r), threads=8)
    qd = torch.from_numpy(q).cuda().to(torch.bfloat16)
    rd = torch.from_numpy(r).cuda().to(torch.bfloat16)
    ix = pkg.Index(rd)
    idx, dist = ix.search(qd, return_distances=True)
    torch.cuda.synchronize()
    st = ix.stats()
    assert st["path"] == 2 and st["k_tile"] == 256, st
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(_bits(dist.cpu().numpy()), _bits(want_dist))
    ix.close()


def test_in_library_multi_gpu_path(pkg, orc):
    """nns_search_f32_multi (the V8/V9 analogue): all visible GPUs, and a virtual rehearsal
    with more shards than GPUs (one host thread per shard, host-side key merge)."""
    rng = np.random.default_rng(55)
    for (m, n, k) in [(300, 40000, 128), (1000, 300000, 3), (5, 7, 16)]:
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        r[n - 1] = r[0]
        q[0] = r[0]                       # cross-shard exact tie -> lowest index
        want_idx, want_dist = orc.v0_search(q, r, threads=8)
        for kwargs in ({"num_devices": 0}, {"num_devices": 3, "virtual": True}, {"num_devices": 8, "virtual": True}):
            idx, dist = pkg.search_multi(q, r, return_distances=True, **kwargs)
            assert np.array_equal(idx, want_idx), kwargs
            assert np.array_equal(_bits(dist), _bits(want_dist)), kwargs


def test_extreme_magnitudes(pkg, orc):
    """Denormal-scale clouds (every squared distance underflows: all ties -> index 0 region)
    and huge clouds (squares overflow: the error bound is void, exact kernels must take over)."""
    rng = np.random.default_rng(91)
    for scale, k in ((1e-25, 128), (1e-20, 64), (1e19, 128), (3e18, 16), (1e-38, 3)):
        q = (rng.random((70, k), dtype=np.float32) * np.float32(scale)).astype(np.float32)
        r = (rng.random((3000, k), dtype=np.float32) * np.float32(scale)).astype(np.float32)
        with np.errstate(all="ignore"):
            _check(pkg, orc, q, r, paths=("auto", "mfma") if k >= 32 else ("auto",), shards=(1, 2))


def test_offset_clouds_need_centring(pkg, orc):
    """Clouds far from the origin with tiny spread: un-centred GEMM scores would lose every
    significant bit; the mean-centred filter plus the tau re-rank must still return V0's index."""
    rng = np.random.default_rng(92)
    for off, spread in ((1000.0, 1.0), (1.0e5, 10.0), (-3.0e4, 0.5)):
        q = (off + spread * rng.random((130, 128))).astype(np.float32)
        r = (off + spread * rng.random((6000, 128))).astype(np.float32)
        _check(pkg, orc, q, r, paths=("mfma",), shards=(1, 3))


@pytest.mark.timeout(600)
def test_midsize_all_queries_vs_oracle(pkg, orc):
    """Every query of a mid-size problem (fp32 4096 x 262144 x 128, bf16 2048 x 131072 x 256)
    against the oracle over all refs — the largest shapes the CPU oracle finishes in seconds."""
    m, n, k = 4096, 262144, 128
    q = orc.rng_uniform(m * k, 77, 0).reshape(m, k)
    r = orc.rng_uniform(n * k, 77, m * k).reshape(n, k)
    want_idx, want_dist = orc.v0_search(q, r, threads=16)
    idx, dist = pkg.search(q, r, return_distances=True)
    assert np.array_equal(idx, want_idx)
    assert np.array_equal(_bits(dist), _bits(want_dist))
    m, n, k = 2048, 131072, 256
    q = orc.rng_uniform(m * k, 78, 0).reshape(m, k)
    r = orc.rng_uniform(n * k, 78, m * k).reshape(n, k)
    want_idx, want_dist = orc.v0_search(orc.round_bf16(q), orc.round_bf16(r), threads=16)
    idx, dist = pkg.search_bf16(pkg.to_bf16_bits(q), pkg.to_bf16_bits(r), return_distances=True)
    assert np.array_equal(idx, want_idx)
    assert np.array_equal(_bits(dist), _bits(want_dist))


def test_c2_shape_all_queries(pkg, orc):
    """BASELINE config C2 at full size (4096 x 65536 x 3): every query vs the oracle."""
    m, n, k = 4096, 65536, 3
    q = orc.rng_uniform(m * k, 1000, 0).reshape(m, k)
    r = orc.rng_uniform(n * k, 1000, m * k).reshape(n, k)
    want_idx, want_dist = orc.v0_search(q, r, threads=16)
    idx, dist = pkg.search(q, r, return_distances=True)
    assert np.array_equal(idx, want_idx) and np.array_equal(_bits(dist), _bits(want_dist))


@pytest.mark.timeout(1100)
def test_c4_shape_on_one_gpu_by_shards(pkg, orc):
    """BASELINE config C4 (65536 x 8388608 x 128, refs sharded 8 ways) rehearsed on ONE GPU: the 8
    shards are searched one after another with their index_base and merged with nns_keys_min —
    exactly what 8 ranks + the min all-reduce compute.  Checked: planted exact matches in every
    shard, sampled queries against the oracle over all 8M refs, index range."""
    m, n, k, shards = 65536, 8388608, 128, 8
    q = torch.empty((m, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1000, 0)
    per = n // shards
    keys = None
    r_host = np.empty((n, k), dtype=np.float32)
    planted_q = np.arange(shards * 16) * 401
    planted_r = np.empty(shards * 16, dtype=np.int64)
    near_all = []
    for s in range(shards):
        r = torch.empty((per, k), dtype=torch.float32, device="cuda")
        pkg.fill_uniform(r, 1000, m * k + s * per * k)       # the bench's global ref stream
        for t in range(16):                                    # plant 16 queries per shard
            qi = planted_q[s * 16 + t]
            lj = (t * 65537 + 11) % per
            r[lj] = q[qi]
            planted_r[s * 16 + t] = s * per + lj
        r_host[s * per:(s + 1) * per] = r.cpu().numpy()
        ix = pkg.Index(r, index_base=s * per)
        ks = ix.search_keys(q)
        if keys is None:
            keys = ks.clone()
        else:
            pkg.keys_min(keys, ks)
        torch.cuda.synchronize()
        near_all.append(ix.near_ties())       # this shard's queries decided among several candidates
        ix.close()
        del r
    idx, dist = pkg.keys_unpack(keys, return_distances=True)
    idx_h, dist_h = idx.cpu().numpy(), dist.cpu().numpy()
    assert idx_h.min() >= 0 and idx_h.max() < n
    assert np.array_equal(idx_h[planted_q], planted_r.astype(np.int32))
    assert (dist_h[planted_q] == 0).all()
    # >= 1024 random queries + every near-tie query of any shard, against the oracle over all 8M refs
    near = np.unique(np.concatenate(near_all))
    assert near.size < m // 2
    sel = np.unique(np.concatenate([np.random.default_rng(4).choice(m, 1024, replace=False), near]))
    print(f"C4: {near.size} near-tie queries over the 8 shards, checking {sel.size} queries in full")
    want_idx, want_dist = orc.v0_search(q[torch.from_numpy(sel).cuda()].cpu().numpy(), r_host, threads=16)
    assert np.array_equal(idx_h[sel], want_idx)
    assert np.array_equal(_bits(dist_h[sel]), _bits(want_dist))


def test_awkward_query_counts_balance(pkg, orc):
    """Query counts just past a workgroup/round boundary (the split chooser must keep the grid
    balanced; results must not depend on the split count)."""
    rng = np.random.default_rng(123)
    r = rng.random((30000, 128), dtype=np.float32)
    for m in (513, 1537, 5000):
        q = rng.random((m, 128), dtype=np.float32)
        _check(pkg, orc, q, r, paths=("mfma",), shards=(1,))


@pytest.mark.parametrize("shape", [(1000, 5000, 16), (300, 70001, 8), (2049, 777, 31), (65, 300000, 32),
                                   (700, 20001, 12), (4096, 4096, 20), (1024, 300001, 16), (130, 1100000, 9)])
def test_filter_k32_tile_shapes(pkg, orc, shape):
    """8 <= k <= 16 runs the 16-deep fp32 MFMA tile (16 image blocks = 512 refs per ring slot), 16 < k <= 32
    the 32-deep one (8 blocks): ragged m / n / k vs the oracle, forced and under AUTO, whole and sharded."""
    m, n, k = shape
    rng = np.random.default_rng(500 + k)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    _check(pkg, orc, q, r, paths=("mfma", "auto"), shards=(1, 3))
    ix = pkg.Index(torch.from_numpy(r).cuda())
    ix.search(torch.from_numpy(q).cuda())
    st = ix.stats()
    if (k == 8 and m * n <= (1 << 27)) or (k == 16 and (m * n <= (1 << 25) or (n < 8192 and m * n <= (1 << 26)))):     # small problems in a K1a dimensionality stay on the exact kernel under AUTO
        assert st["path"] == 1, st
    else:
        assert st["path"] == 2 and st["k_tile"] == (16 if k <= 16 else 32), st
    ix.close()


def test_k32_tile_duplicates_and_specials(pkg, orc):
    """Low-dimensional data collides far more often: exact duplicate refs (lowest index wins),
    1-ulp twins, and NaN / INF refs that must never be selected — through the 32-deep tile."""
    rng = np.random.default_rng(61)
    k = 16
    base = rng.random((5000, k), dtype=np.float32)
    twins = base[:500].copy()
    twins[:, 3] = np.nextafter(twins[:, 3], np.float32(2.0))
    r = np.concatenate([base, base[:800], twins])
    q = np.concatenate([base[:300], base[:200] + rng.normal(0, 1e-5, (200, k)).astype(np.float32)])
    _check(pkg, orc, q, r, paths=("mfma", "auto"), shards=(1, 2))
    r2 = r.copy()
    r2[17, 2] = np.nan
    r2[4000, 0] = np.inf
    _check(pkg, orc, q, r2, paths=("mfma", "auto"), shards=(1, 2))


def test_workspace_pool_reuse_and_trim(pkg, orc):
    """Whole-call searches park their device workspaces in the pool (dev_pool.hip): repeated and
    interleaved calls of different shapes give the same answers on recycled (dirty) memory, and
    nns_trim() hands the parked bytes back."""
    rng = np.random.default_rng(62)
    shapes = [(300, 9000, 128), (1000, 5000, 16), (64, 100000, 3), (300, 9000, 128), (129, 7000, 64)]
    data = [(rng.random((m, k), dtype=np.float32), rng.random((n, k), dtype=np.float32)) for m, n, k in shapes]
    want = [orc.v0_search(q, r)[0] for q, r in data]
    # each round recycles blocks the previous shapes left dirty (tile images, candidate lists)
    for rep in range(3):
        for (q, r), w in zip(data, want):
            assert np.array_equal(pkg.search(q, r), w)
    released = pkg.trim()
    assert released > 0
    assert pkg.trim() == 0
    for (q, r), w in zip(data, want):
        assert np.array_equal(pkg.search(q, r), w)


@pytest.mark.timeout(1100)
def test_c5_headline_shape_properties(pkg, orc):
    """BASELINE C5 (131072 x 2097152 x 256 bf16 points) at full size through the 16x16x32 bf16
    filter: planted exact matches, sampled queries against the oracle (V0 arithmetic on the bf16
    values) over ALL refs, distances bit-equal to V0's pair arithmetic, 2-shard invariance of the
    keys.  bf16 values collide far more often than fp32 ones, so ties are exercised for real."""
    m, n, k = 131072, 2097152, 256
    q = torch.empty((m, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1000, 0)
    qb = q.to(torch.bfloat16)
    del q
    r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(r, 1000, m * k)
    rb = r.to(torch.bfloat16)
    del r
    torch.cuda.empty_cache()
    planted_q = torch.arange(0, 512, device="cuda") * 251
    planted_r = (torch.arange(0, 512, device="cuda") * 4093 + 29) % n
    qb[planted_q] = rb[planted_r]
    ix = pkg.Index(rb)
    keys = ix.search_keys(qb)
    idx, dist = pkg.keys_unpack(keys, return_distances=True)
    torch.cuda.synchronize()
    st = ix.stats()
    assert st["path"] == 2 and st["k_tile"] == 256 and st["nonfinite"] == 0, st
    assert st["ambiguous"] < m // 20, st
    idx_h, dist_h = idx.cpu().numpy(), dist.cpu().numpy()
    pq = planted_q.cpu().numpy()
    assert (dist_h[pq] == 0).all()
    # a planted query's answer is the LOWEST ref index at distance 0 (an identical bf16 row earlier
    # in the cloud would be V0's answer too); with 256 random dims that is the planted row itself
    assert np.array_equal(idx_h[pq], planted_r.cpu().numpy().astype(np.int32))
    assert idx_h.min() >= 0 and idx_h.max() < n
    # sampled queries against the oracle over ALL refs (bf16 bits widened exactly to fp32)
    rh = rb.view(torch.int16).cpu().numpy().view(np.uint16)
    rw = (rh.astype(np.uint32) << 16).view(np.float32)
    del rh
    near = ix.near_ties()
    assert near.size == st["multi_candidate"] and near.size < m // 4, (near.size, st)
    sel = np.unique(np.concatenate([np.random.default_rng(5).choice(m, 1024, replace=False), near]))
    print(f"C5: {near.size} near-tie queries, {st['ambiguous']} exact-scan queries, checking {sel.size} queries in full")
    qsel = qb[torch.from_numpy(sel).cuda()].view(torch.int16).cpu().numpy().view(np.uint16)
    qw = (qsel.astype(np.uint32) << 16).view(np.float32)
    want_idx, want_dist = orc.v0_search(qw, rw, threads=16)
    assert np.array_equal(idx_h[sel], want_idx)
    assert np.array_equal(_bits(dist_h[sel]), _bits(want_dist))
    for i in range(0, m, 8191):
        qi = (qb[i].view(torch.int16).cpu().numpy().view(np.uint16).astype(np.uint32) << 16).view(np.float32)
        assert _bits(dist_h[i:i + 1])[0] == orc.pair_distance(qi, rw[idx_h[i]]).view(np.uint32)
    del rw
    ix.close()
    half = n // 2
    k0 = pkg.Index(rb[:half], index_base=0)
    a = k0.search_keys(qb).clone()
    torch.cuda.synchronize()
    k0.close()
    k1 = pkg.Index(rb[half:], index_base=half)
    b = k1.search_keys(qb)
    pkg.keys_min(a, b)
    torch.cuda.synchronize()
    k1.close()
    assert torch.equal(a, keys)


@pytest.mark.parametrize("shape", [(300, 5001, 128), (1000, 70000, 16), (4096, 8192, 3), (65, 999, 37), (33, 257, 100)])
def test_dimension_major_refs(pkg, orc, shape):
    """NNS_REFS_SOA: refs handed over as a dense [k][n] array (what the reference's mat_inv_kernel
    produces, core.cu:293-306) give the same indices and distance bits as the [n][k] array —
    whole call (with shards) and resident index (fp32 and bf16), including a refresh after the
    caller changed the array."""
    m, n, k = shape
    rng = np.random.default_rng(900 + k)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    want_idx, want_dist = orc.v0_search(q, r, threads=8)
    rt = np.ascontiguousarray(r.T)
    for shards in (1, 3):
        idx, dist = pkg.search(q, rt, return_distances=True, shards=shards, refs_soa=True)
        assert np.array_equal(idx, want_idx) and np.array_equal(_bits(dist), _bits(want_dist))
    qd = torch.from_numpy(q).cuda()
    rtd = torch.from_numpy(rt).cuda()
    ix = pkg.Index(rtd, soa=True, index_base=7)
    idx, dist = ix.search(qd, return_distances=True)
    assert np.array_equal(idx.cpu().numpy(), want_idx + 7) and np.array_equal(_bits(dist.cpu().numpy()), _bits(want_dist))
    # the caller rewrites its array: refresh picks the new values up
    r2 = rng.random((n, k), dtype=np.float32)
    rtd.copy_(torch.from_numpy(np.ascontiguousarray(r2.T)))
    ix.refresh()
    idx2 = ix.search(qd)
    assert np.array_equal(idx2.cpu().numpy(), orc.v0_search(q, r2, threads=8)[0] + 7)
    ix.close()
    if k >= 32:
        qb, rb = orc.round_bf16(q), orc.round_bf16(r)
        wb = orc.v0_search(qb, rb, threads=8)[0]
        ixb = pkg.Index(torch.from_numpy(np.ascontiguousarray(rb.T)).cuda().to(torch.bfloat16), soa=True)
        got = ixb.search(torch.from_numpy(qb).cuda().to(torch.bfloat16))
        assert np.array_equal(got.cpu().numpy(), wb)
        ixb.close()


@pytest.mark.parametrize("shape", [(300, 5000, 256), (700, 20001, 200), (33, 4097, 129), (1100, 33000, 192)])
def test_filter_k256_tile_shapes(pkg, orc, shape):
    """128 < k <= 256 (fp32) runs the 256-deep MFMA tile: one query block per wave, one 32-ref
    block per ring slot.  Ragged m / n / k vs the oracle, forced and under AUTO, whole and sharded;
    near-duplicates force the exact re-rank of the candidate lists."""
    m, n, k = shape
    rng = np.random.default_rng(700 + k)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    r[n // 2:n // 2 + 50] = r[:50]                       # exact duplicates: lowest index must win
    q[:20] = r[:20] + rng.normal(0, 1e-4, (20, k)).astype(np.float32)
    _check(pkg, orc, q, r, paths=("mfma", "auto"), shards=(1, 3))
    ix = pkg.Index(torch.from_numpy(r).cuda())
    ix.search(torch.from_numpy(q).cuda())
    st = ix.stats()
    if m >= 64:      # (a handful of queries stays on the exact lane-per-ref kernel under AUTO)
        assert st["path"] == 2 and st["k_tile"] == 256, st
    ix.close()


@pytest.mark.timeout(600)
def test_bench_multirank_rehearsal():
    """bench.py's N > 1 code path (rank -> ref shard, index_base, min all-reduce of the keys,
    rank-0 report) with two ranks on this box's one GPU and the exchange through gloo: the merged
    indices must equal one unsharded search over all refs.  (The RCCL run itself needs one GPU per
    rank: the driver's 8-GPU bench.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c3s", "--rehearse-one-gpu", "--verify"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["n"] == 2 * d["config"]["refs_per_gpu"]
    assert d["verified_vs_unsharded"] is True
    assert d["scaling"] == "weak" and "cpu_baseline" not in d


@pytest.mark.parametrize("shape", [(300, 5000, 128), (700, 20001, 200), (1000, 70000, 64), (130, 3000, 33),
                                   (2049, 777, 256), (65, 40000, 100)])
def test_bf16_filter_on_fp32_points(pkg, orc, shape):
    """NNS_FILTER_BF16 (opt-in): fp32 points, filter on the centred points rounded to bf16 with the
    margin widened by the rounding bound, exact fp32 re-rank.  Indices and distance bits must equal
    V0's (and therefore the fp32-filter path's) on benign and on adversarial data: exact duplicates,
    near-duplicates below bf16 resolution (they are indistinguishable to the filter and must all be
    re-ranked), offset clouds (centring), NaN / INF refs."""
    m, n, k = shape
    rng = np.random.default_rng(1100 + k)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    r[n // 2:n // 2 + 40] = r[:40]                                   # exact duplicates
    twins = r[100:140].copy()
    twins[:, 1] += np.float32(2e-4)                                  # differ below bf16 resolution
    r[n // 3:n // 3 + 40] = twins
    q[:30] = r[100:130] + rng.normal(0, 1e-4, (30, k)).astype(np.float32)
    want_idx, want_dist = orc.v0_search(q, r, threads=8)
    for shards in (1, 3):
        idx, dist = pkg.search(q, r, return_distances=True, shards=shards, path="mfma", filter_bf16=True)
        assert np.array_equal(idx, want_idx), f"shards={shards}"
        assert np.array_equal(_bits(dist), _bits(want_dist))
    # offset cloud + specials through the resident index
    q2, r2 = q + np.float32(500.0), r + np.float32(500.0)
    r2[7, 3] = np.nan
    r2[n - 1, 0] = np.inf
    with np.errstate(all="ignore"):
        w2 = orc.v0_search(q2, r2, threads=8)
    ix = pkg.Index(torch.from_numpy(r2).cuda(), filter_bf16=True, path="mfma")
    idx, dist = ix.search(torch.from_numpy(q2).cuda(), return_distances=True)
    assert np.array_equal(idx.cpu().numpy(), w2[0]) and np.array_equal(_bits(dist.cpu().numpy()), _bits(w2[1]))
    ix.close()


def test_bf16_filter_flag_is_rejected_for_bf16_points(pkg):
    q = torch.zeros((64, 64), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(pkg.NNSError):
        pkg.Index(q, filter_bf16=True)


@pytest.mark.parametrize("shape", [(300, 5000, 64), (700, 20001, 33), (2049, 777, 50), (1000, 70000, 64)])
def test_filter_k64_tile_shapes(pkg, orc, shape):
    """32 < k <= 64 runs the 64-deep fp32 MFMA tile (4 image blocks = 128 refs per ring slot)."""
    m, n, k = shape
    rng = np.random.default_rng(640 + k + m)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    r[n // 2:n // 2 + 30] = r[:30]
    _check(pkg, orc, q, r, paths=("mfma", "auto"), shards=(1, 3))
    ix = pkg.Index(torch.from_numpy(r).cuda())
    ix.search(torch.from_numpy(q).cuda())
    st = ix.stats()
    assert st["path"] == 2 and st["k_tile"] == 64, st
    ix.close()


@pytest.mark.parametrize("shape", [(300, 5000, 512), (700, 20001, 300), (130, 3000, 400), (2049, 777, 257), (513, 33000, 384),
                                   (300, 5000, 385)])
def test_bf16_k512_tile_shapes(pkg, orc, shape):
    """bf16 points with 256 < k <= 512: the 512-deep tile (16x16x32 MFMAs, four waves) and, up to k = 384, the
    384-deep one (24 fragment steps per block: four blocks over three ring slots, eight waves), forced and under AUTO,
    whole and sharded."""
    m, n, k = shape
    rng = np.random.default_rng(5120 + k)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    r[n // 2:n // 2 + 30] = r[:30]
    _check_bf16(pkg, orc, q, r, paths=("mfma", "auto"), shards=(1, 3))
    ix = pkg.Index(torch.from_numpy(orc.round_bf16(r)).cuda().to(torch.bfloat16))
    ix.search(torch.from_numpy(orc.round_bf16(q)).cuda().to(torch.bfloat16))
    st = ix.stats()
    assert st["path"] == 2 and st["k_tile"] == (384 if k <= 384 else 512), st   # (256 < k <= 384: the 384-deep tile, round 3)
    ix.close()


@pytest.mark.parametrize("shape", [(300, 5000, 512), (700, 20001, 300), (130, 3000, 400), (1100, 9000, 257), (513, 33000, 384), (300, 5000, 385)])
def test_fp32_points_k512_bf16_operand_tile(pkg, orc, shape):
    """fp32 points with 256 < k <= 512: no fp32 tile is that deep, so AUTO runs the bf16-operand filter
    (512-deep tile, rounding-widened margin) + the exact fp32 re-rank.  V0's bits, whole and sharded,
    also when forced with the flag; near-duplicates below bf16 resolution must all be re-ranked."""
    m, n, k = shape
    rng = np.random.default_rng(7000 + k)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    r[n // 2:n // 2 + 30] = r[:30]
    tw = r[60:90].copy()
    tw[:, 5] += np.float32(3e-4)
    r[n // 3:n // 3 + 30] = tw
    q[:25] = r[60:85] + rng.normal(0, 1e-4, (25, k)).astype(np.float32)
    want_idx, want_dist = orc.v0_search(q, r, threads=8)
    for kw in ({"path": "auto"}, {"path": "mfma", "filter_bf16": True}):
        for shards in (1, 3):
            idx, dist = pkg.search(q, r, return_distances=True, shards=shards, **kw)
            assert np.array_equal(idx, want_idx), (kw, shards)
            assert np.array_equal(_bits(dist), _bits(want_dist))
    ix = pkg.Index(torch.from_numpy(r).cuda())
    ix.search(torch.from_numpy(q).cuda())
    st = ix.stats()
    assert st["path"] == 2 and st["k_tile"] == (384 if k <= 384 else 512), st   # (256 < k <= 384: the 384-deep tile, round 3)
    ix.close()


def test_whole_call_entry_points_are_reentrant(pkg, orc):
    """the whole-call entry points are re-entrant.  Four host threads search
    different shapes concurrently (ctypes drops the GIL; the workspace pool is shared and locked):
    every result must be V0's."""
    import threading
    rng = np.random.default_rng(4242)
    jobs = []
    for (m, n, k) in [(300, 9000, 128), (1000, 5000, 16), (200, 30000, 3), (129, 7000, 200)]:
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        jobs.append((q, r, orc.v0_search(q, r, threads=4)[0]))
    errors = []

    def worker(q, r, want):
        try:
            for _ in range(12):
                if not np.array_equal(pkg.search(q, r), want):
                    errors.append("mismatch")
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_pool_can_be_disabled():
    """NNS_POOL_BYTES=0: nothing is parked (every workspace goes back to the runtime on return, like
    the reference's per-call cudaFree), results unchanged."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g\n"
        "pkg, orc = g.load_package(), g.load_oracle()\n"
        "rng = np.random.default_rng(1); q = rng.random((200, 64), dtype=np.float32); r = rng.random((5000, 64), dtype=np.float32)\n"
        "want = orc.v0_search(q, r)[0]\n"
        "for _ in range(3): assert np.array_equal(pkg.search(q, r), want)\n"
        "assert pkg.trim() == 0\n"
        "print('ok')\n" % root)
    env = dict(os.environ, NNS_POOL_BYTES="0")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]


@pytest.mark.timeout(600)
def test_short_randomised_sweep():
    """A 20-second slice of tools/fuzz_parity.py (random shapes, data families, dtypes, paths, shards,
    opt-in bf16 filter) with a fixed seed: every case must match the oracle bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--seconds", "20", "--seed", "11"],
                         cwd=root, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("fuzz_parity:") and " 0 failures" in last, last
    assert int(last.split()[1]) > 200      # it did run a few hundred cases


def test_library_comm_single_rank(pkg):
    """nns_comm_* (the exchange of the one-process-per-GPU form) with ONE rank: unique id, collective
    create, the library-issued ncclAllReduce(uint64, min) on the caller's stream, RCCL's own rank count.
    (More ranks need more GPUs: the driver's multi-GPU bench; bench.py cross-checks the library exchange
    against torch.distributed's all_reduce there.)"""
    uid = pkg.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = pkg.Comm(uid, 1, 0, 0)
    assert comm.size() == 1
    keys = torch.randint(0, 2 ** 62, (65536,), dtype=torch.int64, device="cuda")
    keys[5] = pkg.NNS_KEY_NONE
    before = keys.clone()
    pkg.allreduce_min_keys(keys, comm=comm)
    torch.cuda.synchronize()
    assert torch.equal(keys, before)
    comm.close()
    with pytest.raises(pkg.NNSError):
        pkg.Comm(uid[:64], 1, 0, 0)            # a truncated id is rejected, not passed to RCCL
    with pytest.raises(pkg.NNSError):
        pkg.Comm(uid, 2, 5, 0)                 # rank out of range


def test_in_library_multi_bf16_and_soa(pkg, orc):
    """nns_search_bf16_multi and NNS_REFS_SOA through nns_search_f32_multi (a shard of a dimension-major
    array is a column range: strided upload), rehearsed with virtual shards on the one GPU."""
    rng = np.random.default_rng(56)
    m, n, k = 300, 40000, 64
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    r[n - 1] = r[0]
    q[0] = r[0]
    want_idx, want_dist = orc.v0_search(q, r, threads=8)
    rt = np.ascontiguousarray(r.T)
    for kwargs in ({"num_devices": 0}, {"num_devices": 3, "virtual": True}, {"num_devices": 7, "virtual": True}):
        idx, dist = pkg.search_multi(q, rt, return_distances=True, refs_soa=True, **kwargs)
        assert np.array_equal(idx, want_idx), kwargs
        assert np.array_equal(_bits(dist), _bits(want_dist)), kwargs
    qb, rb = orc.round_bf16(q), orc.round_bf16(r)
    wb_idx, wb_dist = orc.v0_search(qb, rb, threads=8)
    for kwargs in ({"num_devices": 0}, {"num_devices": 4, "virtual": True}):
        idx, dist = pkg.search_multi(pkg.to_bf16_bits(qb), pkg.to_bf16_bits(rb), return_distances=True, bf16=True, **kwargs)
        assert np.array_equal(idx, wb_idx), kwargs
        assert np.array_equal(_bits(dist), _bits(wb_dist)), kwargs
    pkg.shutdown()                                  # cached communicators (none on one GPU) + pool
    assert np.array_equal(pkg.search_multi(q, r, num_devices=2, virtual=True), want_idx)   # still usable


def test_multi_entry_is_reentrant(pkg, orc):
    """nns.h: the whole-call entry points are re-entrant — including nns_search_f32_multi, whose RCCL
    loader used to be guarded by a plain bool.  Several host threads call it at once (virtual shards)."""
    import threading
    rng = np.random.default_rng(4243)
    jobs = []
    for (m, n, k) in [(300, 9000, 128), (500, 20000, 16), (129, 7000, 3)]:
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        jobs.append((q, r, orc.v0_search(q, r, threads=4)[0]))
    errors = []

    def worker(q, r, want):
        try:
            for _ in range(6):
                if not np.array_equal(pkg.search_multi(q, r, num_devices=3, virtual=True), want):
                    errors.append("mismatch")
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_refresh_relatches_nonfinite_refs(pkg, orc):
    """An index built on refs containing a NaN runs the exact kernels; after the caller overwrites the
    refs with clean data and refreshes, the MFMA path must come back (and the other way round: clean ->
    NaN must not produce wrong answers while the flag is stale)."""
    rng = np.random.default_rng(63)
    m, n, k = 200, 6000, 64
    q = rng.random((m, k), dtype=np.float32)
    r_bad = rng.random((n, k), dtype=np.float32)
    r_bad[17, 3] = np.nan
    r_good = rng.random((n, k), dtype=np.float32)
    qd = torch.from_numpy(q).cuda()
    rd = torch.from_numpy(r_bad).cuda()
    ix = pkg.Index(rd)
    with np.errstate(all="ignore"):
        want_bad = orc.v0_search(q, r_bad, threads=8)[0]
    assert np.array_equal(ix.search(qd).cpu().numpy(), want_bad)
    st = ix.stats()
    assert st["path"] == 1 and st["nonfinite"] == 1, st
    rd.copy_(torch.from_numpy(r_good))
    ix.refresh()
    assert np.array_equal(ix.search(qd).cpu().numpy(), orc.v0_search(q, r_good, threads=8)[0])
    st = ix.stats()
    assert st["path"] == 2 and st["nonfinite"] == 0, st
    rd.copy_(torch.from_numpy(r_bad))               # clean -> NaN: the flag is stale until stats()
    ix.refresh()
    assert np.array_equal(ix.search(qd).cpu().numpy(), want_bad)
    st = ix.stats()
    assert st["nonfinite"] == 1 and st["ambiguous"] == m, st       # the device-side check sent every query to the scan
    assert np.array_equal(ix.search(qd).cpu().numpy(), want_bad)
    assert ix.stats()["path"] == 1                  # re-latched by stats(): straight to the exact kernels now
    ix.close()


@pytest.mark.timeout(600)
def test_bench_self_launch_rehearsal():
    """`python bench.py --gpus 2` exactly as the driver starts it (no launcher): bench.py brings up its
    own two ranks.  On this one-GPU box they share cuda:0 and exchange through gloo
    (--rehearse-one-gpu); the merged indices must equal the unsharded search."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--workload", "c3s", "--rehearse-one-gpu", "--verify"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["verified_vs_unsharded"] is True
    assert d["exchange"]["matches_torch_all_reduce"] is True and d["exchange"]["queries_won_sum_over_ranks"] >= d["config"]["m"]
    # round 3: the line says where a step's time went on every rank, and every cross-check is enforced
    rk = d["ranks"]
    for f in ("step_ms", "search_ms", "exchange_ms"):
        assert len(rk[f]) == 2 and all(v > 0 for v in rk[f]), rk
    assert rk["step_ms_max"] == max(rk["step_ms"]) and rk["slowest_rank"] in (0, 1)
    assert max(rk["step_ms"]) <= d["ms_per_step"] * 1.05          # a rank's own time fits the max-over-ranks step
    assert all(s_ + e_ <= t * 1.05 for s_, e_, t in zip(rk["search_ms"], rk["exchange_ms"], rk["step_ms"]))
    assert d["exchange"]["exchange_ms"]["payload_bytes"] == 8 * d["config"]["m"] and d["parity_ok"] is True


def test_search_indices_fused_unpack_and_k1a_rearm(pkg, orc):
    """nns_index_search_indices: keys + unpacked indices / distances from one call on every path, and K1a's
    in-kernel second stage across repeated launches on the same index (its arrival counters must re-arm),
    across different query counts (the workspace is laid out again) and ragged ref counts."""
    rng = np.random.default_rng(97)
    for (m, n, k) in [(4096, 65536, 3), (700, 30001, 3), (64, 100003, 2), (1000, 5000, 16), (300, 7000, 128), (100, 3000, 7)]:
        r = rng.random((n, k), dtype=np.float32)
        r[n - 1] = r[1]                                   # a late duplicate: the lower index must win
        rd = torch.from_numpy(r).cuda()
        ix = pkg.Index(rd)
        for mm in (m, max(64, m // 3), m):               # same index, three query counts
            q = rng.random((mm, k), dtype=np.float32)
            q[0] = r[1]
            qd = torch.from_numpy(q).cuda()
            want_idx, want_dist = orc.v0_search(q, r, threads=8)
            for rep in range(3):                          # repeated launches: counters re-armed by the kernel
                keys = torch.empty(mm, dtype=torch.int64, device="cuda")
                dist = torch.empty(mm, dtype=torch.float32, device="cuda")
                idx = ix.search_indices(qd, keys=keys, dist=dist)
                torch.cuda.synchronize()
                assert np.array_equal(idx.cpu().numpy(), want_idx), (m, n, k, mm, rep)
                assert np.array_equal(_bits(dist.cpu().numpy()), _bits(want_dist))
                kk = keys.cpu().numpy()
                assert np.array_equal((kk & 0xFFFFFFFF).astype(np.int32), want_idx)
        ix.close()
    # bf16 index through the same entry point
    q = orc.round_bf16(rng.random((200, 64), dtype=np.float32))
    r = orc.round_bf16(rng.random((5000, 64), dtype=np.float32))
    ix = pkg.Index(torch.from_numpy(r).cuda().to(torch.bfloat16))
    idx = ix.search_indices(torch.from_numpy(q).cuda().to(torch.bfloat16))
    assert np.array_equal(idx.cpu().numpy(), orc.v0_search(q, r, threads=8)[0])
    ix.close()


@pytest.mark.timeout(600)
def test_whole_call_pipelined_upload(pkg, orc):
    """Whole calls worth it upload their refs in four chunks (1/8, 1/4, 5/16, 5/16) and search each chunk on a
    non-blocking stream while the next one is copied (merged with the packed-key min): the answer must be the
    unsharded V0 answer bit for bit — with exact cross-chunk ties, and with a NaN in a late chunk (found by K5's
    device-side check, since the chunk indexes are built without the synchronising read-back)."""
    m, n, k = 8192, 262144 + 777, 128
    q = orc.rng_uniform(m * k, 91, 0).reshape(m, k)
    r = orc.rng_uniform(n * k, 91, m * k).reshape(n, k)
    r[n - 5] = r[3]                      # cross-chunk exact tie: the lowest index (first chunk) must win
    q[0] = r[3]
    q[1] = r[n - 9]                      # exact hit in the last chunk
    want_idx, want_dist = orc.v0_search(q, r, threads=16)
    for rep in range(2):                 # second call: pooled workspaces
        idx, dist = pkg.search(q, r, return_distances=True)
        assert np.array_equal(idx, want_idx), rep
        assert np.array_equal(_bits(dist), _bits(want_dist))
    assert np.array_equal(pkg.cudaCall(k, m, n, q, r), want_idx)
    assert np.array_equal(pkg.search(q, r, shards=3), want_idx)     # the plain (synchronous) path agrees
    # the in-library multi-GPU entry takes the same overlapped upload per shard (two virtual shards of 64 MiB,
    # two host threads on this one device; the cross-chunk tie also crosses the shard boundary)
    mi, md = pkg.search_multi(q, r, num_devices=2, virtual=True, return_distances=True)
    assert np.array_equal(mi, want_idx) and np.array_equal(_bits(md), _bits(want_dist))
    r2 = r.copy()
    r2[200000, 7] = np.nan
    with np.errstate(all="ignore"):
        w2 = orc.v0_search(q, r2, threads=16)[0]
    assert np.array_equal(pkg.search(q, r2), w2)
    pkg.shutdown()                       # releases communicators / scratch; the next call rebuilds what it needs
    assert np.array_equal(pkg.search(q, r), want_idx)


@pytest.mark.parametrize("shape", [(300, 5000, 1024), (700, 20001, 600), (130, 3000, 800), (1100, 9000, 513),
                                   (300, 5000, 768), (2049, 777, 700), (64, 33000, 769), (300, 5000, 640), (1100, 9000, 641),
                                   (513, 40000, 550)])
def test_k1024_tile_shapes(pkg, orc, shape):
    """512 < k <= 1024: K-split accumulation on the MFMA path — the 1024-deep bf16 tile (one query block per
    wave on one wave per SIMD, a 32-ref block spanning TWO ring slots with its accumulators carried across
    the slot barrier) and, up to k = 768, the 768-deep one (48 fragment steps per block: two blocks over THREE ring
    slots, the second starting in the middle of a slot; two waves per SIMD) and, up to k = 640, the 640-deep one (40 steps per
    block: four blocks over FIVE slots).  bf16 points (forced and AUTO) and fp32 points (AUTO takes the bf16-operand filter with
    the rounding-widened margin + exact fp32 re-rank): V0's bits, whole and sharded, exact duplicates and
    near-duplicates below bf16 resolution included; and faster than the exact VALU scan."""
    m, n, k = shape
    rng = np.random.default_rng(10240 + k)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    r[n // 2:n // 2 + 30] = r[:30]                                   # exact duplicates: lowest index
    tw = r[60:90].copy()
    tw[:, 5] += np.float32(3e-4)                                     # below bf16 resolution
    r[n // 3:n // 3 + 30] = tw
    q[:25] = r[60:85] + rng.normal(0, 1e-4, (25, k)).astype(np.float32)
    _check_bf16(pkg, orc, q, r, paths=("mfma", "auto"), shards=(1, 3))
    want_idx, want_dist = orc.v0_search(q, r, threads=8)
    for kw in ({"path": "auto"}, {"path": "mfma", "filter_bf16": True}):
        for shards in (1, 3):
            idx, dist = pkg.search(q, r, return_distances=True, shards=shards, **kw)
            assert np.array_equal(idx, want_idx), (kw, shards)
            assert np.array_equal(_bits(dist), _bits(want_dist))
    for bf in (False, True):
        rr = torch.from_numpy(orc.round_bf16(r) if bf else r).cuda()
        qq = torch.from_numpy(orc.round_bf16(q) if bf else q).cuda()
        if bf:
            rr, qq = rr.to(torch.bfloat16), qq.to(torch.bfloat16)
        ix = pkg.Index(rr, profile=True)
        ix.search(qq)
        st = ix.stats()
        assert st["path"] == 2 and st["k_tile"] == (640 if k <= 640 else 768 if k <= 768 else 1024), st
        ix.close()
    with pytest.raises(pkg.NNSError):
        pkg.search(np.zeros((4, 1025), np.float32), np.zeros((9, 1025), np.float32), path="mfma", filter_bf16=True)


def test_lane_threshold_sharing_pairs_the_right_lanes(pkg):
    """The slow path lets the lanes that carry one query adopt the smallest of their thresholds (short ref
    streams), through v_permlane32_swap / v_permlane16_swap.  A wrong pairing would hand a query another
    query's threshold (silently dropping candidates): check the instructions' lane pairing on the hardware."""
    rng = np.random.default_rng(606)
    for _ in range(5):
        v = rng.normal(0, 100, 64).astype(np.float32)
        lanes = np.arange(64)
        want32 = np.minimum(v, v[lanes ^ 32])
        want16 = np.minimum(np.minimum(v, v[lanes ^ 16]), np.minimum(v[lanes ^ 32], v[lanes ^ 48]))
        assert np.array_equal(pkg.selftest_lane_share(v, False), want32)
        assert np.array_equal(pkg.selftest_lane_share(v, True), want16)


@pytest.mark.timeout(600)
def test_bench_rccl_path_world_size_one():
    """bench.py as ONE rank under torch.distributed.run with the real `nccl` (= RCCL) backend: process group,
    the 128-byte id carried by torch.distributed, the library's own communicator (nns_comm_create), the
    library-issued ncclAllReduce(uint64, min) inside the timed step, its bit-for-bit cross-check against
    torch.distributed.all_reduce, teardown, and the JSON line last on stdout (RCCL prints a banner).  The
    closest rehearsal of the N > 1 exchange a one-GPU box allows."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "c3s", "--verify", "--no-cpu-baseline",
           "--exchange", "library"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    d = json.loads(last)                                  # the LAST line is the JSON line
    assert d["n_gpus"] == 1 and d["verified_vs_unsharded"] is True
    assert d["exchange"]["impl"].startswith("nns_comm_allreduce_min") and d["exchange"]["rccl_ranks"] == 1
    assert d["exchange"]["matches_torch_all_reduce"] is True


def test_k1a_first_minimum_across_chunks_waves_and_splits(pkg, orc):
    """K1a keeps (distance, chunk start) through its merges and recovers the exact index once per query: exact
    duplicates of the winner straddling every boundary it knows — chunk (8 refs), LDS tile (1280 refs at k = 3),
    wave round-robin, ref split — must still resolve to the LOWEST index; all-identical refs give index 0."""
    rng = np.random.default_rng(808)
    for k, n in ((3, 65536), (2, 40000), (16, 30000), (8, 5000)):
        m = 700
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32) + np.float32(2.0)      # far from the queries
        hot = rng.random((1, k), dtype=np.float32)
        # the same nearest point at positions on both sides of every kind of boundary
        for pos in (7, 8, 1279, 1280, 1359, 1360, 2559, 2560, 4095, 4096, 4097, n // 2 - 1, n // 2, n - 1):
            r[pos] = hot
        q[:350] = hot + rng.normal(0, 1e-3, (350, k)).astype(np.float32)
        want_idx, want_dist = orc.v0_search(q, r, threads=8)
        assert (want_idx[:350] == 7).all()
        for shards in (1, 3):
            idx, dist = pkg.search(q, r, return_distances=True, shards=shards, path="exact")
            assert np.array_equal(idx, want_idx), (k, n, shards)
            assert np.array_equal(_bits(dist), _bits(want_dist))
        r2 = r.copy()
        r2[7] = r2[9]                                                  # now the first copy sits at index 8
        assert (pkg.search(q[:350], r2, path="exact") == 8).all()
    same = np.tile(rng.random((1, 3), dtype=np.float32), (100000, 1))
    assert (pkg.search(rng.random((300, 3), dtype=np.float32), same) == 0).all()


def test_k1a_lds_dma_staging_alignments_and_ragged_tiles(pkg, orc):
    """K1a (k <= 4) stages its ref tiles and its queries by LDS-DMA: 16-byte pieces from a 16-byte-aligned source,
    dword pieces otherwise and for the last partial piece, NaN / zero padding written by the lanes themselves.  Ref
    and query arrays at every 4-byte misalignment, ragged ref counts (last tile / last chunk partial, one ref short
    and one past a tile), ragged query counts — against the oracle, indices and distance bits."""
    rng = np.random.default_rng(909)
    dev = torch.device("cuda", 0)
    for k in (1, 2, 3, 4):
        for off_r, off_q, m, n in ((0, 0, 300, 5121), (1, 0, 129, 1280 * 3 + 7), (2, 1, 128, 1279), (3, 2, 5, 1281),
                                   (1, 3, 1000, 70001), (0, 1, 64, 9), (5, 7, 257, 2560)):
            rbig = rng.random((n + 8, k), dtype=np.float32)
            qbig = rng.random((m + 8, k), dtype=np.float32)
            r_d = torch.from_numpy(rbig).to(dev)[off_r:off_r + n]      # contiguous views at 4 k off_r bytes
            q_d = torch.from_numpy(qbig).to(dev)[off_q:off_q + m]
            assert r_d.is_contiguous() and q_d.is_contiguous()
            want_idx, want_dist = orc.v0_search(qbig[off_q:off_q + m], rbig[off_r:off_r + n], threads=8)
            ix = pkg.Index(r_d, path="exact")
            idx, dist = ix.search(q_d, return_distances=True)
            torch.cuda.synchronize()
            assert np.array_equal(idx.cpu().numpy(), want_idx), (k, off_r, off_q, m, n)
            assert np.array_equal(_bits(dist.cpu().numpy()), _bits(want_dist)), (k, off_r, off_q, m, n)
            ix.close()


def test_whole_call_chunked_upload_exact_path(pkg, orc):
    """The chunked, overlapped upload also serves exact-path whole calls worth >= 0.5 ms (3-D, 4096 x 700001: 8.4 MB of
    refs in four chunks through K1a, merged with the packed-key min): cross-chunk exact ties resolve to the lowest
    index, a ragged last chunk, distances bit-equal; and a 16-D problem just above the small-problem rule (MFMA path,
    16-deep tile, chunked)."""
    m, n, k = 4096, 700001, 3
    q = orc.rng_uniform(m * k, 17, 0).reshape(m, k)
    r = orc.rng_uniform(n * k, 17, m * k).reshape(n, k)
    r[n - 2] = r[11]                     # the same point in the first and in the last chunk
    q[5] = r[11]
    q[6] = r[n - 1]
    want_idx, want_dist = orc.v0_search(q, r, threads=16)
    assert want_idx[5] == 11 and want_idx[6] == n - 1
    idx, dist = pkg.search(q, r, return_distances=True)
    assert np.array_equal(idx, want_idx)
    assert np.array_equal(_bits(dist), _bits(want_dist))
    m, n, k = 4096, 40000, 16            # 1.6e8 pairs > 2^26: MFMA path; est. search < 0.5 ms -> plain upload
    q = orc.rng_uniform(m * k, 18, 0).reshape(m, k)
    r = orc.rng_uniform(n * k, 18, m * k).reshape(n, k)
    assert np.array_equal(pkg.search(q, r), orc.v0_search(q, r, threads=16)[0])
    m, n, k = 4096, 600011, 16           # 38 MB of refs, est. search 0.56 ms: chunked, 16-deep tile
    q = orc.rng_uniform(m * k, 19, 0).reshape(m, k)
    r = orc.rng_uniform(n * k, 19, m * k).reshape(n, k)
    assert np.array_equal(pkg.search(q, r), orc.v0_search(q, r, threads=16)[0])


@pytest.mark.parametrize("shape", [(40, 700, 1025), (65, 500, 2048), (3, 300, 4099), (130, 257, 3000), (2, 100, 16384)])
def test_dimensionality_beyond_the_filter_and_its_limit(pkg, orc, shape):
    """k > 1024 has no MFMA tile: the exact lane-per-ref kernel runs it with the query tile in LDS (k <= 16384: 64 KiB),
    fp32 and bf16, whole call and sharded; k = 16385 is refused with a status code, not a crash."""
    m, n, k = shape
    rng = np.random.default_rng(k)
    q = (rng.random((m, k), dtype=np.float32) - 0.5) * 3
    r = (rng.random((n, k), dtype=np.float32) - 0.5) * 3
    r[n // 2] = q[0]                                  # an exact hit
    _check(pkg, orc, q, r, paths=("auto", "exact"), shards=(1, 3))
    _check_bf16(pkg, orc, q, r, paths=("auto",))
    if k == 16384:
        with pytest.raises(pkg.NNSError):
            pkg.search(np.zeros((2, k + 1), np.float32), np.zeros((5, k + 1), np.float32))


def test_split_api_on_side_streams(pkg, orc):
    """Everything the split API enqueues goes to the CALLER's stream (include/nns.h): two indexes searched from two
    non-blocking side streams at once, results consumed on those streams — exact kernel (merge accumulator armed and
    re-armed on the stream), MFMA filter (prep, filter, finalize, re-rank), refresh + fused index unpack.  A launch
    that strayed onto the default stream would race with the side streams' work and show up as wrong answers."""
    rng = np.random.default_rng(2024)
    dev = torch.device("cuda", 0)
    cases = []
    for k, m, n in ((3, 3000, 120000), (128, 1500, 60000), (16, 2000, 90000), (256, 700, 30000)):
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        cases.append((torch.from_numpy(q).to(dev), torch.from_numpy(r).to(dev), orc.v0_search(q, r, threads=16)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in cases]
    idxs = [pkg.Index(r_d) for (_q, r_d, _w) in cases]            # built on the default stream, synchronous
    torch.cuda.synchronize()
    outs = [None] * len(cases)
    for rep in range(3):
        for i, (q_d, r_d, _want) in enumerate(cases):             # enqueue everything first, then check
            with torch.cuda.stream(streams[i]):
                idxs[i].refresh()
                keys = torch.empty(q_d.shape[0], dtype=torch.int64, device=dev)
                idx = torch.empty(q_d.shape[0], dtype=torch.int32, device=dev)
                dist = torch.empty(q_d.shape[0], dtype=torch.float32, device=dev)
                idxs[i].search_indices(q_d, keys, idx, dist)
                outs[i] = (idx, dist, idxs[i].search(q_d))        # the unfused form on the same stream
        for i, (_q, _r, (want_idx, want_dist)) in enumerate(cases):
            streams[i].synchronize()
            idx, dist, idx2 = outs[i]
            assert np.array_equal(idx.cpu().numpy(), want_idx), (rep, i)
            assert np.array_equal(_bits(dist.cpu().numpy()), _bits(want_dist)), (rep, i)
            assert np.array_equal(idx2.cpu().numpy(), want_idx), (rep, i)
    for ix in idxs:
        ix.close()


def test_index_base_up_to_the_int32_limit(pkg, orc):
    """A shard whose global indices end at 2^31 - 1 (index_base + n = 2^31 - 1 is the last legal placement; one more is
    refused): exact and MFMA paths return base + local index as positive int32, keys order by (distance, global index)."""
    rng = np.random.default_rng(31)
    dev = torch.device("cuda", 0)
    for k, n, m in ((3, 5000, 300), (128, 3000, 200), (16, 70000, 1100)):
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        r[n - 1] = r[4]                                        # a tie inside the shard: the lower global index wins
        q[0] = r[4]
        base = 0x7FFFFFFF - n
        want_idx, want_dist = orc.v0_search(q, r, threads=8)
        r_d, q_d = torch.from_numpy(r).to(dev), torch.from_numpy(q).to(dev)
        for path in ("auto", "exact"):
            ix = pkg.Index(r_d, index_base=base, path=path)
            idx, dist = ix.search(q_d, return_distances=True)
            torch.cuda.synchronize()
            got = idx.cpu().numpy()
            assert got.dtype == np.int32 and (got > 0).all()
            assert np.array_equal(got.astype(np.int64) - base, want_idx), (k, path)
            assert np.array_equal(_bits(dist.cpu().numpy()), _bits(want_dist))
            ix.close()
        with pytest.raises(pkg.NNSError):
            pkg.Index(r_d, index_base=base + 1)


@pytest.mark.timeout(900)
def test_largest_ref_count_int32_boundary(pkg, orc):
    """n = NNS_MAX_POINTS (2^31 - 2^20) one-dimensional refs, 8.6 GB resident: range ends, strides and padded sizes
    of the exact kernels stay inside int32 (the last ref range's end is computed in 64 bits), the winner may sit in the
    very last refs, and with only 2^24 distinct fp32 values in [0, 1) every query has thousands of exact ties — the
    lowest index must win.  K1b (16 queries) and K1a (128 queries) against the oracle; one ref more is rejected."""
    n, k = 0x7FF00000, 1
    dev = torch.device("cuda", 0)
    r_d = torch.empty((n, k), dtype=torch.float32, device=dev)
    pkg.fill_uniform(r_d, 4242, 0)
    r_d += 2.0                                    # refs in [2, 3): the planted points below are the only near ones
    r_d[n - 3, 0] = 0.25
    r_d[n - 2, 0] = 0.25                          # exact duplicate: n - 3 must win
    r_d[7, 0] = 0.75
    r_d[n - 1, 0] = 0.75                          # duplicate of an early ref: 7 must win
    r_h = r_d.cpu().numpy()
    q = np.full((128, k), 0.25, dtype=np.float32)
    q[1::2] = 0.75
    q[5] = 2.5                                    # in the cloud: thousands of exact ties, lowest index wins
    q[6] = 2.999
    want_idx, want_dist = orc.v0_search(q[:16], r_h, threads=16)
    assert want_idx[0] == n - 3 and want_idx[1] == 7
    ix = pkg.Index(r_d, path="exact")
    for m in (16, 128):
        idx, dist = ix.search(torch.from_numpy(q[:m]).to(dev), return_distances=True)
        torch.cuda.synchronize()
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
        assert np.array_equal(idx[:16], want_idx), (m, idx[:16], want_idx)
        assert np.array_equal(_bits(dist[:16]), _bits(want_dist))
        assert (idx[16:][0::2] == n - 3).all() and (idx[16:][1::2] == 7).all()
    ix.close()
    del r_d
    torch.cuda.empty_cache()
    pkg.trim()
    # the whole-call entry on the same host array (8.6 GB pageable upload): plain path (16 queries) and the chunked,
    # overlapped upload (128 queries: four chunks of up to 6.7e8 refs, global indices carried by index_base)
    for m in (16, 128):
        idx, dist = pkg.search(q[:m], r_h, return_distances=True)
        assert np.array_equal(idx[:16], want_idx), (m, idx[:16])
        assert np.array_equal(_bits(dist[:16]), _bits(want_dist))
        assert (idx[16:][0::2] == n - 3).all() and (idx[16:][1::2] == 7).all()
    del r_h
    with pytest.raises(pkg.NNSError):             # one more point than NNS_MAX_POINTS
        big = torch.empty((n + 1, k), dtype=torch.float32, device=dev)
        try:
            pkg.Index(big, path="exact")
        finally:
            del big
            torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_largest_query_count_int32_boundary(pkg):
    """m = NNS_MAX_POINTS queries in one search (1-D, 1000 distinct refs, K1a): query j is a copy of ref j mod 1000, so
    its answer is j mod 1000 at distance 0 — checked on the device in slices (keys, indices and distances of 2^31 - 2^20
    queries are 34 GB).  One query more is rejected."""
    m, n, k = 0x7FF00000, 1000, 1
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    pkg.trim()
    free, _total = torch.cuda.mem_get_info()
    if free < 80 * 10**9:
        pytest.skip(f"needs ~60 GB of device memory ({free / 1e9:.0f} GB free)")
    r_d = (torch.arange(n, dtype=torch.float32, device=dev) * 0.37).reshape(n, k).contiguous()
    q_d = torch.empty((m, k), dtype=torch.float32, device=dev)
    step = 1 << 27
    for a in range(0, m, step):                      # q[j] = r[j mod n], built in slices
        b = min(m, a + step)
        q_d[a:b] = r_d[torch.arange(a, b, device=dev) % n]
    ix = pkg.Index(r_d, path="exact")
    idx, dist = ix.search(q_d, return_distances=True)
    torch.cuda.synchronize()
    for a in range(0, m, step):
        b = min(m, a + step)
        want = (torch.arange(a, b, device=dev) % n).to(torch.int32)
        assert bool(torch.equal(idx[a:b], want)), a
        assert bool((dist[a:b] == 0).all()), a
    del idx, dist
    with pytest.raises(pkg.NNSError):
        q2 = torch.empty((m + 1, k), dtype=torch.float32, device=dev)
        try:
            ix.search(q2)
        finally:
            del q2
    ix.close()
    del q_d, r_d
    torch.cuda.empty_cache()
    pkg.trim()


@pytest.mark.timeout(900)
def test_largest_ref_count_through_the_filter(pkg):
    """The same count (2^31 - 2^20 refs, 8-D: 68.7 GB of points + a 137 GB tile image resident — what 288 GB of HBM are
    for) through the MFMA filter (16-deep tile): every query is a copy of a planted ref — first, last, on slot and split
    edges, random — so its answer is that ref's index at distance 0 (no CPU oracle can hold this; duplicates of an 8-D
    point among 2^31 have probability ~0).  Slot, image and list indices stay inside int32 / use 64-bit addressing."""
    n, k = 0x7FF00000, 8
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    pkg.trim()
    free, _total = torch.cuda.mem_get_info()
    if free < 235 * 10**9:
        pytest.skip(f"needs ~220 GB of device memory ({free / 1e9:.0f} GB free)")
    r_d = torch.empty((n, k), dtype=torch.float32, device=dev)
    pkg.fill_uniform(r_d, 77, 0)
    rng = np.random.default_rng(77)
    pos = np.unique(np.concatenate([[0, 1, 31, 32, 511, 512, n // 2 - 1, n // 2, n - 513, n - 512, n - 33, n - 32, n - 2, n - 1],
                                    rng.integers(0, n, 50)])).astype(np.int64)
    q_d = r_d[torch.from_numpy(pos).to(dev)].contiguous()
    ix = pkg.Index(r_d)
    idx, dist = ix.search(q_d, return_distances=True)
    torch.cuda.synchronize()
    st = ix.stats()
    assert st["path"] == 2 and st["k_tile"] == 16, st
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), pos)
    assert (dist.cpu().numpy() == 0.0).all()
    ix.close()
    del r_d, q_d
    torch.cuda.empty_cache()
    pkg.trim()


def test_small_whole_call_scratch_path_and_its_size_boundary(pkg, orc):
    """Whole calls whose inputs + outputs fit the 2 MiB pinned scratch take one upload / one wait (search_host_small);
    just above it the plain path runs.  Both sides of the boundary, fp32 and bf16, distances, NaN refs (the small
    path builds its index without the synchronising read-back: K5's device-side check must catch them), and two
    threads at once (the second finds the scratch busy and takes the plain path)."""
    import threading
    rng = np.random.default_rng(1234)
    k = 16
    for m, n in ((64, 32000), (64, 32500), (64, 33500), (1, 1024), (300, 20000)):   # 2 MiB = 32768 rows of 64 B
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        _check(pkg, orc, q, r, paths=("auto", "exact", "mfma"))
    q = rng.random((200, 128), dtype=np.float32)
    r = rng.random((3000, 128), dtype=np.float32)
    r[5, 3] = np.nan
    r[77, 0] = np.inf
    r[100] = 3e18
    _check(pkg, orc, q, r, paths=("auto", "mfma"))
    qb = (rng.random((300, 256), dtype=np.float32)).astype(np.float32)
    rb = (rng.random((1500, 256), dtype=np.float32)).astype(np.float32)
    _check_bf16(pkg, orc, qb, rb)
    want = orc.v0_search(q, r, threads=4)[0]
    out = [None, None]

    def work(i):
        for _ in range(20):
            out[i] = pkg.search(q, r)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert np.array_equal(out[0], want) and np.array_equal(out[1], want)


def test_deep_tile_specials_and_whole_call_bf16_pipeline(pkg, orc):
    """The 1024-deep tile with NaN / INF refs and magnitudes that void the error bound (exact kernels must take
    over), and a bf16 whole call large enough for the pipelined upload."""
    rng = np.random.default_rng(909)
    m, n, k = 130, 4000, 700
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    r[17, 600] = np.nan
    r[3000, 5] = np.inf
    q[4, 650] = np.nan
    with np.errstate(all="ignore"):
        _check(pkg, orc, q, r, paths=("auto",), shards=(1, 2))
        _check_bf16(pkg, orc, q, r, paths=("auto", "mfma"), shards=(1, 2))
        big = (rng.random((m, k), dtype=np.float32) * np.float32(3e18), rng.random((n, k), dtype=np.float32) * np.float32(3e18))
        _check(pkg, orc, big[0], big[1], paths=("auto",), shards=(1,))
    m, n, k = 8192, 262144, 256                                       # 128 MiB of bf16 refs: chunked upload
    q = orc.round_bf16(orc.rng_uniform(m * k, 92, 0).reshape(m, k))
    r = orc.round_bf16(orc.rng_uniform(n * k, 92, m * k).reshape(n, k))
    r[n - 3] = r[5]
    q[0] = r[5]
    want_idx, want_dist = orc.v0_search(q, r, threads=16)
    idx, dist = pkg.search_bf16(pkg.to_bf16_bits(q), pkg.to_bf16_bits(r), return_distances=True)
    assert np.array_equal(idx, want_idx) and np.array_equal(_bits(dist), _bits(want_dist))


def test_grouped_rccl_all_reduce_runs_with_one_rank(pkg, orc):
    """nns_search_*_multi's collective branch — ncclCommInitAll over the device list, ncclGroupStart / one
    ncclAllReduce(uint64, min) per device on its own stream / ncclGroupEnd (the V8/V9 shape, core.cu:965-1057) — with
    NNS_MULTI_FORCE_COLLECTIVE: no single-GPU shortcut, so the branch executes on a one-GPU box as a 1-rank group.
    (Among >= 2 ranks it first runs in the driver's multi-GPU bench.)"""
    rng = np.random.default_rng(57)
    for (m, n, k, bf16) in [(300, 40000, 128, False), (1000, 300000, 3, False), (130, 9000, 256, True)]:
        q = rng.random((m, k), dtype=np.float32)
        r = rng.random((n, k), dtype=np.float32)
        r[n - 1] = r[0]
        q[0] = r[0]
        if bf16:
            q, r = orc.round_bf16(q), orc.round_bf16(r)
        want_idx, want_dist = orc.v0_search(q, r, threads=8)
        qa, ra = (pkg.to_bf16_bits(q), pkg.to_bf16_bits(r)) if bf16 else (q, r)
        for rep in range(2):               # second call: the cached communicator
            idx, dist = pkg.search_multi(qa, ra, num_devices=1, return_distances=True, force_collective=True, bf16=bf16)
            assert pkg.multi_last_exchange_ranks() == 1          # the grouped all-reduce completed, on 1 rank
            assert np.array_equal(idx, want_idx)
            assert np.array_equal(_bits(dist), _bits(want_dist))
    pkg.shutdown()                         # destroys the cached communicator
    idx = pkg.search_multi(qa, ra, num_devices=1, force_collective=True, bf16=True)       # and builds a new one
    assert np.array_equal(idx, want_idx) and pkg.multi_last_exchange_ranks() == 1


def test_library_never_waits_for_foreign_streams(pkg, orc):
    """Library manners: a long kernel the APPLICATION has running on another (non-blocking) stream must still be
    running when index destroy / workspace regrow / stats read-outs / whole calls of every size class return —
    none of them may synchronise the device (round 2: 17 hipDeviceSynchronize() calls on those paths)."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(58)
    small = (rng.random((64, 16), dtype=np.float32), rng.random((1024, 16), dtype=np.float32))      # pinned-scratch path
    mid = (rng.random((300, 64), dtype=np.float32), rng.random((20000, 64), dtype=np.float32))      # plain path (> 2 MiB)
    big = (rng.random((2048, 128), dtype=np.float32), rng.random((65536, 128), dtype=np.float32))   # chunked upload
    wants = [orc.v0_search(q, r, threads=8)[0] for q, r in (small, mid, big)]
    qd, rd = torch.from_numpy(mid[0]).to(dev), torch.from_numpy(mid[1]).to(dev)
    qd2 = torch.from_numpy(rng.random((3000, 64), dtype=np.float32)).to(dev)

    marks = []

    def mark(name):
        marks.append((name, __import__("time").perf_counter()))

    def work():
        mark("begin")
        for name, (q, r), want in zip(("small", "mid", "big"), (small, mid, big), wants):
            assert np.array_equal(pkg.search(q, r), want)
            mark("whole call " + name)
        assert np.array_equal(pkg.search_multi(mid[0], mid[1], num_devices=2, virtual=True), wants[1])
        mark("search_multi (virtual shards)")
        ix = pkg.Index(rd, profile=True)
        mark("index create")
        a = ix.search(qd)
        mark("search")
        ix.search(qd2)                       # query workspace regrow while the first search may still run
        mark("search, workspace regrow")
        ix.stats()
        mark("stats")
        ix.near_ties()
        mark("near_ties")
        ix.close()                           # destroy right behind asynchronous work
        mark("destroy")
        ix2 = pkg.Index(rd)                  # re-uses (or not) the blocks just handed back
        b = ix2.search(qd)
        ix2.close()
        mark("create + search + destroy")
        torch.cuda.current_stream().synchronize()
        assert np.array_equal(a.cpu().numpy(), wants[1]) and np.array_equal(b.cpu().numpy(), wants[1])
        mark("results to the host")

    work()                                   # warm: code objects, pool, pinned scratch, library streams
    torch.cuda.synchronize()
    # The runtime multiplexes HIP streams onto a few hardware queues (four by default), and work behind the foreign
    # kernel on ITS queue waits for it whatever a library does: which of torch's pooled side streams shares a queue
    # with one of the library's streams is the runtime's business.  A device-wide synchronisation, on the other hand,
    # waits for EVERY foreign stream.  So: up to four foreign streams in turn (consecutive pool streams land on
    # different queues); the library is clean if for at least one of them nothing waited.
    outcomes = []
    for attempt in range(4):
        side = torch.cuda.Stream()           # torch's side streams are hipStreamNonBlocking
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(4e9))      # a foreign kernel that spins for ~1.7 s
            done.record()
        marks.clear()
        work()
        still_running = not done.query()
        side.synchronize()
        spans = [(n1, round(1e3 * (t1 - t0), 2)) for (n0, t0), (n1, t1) in zip(marks, marks[1:])]
        outcomes.append((still_running, round(marks[-1][1] - marks[0][1], 3), spans))
        if still_running and marks[-1][1] - marks[0][1] < 1.0:
            break
    assert any(ok and dt < 1.0 for ok, dt, _ in outcomes), \
        f"the library waited for the application's stream on every attempt; (still running, seconds, ms per operation): {outcomes}"


def test_entry_points_restore_the_current_device(pkg, orc):
    """Every C-ABI entry point that selects a device puts the caller's current device back (round 2 left the
    index's device selected, and nns_search_*_multi ended with hipSetDevice(0)).  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the driver's multi-GPU box)")
    rng = np.random.default_rng(59)
    q, r = rng.random((200, 32), dtype=np.float32), rng.random((5000, 32), dtype=np.float32)
    want = orc.v0_search(q, r, threads=8)[0]
    torch.cuda.set_device(1)
    assert np.array_equal(pkg.search(q, r, device=0), want) and torch.cuda.current_device() == 1
    assert np.array_equal(pkg.search_multi(q, r, num_devices=2, force_collective=True), want)
    assert pkg.multi_last_exchange_ranks() == 2 and torch.cuda.current_device() == 1
    rd = torch.from_numpy(r).to("cuda:0")
    ix = pkg.Index(rd)
    assert torch.cuda.current_device() == 1
    got = ix.search(torch.from_numpy(q).to("cuda:0"), stream=torch.cuda.current_stream(0))
    assert torch.cuda.current_device() == 1
    ix.stats()
    ix.close()
    assert torch.cuda.current_device() == 1
    torch.cuda.synchronize(0)
    assert np.array_equal(got.cpu().numpy(), want)
    torch.cuda.set_device(0)


def test_k1c_streaming_kernel_few_queries(pkg, orc):
    """K1c (m <= 4 queries over short rows: the HBM-bound shapes of main.cu:39-42): rows spread over L lanes with V0's
    chain passed along them (k = 4, 8, 16, 32), a row per lane with one k-dword load (k = 1, 2, 3); full and ragged
    tiles, exact ties (lowest index), NaN / INF rows, index_base, the in-kernel sharded merge re-arming itself
    call after call, the fused unpack, and an unaligned ref pointer falling back to K1b."""
    rng = np.random.default_rng(60)
    dev = torch.device("cuda:0")
    for k in (1, 2, 3, 4, 8, 16, 32):
        for n in (1, 63, 64, 257, 1000, 4097, 70001, 300000):
            for m in (1, 2, 3, 4):
                if n > 5000 and m in (2, 3) and k not in (3, 16):
                    continue
                q = rng.random((m, k), dtype=np.float32)
                r = rng.random((n, k), dtype=np.float32)
                if n >= 64:
                    r[n - 1] = r[7]                       # exact tie across the stream: index 7 wins if it is the minimum
                    q[0] = r[7]
                    r[n // 2] = np.nan
                    r[n // 3, 0] = np.inf
                want_idx, want_dist = orc.v0_search(q, r, threads=8)
                idx, dist = pkg.search(q, r, return_distances=True, path="exact")
                assert np.array_equal(idx, want_idx), (k, n, m)
                assert np.array_equal(_bits(dist), _bits(want_dist)), (k, n, m)
    # device API: one index searched repeatedly (accumulators / counters re-arm), index_base, fused unpack
    for k, n in ((16, 100000), (3, 65536), (8, 5000)):
        r = rng.random((n, k), dtype=np.float32)
        rd = torch.from_numpy(r).to(dev)
        ix = pkg.Index(rd, index_base=1000, path="exact")
        for rep in range(5):
            m = 1 + rep % 4
            q = rng.random((m, k), dtype=np.float32)
            want_idx, want_dist = orc.v0_search(q, r, threads=8)
            qd = torch.from_numpy(q).to(dev)
            keys = torch.empty(m, dtype=torch.int64, device=dev)
            dist = torch.empty(m, dtype=torch.float32, device=dev)
            idx = ix.search_indices(qd, keys, None, dist)
            torch.cuda.synchronize()
            assert np.array_equal(idx.cpu().numpy(), want_idx + 1000), (k, n, rep)
            assert np.array_equal(_bits(dist.cpu().numpy()), _bits(want_dist))
            assert np.array_equal((keys.cpu().numpy() & 0xFFFFFFFF).astype(np.int64), want_idx + 1000)
        ix.close()
    # refs that start 4 bytes off a 16-byte boundary: not K1c's 16-byte pieces — K1b takes over, same answer
    k, n = 16, 5000
    flat = torch.from_numpy(rng.random(n * k + 1, dtype=np.float32)).to(dev)
    r_off = flat[1:].view(n, k)
    assert r_off.data_ptr() % 16 == 4
    q = rng.random((1, k), dtype=np.float32)
    want_idx, _ = orc.v0_search(q, r_off.cpu().numpy(), threads=8)
    ix = pkg.Index(r_off, path="exact")
    got = ix.search(torch.from_numpy(q).to(dev))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want_idx)
    ix.close()


@pytest.mark.parametrize("k", [1, 2, 3])
def test_k1f_low_dim_filter_and_rerank_paths(pkg, orc, k):
    """K1f (exact_kernels.hip): k <= 3 from 2^27 pairs runs as a VALU FMA filter + V0 re-rank of the best two 16-ref
    chunks.  Every exit of it against the oracle: the plain case, near-ties that put the answer in the SECOND chunk,
    three chunks within tau (the workgroup scans the range for that query), so many of those that the workgroup walks
    its range exactly, NaN / INF / huge refs (the same walk), clouds far from the origin, lattices, ragged sizes."""
    rng = np.random.default_rng(900 + k)
    m, n = 2048 + 37, 65536 + 13        # > 2^27 pairs, ranges of >= 512 refs, both sizes ragged (two shards: K1a)
    q = rng.random((m, k), dtype=np.float32)
    r = rng.random((n, k), dtype=np.float32)
    _check(pkg, orc, q, r, paths=("auto", "exact"), shards=(1, 2))
    # a cloud far from the origin (the centred scores keep their resolution)
    _check(pkg, orc, q + np.float32(4096.0), r + np.float32(4096.0), paths=("exact",))
    # every ref four times, 16411 refs apart: ties across chunks, waves and ranges (the lowest index wins), and more
    # than two chunks within tau for EVERY query
    base = rng.random((16411, k), dtype=np.float32)
    r4 = np.concatenate([base, base, base, base])[:n]
    _check(pkg, orc, q, r4, paths=("exact",))
    # a few planted near-duplicates: for some queries three chunks hold a ref within tau of the best
    r5 = r.copy()
    for t in range(40):
        src = int(rng.integers(0, n))
        for d in (1, 2, 3):
            r5[(src + 1777 * d) % n] = r5[src] + np.float32(1e-7) * d
    q5 = q.copy()
    q5[:40] = r5[rng.integers(0, n, 40)] + np.float32(1e-3)
    _check(pkg, orc, q5, r5, paths=("exact",))
    # integer lattice: exact ties everywhere
    rl = rng.integers(0, 24, (n, k)).astype(np.float32)
    ql = rng.integers(0, 24, (m, k)).astype(np.float32) + np.float32(0.5)
    _check(pkg, orc, ql, rl, paths=("exact",))
    # non-finite and huge refs (a workgroup that sees one walks its range with V0's arithmetic), non-finite queries
    r6 = r.copy()
    r6[5, 0] = np.nan
    r6[700, k - 1] = np.inf
    r6[9000, 0] = np.float32(3e18)
    r6[12000, 0] = np.float32(-1e30)
    q6 = q.copy()
    q6[3, 0] = np.nan
    q6[4, k - 1] = np.inf
    q6[5, 0] = np.float32(2e19)
    _check(pkg, orc, q6, r6, paths=("exact",))
