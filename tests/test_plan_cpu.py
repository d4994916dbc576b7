"""CPU tests of the host-side planning logic behind the MFMA filter launch (nns_plan_filter, no device):
tile depth by dimensionality and dtype, padding, and that the ref-range splits tile the ring slots exactly —
a wrong plan here is an out-of-bounds DMA on the GPU."""
import numpy as np
import pytest


def _check_plan(p, k, m, n):
    assert p["kt"] >= k
    assert p["m_pad"] >= m and p["m_pad"] % p["queries_per_wg"] == 0 and p["m_pad"] - m < p["queries_per_wg"]
    assert p["qgroups"] * p["queries_per_wg"] == p["m_pad"]
    assert p["n_pad"] >= n and p["n_pad"] % 32 == 0
    if p["kt"] == 768:      # two 48-step blocks (64 refs) per three ring slots
        assert p["n_pad"] % 64 == 0 and p["total_slots"] * 64 == p["n_pad"] * 3 and p["slots_per_split"] % 3 == 0
    elif p["kt"] == 640:    # four 40-step blocks (128 refs) per five ring slots
        assert p["n_pad"] % 128 == 0 and p["total_slots"] * 128 == p["n_pad"] * 5 and p["slots_per_split"] % 5 == 0
    elif p["kt"] == 384:    # four 24-step blocks (128 refs) per three ring slots
        assert p["n_pad"] % 128 == 0 and p["total_slots"] * 128 == p["n_pad"] * 3 and p["slots_per_split"] % 3 == 0
    else:
        assert p["n_pad"] == p["total_slots"] * p["slot_pts"]
    # splits x slots_per_split covers every slot, the last split is not empty
    assert p["splits"] >= 1 and p["slots_per_split"] >= 1
    assert p["splits"] * p["slots_per_split"] >= p["total_slots"] > (p["splits"] - 1) * p["slots_per_split"]
    assert p["splits"] <= 65535
    # record form: per (lane, ref tile) only on short streams, and only with K5's one-wave-per-query form
    tiles_per_stream = (p["slots_per_split"] * 2 // 3 if p["kt"] == 768 else p["slots_per_split"] * 4 // 5 if p["kt"] == 640 else
                        p["slots_per_split"] * 4 // 3 if p["kt"] == 384 else
                        p["slots_per_split"] * max(p["slot_pts"], 32) // 32 // (2 if p["kt"] == 1024 else 1))
    if p["tile_rec"]:
        assert p["splits"] >= 4 and tiles_per_stream <= 2048 and p["share_thr"] == 1
    elif p["splits"] >= 4:
        assert tiles_per_stream > 2048
    # candidate list memory stays bounded (512 B per lane-list)
    assert p["splits"] * p["m_pad"] * p["lpq"] * 512 <= (2 << 30) or p["splits"] * p["qgroups"] <= 512


def test_tile_depth_by_dimensionality(pkg):
    want_f32 = {8: 16, 16: 16, 17: 32, 32: 32, 33: 64, 64: 64, 65: 128, 128: 128, 129: 256, 256: 256}
    for k, kt in want_f32.items():
        p = pkg.plan_filter(k, 1000, 50000)
        assert (p["kt"], p["bf16"], p["mixed"]) == (kt, 0, 0), (k, p)
    for k, kt in {257: 384, 384: 384, 385: 512, 512: 512, 513: 640, 640: 640, 641: 768, 768: 768, 769: 1024, 1024: 1024}.items():      # fp32 points beyond the fp32 tiles
        p = pkg.plan_filter(k, 1000, 50000)
        assert (p["kt"], p["bf16"], p["mixed"]) == (kt, 1, 1), (k, p)
    for k, kt in {32: 128, 128: 128, 129: 256, 256: 256, 257: 384, 384: 384, 385: 512, 512: 512, 600: 640, 640: 640, 700: 768, 768: 768, 769: 1024, 1024: 1024}.items():
        p = pkg.plan_filter(k, 1000, 50000, bf16=True)
        assert (p["kt"], p["bf16"], p["mixed"]) == (kt, 1, 0), (k, p)
    assert pkg.plan_filter(100, 1000, 50000, flags=pkg.NNS_FILTER_BF16)["mixed"] == 1
    with pytest.raises(pkg.NNSError) as e:
        pkg.plan_filter(1025, 10, 10, bf16=True)
    assert e.value.status == 5


def test_headline_geometries(pkg):
    p = pkg.plan_filter(128, 65536, 1048576)                 # C3: one workgroup per CU, two ref ranges
    assert (p["qgroups"], p["splits"], p["slot_pts"], p["queries_per_wg"], p["lpq"]) == (128, 2, 64, 512, 2)
    assert (p["share_thr"], p["tile_rec"]) == (0, 0)          # long streams: private thresholds, a record per score
    p = pkg.plan_filter(256, 131072, 2097152, bf16=True)     # C5: 16x16 tiles, four lists per query
    assert (p["qgroups"], p["splits"], p["lpq"]) == (256, 1, 4) and (p["share_thr"], p["tile_rec"]) == (0, 0)
    p = pkg.plan_filter(16, 1024, 1048576)                   # the reference driver's 16-D sample: 16-deep tile
    assert (p["kt"], p["slot_pts"], p["qgroups"]) == (16, 512, 2) and p["qgroups"] * p["splits"] >= 256
    assert (p["share_thr"], p["tile_rec"]) == (1, 2)          # 128 short streams, 32 x 32 tiles: the lane's two best tiles
    assert pkg.plan_filter(256, 1024, 1048576, bf16=True)["tile_rec"] == 1      # 16 x 16 tiles: a record per tile, thresholds
    assert pkg.plan_filter(16, 1024, 1048576, flags=pkg.NNS_RECORDS_PER_REF)["tile_rec"] == 0
    p = pkg.plan_filter(1024, 65536, 1048576, bf16=True)     # 1024-deep: 128 queries per workgroup, blocks of two slots
    assert (p["queries_per_wg"], p["slot_pts"]) == (128, 16) and p["slots_per_split"] % 2 == 0 and p["total_slots"] % 2 == 0
    p = pkg.plan_filter(700, 65536, 1048576, bf16=True)      # 768-deep: 256 queries per workgroup (two waves per SIMD)
    assert (p["kt"], p["queries_per_wg"], p["lpq"]) == (768, 256, 2) and p["total_slots"] == 1048576 // 64 * 3
    p = pkg.plan_filter(600, 65536, 1048576, bf16=True)      # 640-deep: four blocks per five slots
    assert (p["kt"], p["queries_per_wg"], p["lpq"]) == (640, 256, 2) and p["total_slots"] == 1048576 // 128 * 5


def test_plan_invariants_random_shapes(pkg):
    rng = np.random.default_rng(2026)
    for _ in range(3000):
        bf16 = bool(rng.integers(0, 2))
        k = int(rng.integers(32 if bf16 else 8, 1025))
        m = int(rng.choice([1, 63, 64, 65, 511, 512, 513, 4096, 65536, 200000]))
        n = int(rng.choice([1, 31, 32, 33, 511, 513, 4097, 65536, 1000003, 8388608]))
        p = pkg.plan_filter(k, m, n, bf16=bf16)
        _check_plan(p, k, m, n)
        if p["kt"] == 1024:
            assert p["slots_per_split"] % 2 == 0 and p["total_slots"] % 2 == 0


def test_exact_path_plans(pkg):
    """nns_plan_exact (host only): which exact kernel a shape takes and how its grid covers the refs.  K1f (the
    vector-ALU filter + V0 re-rank, k <= 3) starts at 2^27 pairs and needs ranges of >= 512 refs in whole 16-ref chunks;
    the reference driver's samples (main.cu:38-51) land where DESIGN says; sizes near 2^31 do not overflow."""
    P = pkg.plan_exact
    # the reference driver's table
    assert P(3, 1, 1024)["kernel"] == "k1c" and P(16, 1, 65536)["kernel"] == "k1c"
    assert P(3, 1024, 1024)["kernel"] == "k1a" and P(3, 1024, 65536)["kernel"] == "k1a"     # 2^26 pairs: K1a
    assert P(3, 1024, 1048576)["kernel"] == "k1f"
    c2 = P(3, 4096, 65536)
    assert c2["kernel"] == "k1f" and c2["waves"] == 8 and c2["queries_per_wg"] == 128
    assert c2["qtiles"] * c2["splits"] * 8 <= 4096 and c2["per"] % 16 == 0 and c2["per"] >= 512
    # K1f needs k <= 3, the merge workspace (or a single range) and >= 2^27 pairs
    assert P(4, 4096, 65536)["kernel"] == "k1a" and P(8, 4096, 65536)["kernel"] == "k1a"
    assert P(3, 4096, 65536, have_workspace=False)["kernel"] == "k1a"
    assert P(3, 2048, 65535)["kernel"] == "k1a" and P(3, 2048, 65536)["kernel"] == "k1f"
    assert P(5, 4096, 65536)["kernel"] == "k1b" and P(3, 63, 65536)["kernel"] == "k1b"
    assert P(16, 4, 1048576, refs_aligned=False)["kernel"] == "k1b" and P(3, 4, 1048576, refs_aligned=False)["kernel"] == "k1c"
    rng = np.random.default_rng(5)
    for _ in range(3000):
        k = int(rng.choice([1, 2, 3, 4, 8, 16]))
        m = int(rng.integers(64, 200000)) if rng.integers(0, 4) else int(rng.integers(64, 2**31 - 2**20))
        n = int(rng.integers(1, 3000000)) if rng.integers(0, 4) else int(rng.integers(1, 2**31 - 2**20))
        p = P(k, m, n)
        assert p["kernel"] in ("k1a", "k1f")
        assert p["splits"] >= 1 and p["per"] >= 1 and p["qtiles"] * p["queries_per_wg"] >= m
        assert p["splits"] * p["per"] >= n and (p["splits"] - 1) * p["per"] < n          # the ranges cover the refs, none is empty
        assert p["splits"] <= 65535 and 1 <= p["waves"] <= 16
        if p["kernel"] == "k1f":
            assert k <= 3 and m * n >= 2**27 and p["per"] % 16 == 0 and p["waves"] == 8
            assert p["per"] >= 512 or p["splits"] == 1
