"""CPU tests of the boundary: the C-ABI library loads, exports every symbol of
include/nns.h, validates arguments, and fails loudly without a device (no CPU
fallback).  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(os.path.dirname(pkg.LIB_PATH), "..", "include", "nns.h")).read()
    declared = set(re.findall(r"\b(nns_[a-z0-9_]+)\s*\(", hdr)) - {"nns_rng_fill"}
    assert declared == set(pkg.ABI_SYMBOLS), declared ^ set(pkg.ABI_SYMBOLS)
    raw = ctypes.CDLL(pkg.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None


def test_version_and_strerror(pkg):
    assert pkg.lib.nns_version() == 1
    assert pkg.lib.nns_strerror(0) == b"ok"
    assert b"invalid" in pkg.lib.nns_strerror(1)


def test_no_torch_types_in_header(pkg):
    hdr = open(os.path.join(os.path.dirname(pkg.LIB_PATH), "..", "include", "nns.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # strip comments
    assert "torch" not in code.lower() and "at::" not in code and "hipStream" not in code
    assert "#include <hip" not in hdr


def test_argument_validation_returns_status_not_exit(pkg):
    q = np.zeros((4, 3), np.float32)
    with pytest.raises(pkg.NNSError) as e:
        pkg.search(q, np.zeros((0, 3), np.float32))
    assert e.value.status == 1
    res = ctypes.POINTER(ctypes.c_int)()
    assert pkg.lib.nns_search_f32(3, 0, 5, q.ctypes.data, q.ctypes.data, ctypes.byref(res)) == 1
    assert pkg.lib.nns_search_f32(3, 4, 4, q.ctypes.data, q.ctypes.data, None) == 1
    assert pkg.lib.nns_index_create(None, 0, 3, 4, q.ctypes.data, 0, 0, None) == 1
    assert b"nns_" in pkg.lib.nns_last_error()
    # NNS_MAX_POINTS = 2^31 - 2^20 points per set (validated before anything is read or allocated)
    idx = np.zeros(4, np.int32)
    assert pkg.lib.nns_search_f32_ex(1, 4, 0x7FF00001, q.ctypes.data, q.ctypes.data, idx.ctypes.data, None, 1, 0, 0) == 1
    assert b"NNS_MAX_POINTS" in pkg.lib.nns_last_error()
    # the multi-GPU entry points apply the same limits, before any thread or allocation
    assert pkg.lib.nns_search_f32_multi(1, 4, 0x7FF00001, q.ctypes.data, q.ctypes.data, idx.ctypes.data, None, 0, 0) == 1
    assert b"NNS_MAX_POINTS" in pkg.lib.nns_last_error()
    assert pkg.lib.nns_search_bf16_multi(1, 0x7FF00001, 4, q.ctypes.data, q.ctypes.data, idx.ctypes.data, None, 0, 0) == 1
    assert pkg.lib.nns_search_f32_multi(3, 0, 4, q.ctypes.data, q.ctypes.data, idx.ctypes.data, None, 0, 0) == 1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device error path")
def test_fails_loudly_without_device(pkg):
    """The product path has no CPU fallback: without a HIP device it must raise."""
    assert pkg.device_count() == 0
    q = np.random.default_rng(0).random((4, 3), dtype=np.float32)
    with pytest.raises(pkg.NNSError) as e:
        pkg.cudaCall(3, 4, 4, q, q)
    assert e.value.status == 4   # NNS_ERR_NODEVICE
    with pytest.raises(pkg.NNSError):
        pkg.search(q, q)


def test_shipped_library_has_no_diagnostic_switches(pkg):
    """The product build carries no environment variable that changes results (the timing
    diagnostics NNS_DIAG_FILTER_ONLY / NNS_FILTER_CLOCK exist only under -DNNS_DIAG)."""
    blob = open(pkg.LIB_PATH, "rb").read()
    for name in (b"NNS_DIAG_FILTER_ONLY", b"NNS_FILTER_CLOCK", b"NNS_FILTER_ABLATE"):
        assert name not in blob, name
    # the only environment variable the library reads: the pool cap
    # (other NNS_* strings are names of include/nns.h flags quoted in error messages)
    names = {n for n in re.findall(rb"NNS_[A-Z_]{3,}", blob)
             if not re.match(rb"NNS_(PATH_|FILTER_BF|REFS_SOA|COMM_ID_BYTES|MULTI_VIRTUAL|MULTI_FORCE_COLLECTIVE|ERR_|KEY_NONE|MAX_POINTS)", n)}
    assert names == {b"NNS_POOL_BYTES"}, names


def test_no_device_wide_synchronisation_in_the_product():
    """Library manners: destroy / workspace regrow / whole-call exits free behind stream events (dev_pool.hip), read-outs
    wait for the index's stream.  hipDeviceSynchronize() may only appear as the documented last resort when a stream
    handle has already been destroyed by its owner, and hipSetDevice(0) never as an exit path."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "nns-cuda_amd", "csrc")
    allowed = {"dev_pool.hip": 1, "nns_multi.hip": 1}      # pool_free_after's and nns_comm_destroy's fallbacks
    for f in sorted(os.listdir(src)):
        if not f.endswith((".hip", ".h", ".hpp", ".cpp")):
            continue
        text = open(os.path.join(src, f)).read()
        code = re.sub(r"//[^\n]*", "", text)
        assert code.count("hipDeviceSynchronize(") <= allowed.get(f, 0), f
        assert "hipSetDevice(0)" not in code, f


def test_shard_range_is_reference_split(pkg):
    # core.cu:781-791: contiguous ceil(n/G), last takes the remainder
    assert [pkg.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert [pkg.shard_range(8, 8, r) for r in range(8)] == [(i, 1) for i in range(8)]
    assert pkg.shard_range(5, 8, 7) == (7, 0)
    n = 8388608
    assert sum(pkg.shard_range(n, 8, r)[1] for r in range(8)) == n


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing in the package or bench's product
    leg may reference it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "nns-cuda_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "libv0oracle" not in text and "v0_oracle" not in text.replace(
                    "oracle/v0_oracle.c:nns_rng_fill", ""), f
