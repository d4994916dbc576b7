"""Host-side mirror of the reference's entry point over the C ABI (include/nns.h).

The reference's operator interface for this path is a single C++ function,
``vN::cudaCall(k, m, n, s_points, r_points, &results)`` (reference
core.cu:23-29, dispatched through the function pointer of main.cu:7 and called at
main.cu:74).  ``cudaCall`` below has the same argument order and meaning and the
same result (a fresh int32 array of m global reference indices); the C++ twin is
``mi355x::cudaCall`` in csrc/nns_cudacall.hpp.  Everything else here is the
split (device-resident) API used by the tests and bench.py.

This module is plumbing: ctypes bindings to libnns_mi355x.so (hand-written HIP for
gfx950).  There is NO CPU fallback: if the library is missing the import raises,
and every entry point raises ``NNSError`` when the library reports a failure
(no device, bad arguments, HIP error) — the reference's CHECK macro prints and
exits (utils.h:16-26); a Python host gets an exception instead.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NNS_LIB_PATH") or os.path.join(_HERE, "libnns_mi355x.so")   # override: A/B of builds

NNS_OK = 0
NNS_PATH_AUTO, NNS_PATH_EXACT, NNS_PATH_MFMA, NNS_PROFILE, NNS_MULTI_VIRTUAL, NNS_REFS_SOA = 0, 1, 2, 16, 32, 64
NNS_FILTER_BF16 = 128
NNS_MULTI_FORCE_COLLECTIVE = 256
NNS_KEY_NONE = 0x7F80000000000000

NNS_RECORDS_PER_REF = 512
# "mfma_perref": the MFMA filter with per-score candidate records forced (the long-stream form) at any size
_PATHS = {"auto": NNS_PATH_AUTO, "exact": NNS_PATH_EXACT, "mfma": NNS_PATH_MFMA,
          "mfma_perref": NNS_PATH_MFMA | NNS_RECORDS_PER_REF}

# every symbol include/nns.h declares (tests check the library exports them all)
ABI_SYMBOLS = (
    "nns_search_f32", "nns_search_f32_ex", "nns_index_create", "nns_index_destroy",
    "nns_index_refresh", "nns_index_search", "nns_index_stats", "nns_keys_min",
    "nns_keys_unpack", "nns_fill_uniform", "nns_device_count", "nns_strerror",
    "nns_last_error", "nns_version", "nns_selftest_mfma",
    "nns_index_create_bf16", "nns_index_search_bf16", "nns_search_bf16_ex", "nns_search_f32_multi",
    "nns_trim", "nns_warmup", "nns_shutdown", "nns_search_bf16_multi",
    "nns_index_near_ties", "nns_tau_consts", "nns_index_search_indices", "nns_selftest_lane_share", "nns_plan_filter", "nns_plan_exact",
    "nns_comm_unique_id", "nns_comm_create", "nns_comm_size", "nns_comm_allreduce_min", "nns_comm_destroy",
    "nns_multi_last_exchange_ranks",
)
NNS_COMM_ID_BYTES = 128


class NNSError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, status: int, where: str, detail: str):
        super().__init__(f"{where}: status {status} ({detail})")
        self.status = status


class nns_stats(ctypes.Structure):
    _fields_ = [
        ("path", ctypes.c_int), ("k_tile", ctypes.c_int), ("splits", ctypes.c_int),
        ("ambiguous", ctypes.c_int), ("nonfinite", ctypes.c_int),
        ("prep_refs_ms", ctypes.c_float), ("prep_queries_ms", ctypes.c_float),
        ("filter_ms", ctypes.c_float), ("finalize_ms", ctypes.c_float),
        ("rerank_ms", ctypes.c_float), ("exact_ms", ctypes.c_float),
        ("total_ms", ctypes.c_float), ("multi_candidate", ctypes.c_int),
    ]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C nns-cuda_amd/csrc`). The HIP extension is the product; there is no CPU fallback.")
    # If torch is (going to be) in the process, let it bring the HIP runtime first so
    # that one libamdhip64.so.7 serves both (same SONAME in torch/lib and /opt/rocm).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the host-buffer API
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    c_int, c_vp, c_sz, c_u64, c_i64, c_u = (ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_uint64, ctypes.c_int64, ctypes.c_uint)
    lib.nns_search_f32.argtypes = [c_int, c_int, c_int, c_vp, c_vp, ctypes.POINTER(ctypes.POINTER(c_int))]
    lib.nns_search_f32_ex.argtypes = [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_u, c_int]
    lib.nns_index_create.argtypes = [ctypes.POINTER(c_vp), c_int, c_int, c_int, c_vp, c_i64, c_u, c_vp]
    lib.nns_index_create_bf16.argtypes = lib.nns_index_create.argtypes
    lib.nns_index_search_bf16.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp]
    lib.nns_search_bf16_ex.argtypes = lib.nns_search_f32_ex.argtypes
    lib.nns_search_f32_multi.argtypes = [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_u]
    lib.nns_search_bf16_multi.argtypes = lib.nns_search_f32_multi.argtypes
    lib.nns_comm_unique_id.argtypes = [c_vp, c_sz]
    lib.nns_comm_create.argtypes = [ctypes.POINTER(c_vp), c_vp, c_sz, c_int, c_int, c_int]
    lib.nns_comm_size.argtypes = [c_vp]
    lib.nns_comm_allreduce_min.argtypes = [c_vp, c_vp, c_int, c_vp]
    lib.nns_comm_destroy.argtypes = [c_vp]
    lib.nns_shutdown.argtypes = []
    lib.nns_multi_last_exchange_ranks.argtypes = []
    lib.nns_index_destroy.argtypes = [c_vp]
    lib.nns_index_refresh.argtypes = [c_vp, c_vp]
    lib.nns_index_search.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp]
    lib.nns_index_stats.argtypes = [c_vp, ctypes.POINTER(nns_stats)]
    lib.nns_index_search_indices.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
    lib.nns_index_near_ties.argtypes = [c_vp, c_vp, c_int, ctypes.POINTER(c_int)]
    lib.nns_tau_consts.argtypes = [c_int, ctypes.c_float, ctypes.c_float, c_int, c_vp]
    lib.nns_selftest_lane_share.argtypes = [c_int, c_vp, c_vp]
    lib.nns_plan_filter.argtypes = [c_int, c_int, c_int, c_int, c_u, c_vp, c_int]
    lib.nns_plan_exact.argtypes = [c_int, c_int, c_int, c_int, c_int, c_vp, c_int]
    lib.nns_keys_min.argtypes = [c_vp, c_vp, c_int, c_vp]
    lib.nns_keys_unpack.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp]
    lib.nns_fill_uniform.argtypes = [c_vp, c_sz, c_u64, c_u64, c_vp]
    lib.nns_selftest_mfma.argtypes = [c_int, c_int, c_vp, c_vp, c_vp, c_vp]
    lib.nns_device_count.argtypes = []
    lib.nns_strerror.argtypes = [c_int]
    lib.nns_strerror.restype = ctypes.c_char_p
    lib.nns_last_error.argtypes = []
    lib.nns_last_error.restype = ctypes.c_char_p
    lib.nns_version.argtypes = []
    for name in ABI_SYMBOLS:
        if name not in ("nns_strerror", "nns_last_error", "nns_trim"):
            getattr(lib, name).restype = c_int
    lib.nns_warmup.argtypes = [c_int]
    lib.nns_trim.argtypes = []
    lib.nns_trim.restype = ctypes.c_size_t
    return lib


lib = _load()
_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def _check(status: int, where: str) -> None:
    if status != NNS_OK:
        detail = lib.nns_last_error().decode() or lib.nns_strerror(status).decode()
        raise NNSError(status, where, detail)


def selftest_mfma(a: np.ndarray, b: np.ndarray, c0: np.ndarray, bf16: bool = False) -> np.ndarray:
    """out[i][j] of one 32x32 MFMA tile (diagnostic, see include/nns.h)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    c0 = np.ascontiguousarray(c0, np.float32)
    assert a.shape == b.shape and a.shape[0] == 32 and c0.shape == (32,)
    out = np.empty((32, 32), np.float32)
    _check(lib.nns_selftest_mfma(a.shape[1], int(bf16), a.ctypes.data, b.ctypes.data, c0.ctypes.data, out.ctypes.data),
           "nns_selftest_mfma")
    return out


def plan_filter(k: int, m: int, n: int, bf16: bool = False, flags: int = 0) -> dict:
    """nns_plan_filter: the MFMA filter's launch geometry for a shape (host only)."""
    out = np.zeros(14, np.int32)
    _check(lib.nns_plan_filter(k, m, n, int(bf16), flags, out.ctypes.data, 14), "nns_plan_filter")
    names = ("kt", "bf16", "mixed", "lpq", "m_pad", "n_pad", "total_slots", "splits", "slots_per_split", "qgroups",
             "slot_pts", "queries_per_wg", "share_thr", "tile_rec")
    return dict(zip(names, (int(v) for v in out)))


def plan_exact(k: int, m: int, n: int, refs_aligned: bool = True, have_workspace: bool = True) -> dict:
    """nns_plan_exact: the exact path's launch geometry for a shape (host only)."""
    out = np.zeros(6, np.int32)
    _check(lib.nns_plan_exact(k, m, n, int(refs_aligned), int(have_workspace), out.ctypes.data, 6), "nns_plan_exact")
    names = ("kernel", "qtiles", "splits", "per", "waves", "queries_per_wg")
    d = dict(zip(names, (int(v) for v in out)))
    d["kernel"] = ("k1a", "k1f", "k1b", "k1c")[d["kernel"]]
    return d


def selftest_lane_share(values, tile16: bool) -> np.ndarray:
    """out[l] of nns_selftest_lane_share for 64 lane values."""
    v = np.ascontiguousarray(values, np.float32)
    assert v.shape == (64,)
    out = np.empty(64, np.float32)
    _check(lib.nns_selftest_lane_share(int(tile16), v.ctypes.data, out.ctypes.data), "nns_selftest_lane_share")
    return out


def tau_consts(kt: int, qnorm2: float, ymax2: float, mode: int):
    """(c0, c1, x2) of the proof margin tau(a) = c0 + c1 * max(a + x2, 0) (nns_tau_consts)."""
    out = np.empty(3, np.float32)
    _check(lib.nns_tau_consts(kt, qnorm2, ymax2, mode, out.ctypes.data), "nns_tau_consts")
    return float(out[0]), float(out[1]), float(out[2])


def device_count() -> int:
    return lib.nns_device_count()


def _as_f32(a, what: str) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError(f"{what} must be [points][k]")
    return a


def cudaCall(k: int, m: int, n: int, s_points, r_points) -> np.ndarray:
    """Drop-in for the reference's ``vN::cudaCall(k, m, n, s_points, r_points, &results)``.

    s_points: m*k fp32 (row-major queries), r_points: n*k fp32 (row-major refs);
    returns the malloc'd int32[m] result the C ABI hands back, copied into numpy.
    """
    s = np.ascontiguousarray(s_points, dtype=np.float32).reshape(-1)
    r = np.ascontiguousarray(r_points, dtype=np.float32).reshape(-1)
    if s.size < k * m or r.size < k * n:
        raise ValueError("point arrays shorter than k*m / k*n")
    res = ctypes.POINTER(ctypes.c_int)()
    _check(lib.nns_search_f32(k, m, n, s.ctypes.data, r.ctypes.data, ctypes.byref(res)), "nns_search_f32")
    try:
        return np.ctypeslib.as_array(res, shape=(m,)).astype(np.int32, copy=True)
    finally:
        _libc.free(res)


def search(query_points, reference_points, *, return_distances: bool = False, shards: int = 1,
           path: str = "auto", device: int = 0, refs_soa: bool = False, filter_bf16: bool = False):
    """``search(query_points, reference_points)`` — the entry point BASELINE.json names.

    Host arrays in, nearest-reference index per query out (and, optionally, V0's
    fp32 squared distance).  ``shards`` > 1 rehearses the multi-GPU ref split on one
    device (contiguous ceil(n/shards) ranges merged with the packed-key min).  ``refs_soa``:
    reference_points is dimension-major [k][n] (NNS_REFS_SOA, the reference's V4 layout)."""
    q = _as_f32(query_points, "query_points")
    r = _as_f32(reference_points, "reference_points")
    if q.shape[1] != (r.shape[0] if refs_soa else r.shape[1]):
        raise ValueError("query and reference dimensionality differ")
    m, k = q.shape
    n = r.shape[1] if refs_soa else r.shape[0]
    idx = np.empty(m, dtype=np.int32)
    dist = np.empty(m, dtype=np.float32) if return_distances else None
    _check(lib.nns_search_f32_ex(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data,
                                 dist.ctypes.data if dist is not None else None, shards,
                                 _PATHS[path] | (NNS_REFS_SOA if refs_soa else 0)
                                 | (NNS_FILTER_BF16 if filter_bf16 else 0), device), "nns_search_f32_ex")
    return (idx, dist) if return_distances else idx


def search_multi(query_points, reference_points, *, num_devices: int = 0, return_distances: bool = False,
                 path: str = "auto", virtual: bool = False, refs_soa: bool = False, bf16: bool = False,
                 force_collective: bool = False):
    """The V8/V9 analogue: refs sharded over `num_devices` GPUs of this process (0 = all),
    per-GPU keys combined with one RCCL min all-reduce.  `virtual` lets a 1-GPU box rehearse
    more shards than it has GPUs (host-side key merge).  `bf16`: the arrays hold bf16 bit
    patterns (uint16); `refs_soa`: reference_points is dimension-major [k][n].  `force_collective`
    (tests): no single-GPU shortcut, so that ncclCommInitAll + the grouped all-reduce also run with one
    shard on a one-GPU box (NNS_MULTI_FORCE_COLLECTIVE)."""
    if bf16:
        q = np.ascontiguousarray(query_points, dtype=np.uint16)
        r = np.ascontiguousarray(reference_points, dtype=np.uint16)
        if q.ndim != 2 or r.ndim != 2:
            raise ValueError("bf16 point sets must be 2-D arrays of bit patterns")
    else:
        q = _as_f32(query_points, "query_points")
        r = _as_f32(reference_points, "reference_points")
    if q.shape[1] != (r.shape[0] if refs_soa else r.shape[1]):
        raise ValueError("query and reference dimensionality differ")
    m, k = q.shape
    n = r.shape[1] if refs_soa else r.shape[0]
    idx = np.empty(m, dtype=np.int32)
    dist = np.empty(m, dtype=np.float32) if return_distances else None
    flags = _PATHS[path] | (NNS_MULTI_VIRTUAL if virtual else 0) | (NNS_REFS_SOA if refs_soa else 0) \
        | (NNS_MULTI_FORCE_COLLECTIVE if force_collective else 0)
    fn = lib.nns_search_bf16_multi if bf16 else lib.nns_search_f32_multi
    _check(fn(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data,
              dist.ctypes.data if dist is not None else None, num_devices, flags), "nns_search_multi")
    return (idx, dist) if return_distances else idx


def to_bf16_bits(a) -> np.ndarray:
    """fp32 array -> bf16 bit patterns (uint16), round-to-nearest-even (NaN stays NaN)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(a)
    if nan.any():
        r[nan] = ((u[nan] >> 16) | 0x0040).astype(np.uint16)
    return r


def search_bf16(query_bits, reference_bits, *, return_distances: bool = False, shards: int = 1,
                path: str = "auto", device: int = 0):
    """search() for bf16 point sets given as uint16 bit patterns [points][k] (config C5):
    V0's arithmetic on the bf16 values widened to fp32."""
    q = np.ascontiguousarray(query_bits, dtype=np.uint16)
    r = np.ascontiguousarray(reference_bits, dtype=np.uint16)
    if q.ndim != 2 or r.ndim != 2 or q.shape[1] != r.shape[1]:
        raise ValueError("bf16 point sets must be [points][k] with equal k")
    m, k = q.shape
    n = r.shape[0]
    idx = np.empty(m, dtype=np.int32)
    dist = np.empty(m, dtype=np.float32) if return_distances else None
    _check(lib.nns_search_bf16_ex(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data,
                                  dist.ctypes.data if dist is not None else None, shards,
                                  _PATHS[path], device), "nns_search_bf16_ex")
    return (idx, dist) if return_distances else idx


# ---------------------------------------------------------------------------
# device-resident API (torch tensors are only the owners of device memory)
# ---------------------------------------------------------------------------
def _stream_ptr(stream) -> Optional[int]:
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream or None
    return getattr(stream, "cuda_stream", stream) or None


def fill_uniform(t, seed: int, offset: int = 0, stream=None) -> None:
    """t[i] = synthetic uniform [0,1) value #(offset+i) of stream `seed` (same bits as
    oracle nns_rng_fill)."""
    _check(lib.nns_fill_uniform(t.data_ptr(), t.numel(), seed, offset, _stream_ptr(stream)), "nns_fill_uniform")


class Index:
    """One prepared, device-resident shard of reference points (nns_index)."""

    def __init__(self, refs, *, index_base: int = 0, path: str = "auto", profile: bool = False, stream=None,
                 soa: bool = False, filter_bf16: bool = False):
        """refs: [n][k] (or, with soa=True, dimension-major [k][n]: NNS_REFS_SOA) on a HIP device."""
        import torch
        if refs.dtype not in (torch.float32, torch.bfloat16) or refs.dim() != 2 or not refs.is_contiguous() \
                or not refs.is_cuda:
            raise ValueError("refs must be a contiguous fp32 or bf16 [n][k] tensor on a HIP device")
        self.bf16 = refs.dtype == torch.bfloat16
        self.refs = refs  # keep alive: the index reads the original values
        self.n, self.k = (refs.shape[1], refs.shape[0]) if soa else refs.shape
        self.device = refs.device.index or 0
        flags = _PATHS[path] | (NNS_PROFILE if profile else 0) | (NNS_REFS_SOA if soa else 0) \
            | (NNS_FILTER_BF16 if filter_bf16 else 0)
        h = ctypes.c_void_p()
        create = lib.nns_index_create_bf16 if self.bf16 else lib.nns_index_create
        _check(create(ctypes.byref(h), self.device, self.k, self.n, refs.data_ptr(),
                      index_base, flags, _stream_ptr(stream)), "nns_index_create")
        self._h = h

    def refresh(self, stream=None) -> None:
        _check(lib.nns_index_refresh(self._h, _stream_ptr(stream)), "nns_index_refresh")

    def search_keys(self, queries, keys=None, stream=None):
        """Packed (V0 distance, global index) int64 key per query (NNS_KEY_NONE if none)."""
        import torch
        if queries.dtype != self.refs.dtype or queries.dim() != 2 or not queries.is_contiguous():
            raise ValueError("queries must be a contiguous [m][k] tensor of the index's dtype")
        if queries.shape[1] != self.k:
            raise ValueError("query dimensionality differs from the index")
        m = queries.shape[0]
        if keys is None:
            keys = torch.empty(m, dtype=torch.int64, device=queries.device)
        fn = lib.nns_index_search_bf16 if self.bf16 else lib.nns_index_search
        _check(fn(self._h, m, queries.data_ptr(), keys.data_ptr(), _stream_ptr(stream)), "nns_index_search")
        return keys

    def search(self, queries, return_distances: bool = False, stream=None):
        keys = self.search_keys(queries, stream=stream)
        return keys_unpack(keys, return_distances=return_distances, stream=stream)

    def search_indices(self, queries, keys=None, idx=None, dist=None, stream=None):
        """nns_index_search_indices: keys AND unpacked int32 indices (and, if `dist` is given,
        distances) of a single-shard search in as few launches as the path allows."""
        import torch
        if queries.dtype != self.refs.dtype or queries.dim() != 2 or not queries.is_contiguous() \
                or queries.shape[1] != self.k:
            raise ValueError("queries must be a contiguous [m][k] tensor of the index's dtype")
        m = queries.shape[0]
        if keys is None:
            keys = torch.empty(m, dtype=torch.int64, device=queries.device)
        if idx is None:
            idx = torch.empty(m, dtype=torch.int32, device=queries.device)
        _check(lib.nns_index_search_indices(self._h, m, queries.data_ptr(), keys.data_ptr(), idx.data_ptr(),
                                            dist.data_ptr() if dist is not None else None, _stream_ptr(stream)),
               "nns_index_search_indices")
        return idx

    def stats(self) -> dict:
        st = nns_stats()
        _check(lib.nns_index_stats(self._h, ctypes.byref(st)), "nns_index_stats")
        return st.asdict()

    def near_ties(self) -> np.ndarray:
        """Query numbers of the last search that K5 decided among > 1 candidates within tau."""
        cnt = ctypes.c_int(0)
        _check(lib.nns_index_near_ties(self._h, None, 0, ctypes.byref(cnt)), "nns_index_near_ties")
        ids = np.empty(max(cnt.value, 1), dtype=np.int32)
        _check(lib.nns_index_near_ties(self._h, ids.ctypes.data, cnt.value, ctypes.byref(cnt)), "nns_index_near_ties")
        return np.sort(ids[:cnt.value])

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib.nns_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def warmup(device: int = 0) -> None:
    """nns_warmup: touch every kernel family once (code load, LDS opt-in, pool)."""
    _check(lib.nns_warmup(device), "nns_warmup")


def trim() -> int:
    """nns_trim: return the pooled device workspaces to the runtime; bytes released."""
    return int(lib.nns_trim())


def keys_min(inout, other, stream=None) -> None:
    _check(lib.nns_keys_min(inout.data_ptr(), other.data_ptr(), inout.numel(), _stream_ptr(stream)), "nns_keys_min")


def keys_unpack(keys, return_distances: bool = False, stream=None):
    import torch
    m = keys.numel()
    idx = torch.empty(m, dtype=torch.int32, device=keys.device)
    dist = torch.empty(m, dtype=torch.float32, device=keys.device) if return_distances else None
    _check(lib.nns_keys_unpack(keys.data_ptr(), m, idx.data_ptr(),
                               dist.data_ptr() if dist is not None else None, _stream_ptr(stream)),
           "nns_keys_unpack")
    return (idx, dist) if return_distances else idx


def shard_range(n: int, shards: int, rank: int) -> Tuple[int, int]:
    """Contiguous ceil(n/shards) split of the refs, last shard takes the remainder
    (reference core.cu:781-791).  Returns (begin, count); count may be 0."""
    per = -(-n // shards)
    beg = rank * per
    cnt = max(0, min(per, n - beg))
    return beg, cnt


def comm_unique_id() -> bytes:
    """nns_comm_unique_id: the 128 bytes rank 0 hands to every rank (RCCL's ncclUniqueId)."""
    buf = ctypes.create_string_buffer(NNS_COMM_ID_BYTES)
    _check(lib.nns_comm_unique_id(buf, NNS_COMM_ID_BYTES), "nns_comm_unique_id")
    return buf.raw


class Comm:
    """One rank of the one-process-per-GPU exchange (nns_comm): ONE
    ncclAllReduce(ncclUint64, ncclMin) of the packed keys per search, issued by the library —
    the same call site nns_search_f32_multi uses.  Creation is collective over all ranks."""

    def __init__(self, unique_id: bytes, nranks: int, rank: int, device: int):
        h = ctypes.c_void_p()
        _check(lib.nns_comm_create(ctypes.byref(h), unique_id, len(unique_id), nranks, rank, device),
               "nns_comm_create")
        self._h = h
        self.nranks, self.rank, self.device = nranks, rank, device

    def size(self) -> int:
        return lib.nns_comm_size(self._h)

    def allreduce_min(self, keys, stream=None) -> None:
        _check(lib.nns_comm_allreduce_min(self._h, keys.data_ptr(), keys.numel(), _stream_ptr(stream)),
               "nns_comm_allreduce_min")

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib.nns_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multi_last_exchange_ranks() -> int:
    """Ranks of the last grouped RCCL all-reduce a search_multi call completed (0: none / host merge)."""
    return int(lib.nns_multi_last_exchange_ranks())


def shutdown() -> None:
    """nns_shutdown: destroy the cached communicators of search_multi, trim the pool."""
    _check(lib.nns_shutdown(), "nns_shutdown")


def allreduce_min_keys(keys, group=None, comm: Optional[Comm] = None) -> None:
    """The cross-GPU exchange: ONE min all-reduce of the packed keys.  With `comm` the library
    issues it itself (ncclAllReduce(uint64, min) over xGMI, nns_comm_allreduce_min); without,
    it goes through torch.distributed (RCCL when the process group is 'nccl'; gloo in CPU
    tests).  Keys are < 2^63, so the signed int64 order torch reduces in is the unsigned key
    order and both give the same bits."""
    if comm is not None:
        comm.allreduce_min(keys)
        return
    import torch.distributed as dist
    if keys.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a box without one GPU per rank: same operator through the host
        h = keys.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
        keys.copy_(h)
        return
    dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)
