"""ctypes access to the CPU oracle (TEST INFRASTRUCTURE — see oracle/v0_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  `ref` (the reference's own V0, oracle/_ref/libv0ref.so) is optional: it is
built only where /root/reference exists and travels to the GPU box as a binary.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libv0oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libv0ref.so")

if not os.path.exists(ORACLE_SO):
    raise ImportError(f"{ORACLE_SO} missing: run `make -C oracle` (or __graft_entry__.build())")

_o = ctypes.CDLL(ORACLE_SO)
_vp, _i, _sz, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64
_o.v0_search.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp]
_o.v0_search_omp.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _i]
_o.v0_search_sharded.argtypes = [_i, _i, _i, _i, _vp, _vp, _vp, _vp]
_o.kdtree_search.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _i]
_o.octree_search.argtypes = [_i, _i, _i, _vp, _vp, _vp, _vp, _i]
_o.v0_pair_distance.argtypes = [_i, _vp, _vp]
_o.v0_pair_distance.restype = ctypes.c_float
_o.ref_recipe_seed.argtypes = [ctypes.c_uint]
_o.ref_recipe_fill.argtypes = [_vp, _sz]
_o.nns_rng_fill.argtypes = [_vp, _sz, _u64, _u64]
_o.nns_round_bf16.argtypes = [_vp, _sz]
_o.nns_fnv1a64.argtypes = [_vp, _sz]
_o.nns_fnv1a64.restype = _u64
_o.fmaf_chain.argtypes = [_i, _vp, _vp, ctypes.c_float]
_o.fmaf_chain.restype = ctypes.c_float

_libc = ctypes.CDLL(None)
_libc.free.argtypes = [_vp]

_ref = ctypes.CDLL(REF_SO) if os.path.exists(REF_SO) else None
if _ref is not None:
    _ref.v0_ref_cudaCall.argtypes = [_i, _i, _i, _vp, _vp, ctypes.POINTER(ctypes.POINTER(_i))]
    _ref.v0_ref_cudaCall.restype = None


def have_reference() -> bool:
    return _ref is not None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def v0_search(q, r, threads: int = 1):
    """(indices int32[m], distances fp32[m]) by the restated V0 (core.cu:31-52)."""
    q, r = _f32(q), _f32(r)
    m, k = q.shape
    n = r.shape[0]
    idx = np.empty(m, np.int32)
    dist = np.empty(m, np.float32)
    if threads > 1:
        _o.v0_search_omp(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data, dist.ctypes.data, threads)
    else:
        _o.v0_search(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data, dist.ctypes.data)
    return idx, dist


def kdtree_search(q, r, threads: int = 1):
    """(indices, V0 distances) by the exact k-d tree comparator (oracle/kdtree.c; finite inputs only)."""
    q, r = _f32(q), _f32(r)
    m, k = q.shape
    n = r.shape[0]
    idx = np.empty(m, np.int32)
    dist = np.empty(m, np.float32)
    rc = _o.kdtree_search(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data, dist.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError(f"kdtree_search failed ({rc})")
    return idx, dist


def octree_search(q, r, threads: int = 1):
    """(indices, V0 distances) by the exact octree comparator (oracle/octree.c; 3-D, finite inputs only)."""
    q, r = _f32(q), _f32(r)
    m, k = q.shape
    n = r.shape[0]
    idx = np.empty(m, np.int32)
    dist = np.empty(m, np.float32)
    rc = _o.octree_search(k, m, n, q.ctypes.data, r.ctypes.data, idx.ctypes.data, dist.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError(f"octree_search failed ({rc})")
    return idx, dist


def v0_search_sharded(q, r, shards: int):
    q, r = _f32(q), _f32(r)
    m, k = q.shape
    n = r.shape[0]
    idx = np.empty(m, np.int32)
    dist = np.empty(m, np.float32)
    _o.v0_search_sharded(k, m, n, shards, q.ctypes.data, r.ctypes.data, idx.ctypes.data, dist.ctypes.data)
    return idx, dist


def v0_reference(q, r):
    """indices by the reference's OWN V0 binary (oracle/_ref); raises if absent."""
    if _ref is None:
        raise RuntimeError("oracle/_ref/libv0ref.so not built (needs /root/reference)")
    q, r = _f32(q), _f32(r)
    m, k = q.shape
    n = r.shape[0]
    res = ctypes.POINTER(_i)()
    _ref.v0_ref_cudaCall(k, m, n, q.ctypes.data, r.ctypes.data, ctypes.byref(res))
    out = np.ctypeslib.as_array(res, shape=(m,)).astype(np.int32, copy=True)
    _libc.free(res)
    return out


def pair_distance(q_row, r_row) -> np.float32:
    q_row, r_row = _f32(q_row), _f32(r_row)
    return np.float32(_o.v0_pair_distance(q_row.size, q_row.ctypes.data, r_row.ctypes.data))


def ref_recipe(samples, seed: int = 1000):
    """The reference driver's data recipe (main.cu:10-13, 27-34, 54, 64): srand(seed)
    once, then per sample queries first, then refs, float(rand()/double(RAND_MAX)).
    Yields (k, m, n, q[m][k], r[n][k]) in table order (glibc rand stream)."""
    _o.ref_recipe_seed(seed)
    for (k, m, n) in samples:
        q = np.empty((m, k), np.float32)
        r = np.empty((n, k), np.float32)
        _o.ref_recipe_fill(q.ctypes.data, q.size)
        _o.ref_recipe_fill(r.ctypes.data, r.size)
        yield k, m, n, q, r


def rng_uniform(count: int, seed: int, offset: int = 0) -> np.ndarray:
    out = np.empty(count, np.float32)
    _o.nns_rng_fill(out.ctypes.data, count, seed, offset)
    return out


def round_bf16(a) -> np.ndarray:
    a = _f32(a).copy()
    _o.nns_round_bf16(a.ctypes.data, a.size)
    return a


def fnv1a64(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return int(_o.nns_fnv1a64(a.ctypes.data, a.nbytes))


def fmaf_chain(a, b, c0: float) -> np.float32:
    """c = fmaf(a[t], b[t], c) for t ascending — the arithmetic v_mfma_f32_32x32x2_f32
    is documented to perform along k (used to check the filter's error model)."""
    a, b = _f32(a), _f32(b)
    return np.float32(_o.fmaf_chain(a.size, a.ctypes.data, b.ctypes.data, ctypes.c_float(c0)))
