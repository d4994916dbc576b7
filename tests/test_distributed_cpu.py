"""world_size-2 gloo test of the N>1 path: per-shard packed keys -> ONE min
all-reduce -> V0's answer.  The per-shard keys come from the ORACLE here (no GPU in
this container); the exchange + key algebra are the product's (host.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, r, want_idx, want_dist_bits, ret):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    orc = graft.load_oracle()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = r.shape[0]
    beg, cnt = pkg.shard_range(n, world, rank)
    if cnt > 0:
        li, ld = orc.v0_search(q, r[beg:beg + cnt])
        bits = ld.view(np.uint32).astype(np.int64)
        keys = (bits << 32) | (li.astype(np.int64) + beg)
        keys[np.isinf(ld)] = pkg.NNS_KEY_NONE
    else:
        keys = np.full(q.shape[0], pkg.NNS_KEY_NONE, np.int64)
    t = torch.from_numpy(keys)
    pkg.allreduce_min_keys(t)
    got = t.numpy()
    idx = (got & 0xFFFFFFFF).astype(np.int32)
    dbits = (got >> 32).astype(np.uint32)
    ok = np.array_equal(idx, want_idx) and np.array_equal(dbits, want_dist_bits)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def _run(world, q, r):
    graft = __import__("__graft_entry__")
    orc = graft.load_oracle()
    with np.errstate(all="ignore"):
        want_idx, want_dist = orc.v0_search(q, r)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), q, r, want_idx, want_dist.view(np.uint32), ret),
             nprocs=world, join=True)
    assert all(ret[i] for i in range(world)), dict(ret)


def test_two_rank_key_allreduce_matches_v0(built):
    rng = np.random.default_rng(5)
    q = rng.random((50, 16), dtype=np.float32)
    r = rng.random((2001, 16), dtype=np.float32)
    r[1500] = r[20]          # cross-shard exact tie: the lower (rank-0) index must win
    q[0] = r[20]
    q[1, 0] = np.nan         # no selectable distance on any rank -> index 0
    _run(2, q, r)


def test_more_ranks_than_refs(built):
    rng = np.random.default_rng(6)
    q = rng.random((5, 3), dtype=np.float32)
    r = rng.random((1, 3), dtype=np.float32)
    _run(2, q, r)
