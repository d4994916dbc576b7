#!/usr/bin/env python3
"""The exact operation sequence of test_library_never_waits_for_foreign_streams under ONE long foreign kernel,
with a timestamp after every operation: which one (if any) returns only when the foreign kernel has finished?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(58)
small = (rng.random((64, 16), dtype=np.float32), rng.random((1024, 16), dtype=np.float32))
mid = (rng.random((300, 64), dtype=np.float32), rng.random((20000, 64), dtype=np.float32))
big = (rng.random((2048, 128), dtype=np.float32), rng.random((65536, 128), dtype=np.float32))
qd, rd = torch.from_numpy(mid[0]).to(dev), torch.from_numpy(mid[1]).to(dev)
qd2 = torch.from_numpy(rng.random((3000, 64), dtype=np.float32)).to(dev)
T = []
def mark(name):
    T.append((name, time.perf_counter()))
def work():
    mark("begin")
    pkg.search(*small); mark("small")
    pkg.search(*mid); mark("mid")
    pkg.search(*big); mark("big")
    pkg.search_multi(mid[0], mid[1], num_devices=2, virtual=True); mark("multi")
    ix = pkg.Index(rd, profile=True); mark("create")
    a = ix.search(qd); mark("search")
    ix.search(qd2); mark("regrow")
    ix.stats(); mark("stats")
    ix.near_ties(); mark("ties")
    ix.close(); mark("close")
    ix2 = pkg.Index(rd); mark("create2")
    b = ix2.search(qd); mark("search2")
    ix2.close(); mark("close2")
    torch.cuda.current_stream().synchronize(); mark("sync_null")
    a.cpu(); b.cpu(); mark("d2h")
work()
torch.cuda.synchronize()
side = torch.cuda.Stream()
for secs in (1.0, 5.0):
    T.clear()
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        torch.cuda._sleep(int(2.4e9 * secs))
        done.record()
    work()
    running = not done.query()
    side.synchronize()
    print(f"foreign kernel of ~{secs} s: still running after the sequence: {running}")
    for (n0, t0), (n1, t1) in zip(T, T[1:]):
        print(f"   {n1:10s} {1e3 * (t1 - t0):9.2f} ms")
