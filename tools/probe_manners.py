#!/usr/bin/env python3
"""Which library call (if any) waits for a kernel the application has running on another stream?
Times every operation of tests/test_gpu_parity.py::test_library_never_waits_for_foreign_streams on its own while a
multi-second spin kernel occupies a non-blocking side stream."""
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
state = {}
def op_small(): pkg.search(*small)
def op_mid(): pkg.search(*mid)
def op_big(): pkg.search(*big)
def op_multi(): pkg.search_multi(mid[0], mid[1], num_devices=2, virtual=True)
def op_create(): state["ix"] = pkg.Index(rd, profile=True)
def op_search(): state["a"] = state["ix"].search(qd)
def op_regrow(): state["ix"].search(qd2)
def op_stats(): state["ix"].stats()
def op_ties(): state["ix"].near_ties()
def op_close(): state["ix"].close()
def op_h2d(): state["t"] = torch.from_numpy(mid[0]).to(dev)
def op_d2h(): state["a"].cpu()
def op_sync_null(): torch.cuda.current_stream().synchronize()
ops = [op_small, op_mid, op_big, op_multi, op_create, op_search, op_regrow, op_stats, op_ties, op_close, op_h2d, op_d2h, op_sync_null]
for o in ops: o()                      # warm
torch.cuda.synchronize()
side = torch.cuda.Stream()
for o in ops:
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        torch.cuda._sleep(int(2.4e9))   # ~1 s
        done.record()
    t0 = time.perf_counter(); o(); dt = time.perf_counter() - t0
    running = not done.query()
    side.synchronize()
    print(f"{o.__name__:14s} {dt*1e3:9.2f} ms  foreign kernel still running afterwards: {running}", flush=True)
