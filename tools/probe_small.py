import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
for (k, m, n) in [(3,1024,1024),(16,1024,1024),(3,1024,4096),(8,1024,4096),(16,4096,4096),(3,4096,65536),(16,512,8192),(2,2048,2048),(3,1024,65536),(3,1024,1048576),(3,16384,16384),(4,8192,32768),(1,4096,4096),(3,100,100000)]:
    q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
    ix = pkg.Index(r, path="exact"); keys = torch.empty(m, dtype=torch.int64, device="cuda")
    for _ in range(5): ix.search_keys(q, keys)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ix.search_keys(q, keys)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"k={k:2d} m={m:5d} n={n:6d}: {dt*1e6:7.1f} us", flush=True); ix.close()
