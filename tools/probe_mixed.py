import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as graft
pkg, orc = graft.load_package(), graft.load_oracle()
rng = np.random.default_rng(3)
for (m, n, k) in [(300, 5000, 128), (700, 20001, 200), (1000, 70000, 64), (130, 3000, 33), (2049, 777, 256)]:
    q = rng.random((m, k), dtype=np.float32); r = rng.random((n, k), dtype=np.float32)
    r[n // 2: n // 2 + 40] = r[:40]
    want = orc.v0_search(q, r, threads=16)
    for sh in (1, 3):
        idx, dist = pkg.search(q, r, return_distances=True, shards=sh, path="mfma", filter_bf16=True)
        ok = np.array_equal(idx, want[0]) and np.array_equal(dist.view(np.uint32), want[1].view(np.uint32))
        print((m, n, k), "shards", sh, "OK" if ok else "MISMATCH", flush=True)
# C3 timing
m, n, k = 65536, 1048576, 128
q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
pkg.fill_uniform(q, 1000, 0); pkg.fill_uniform(r, 1000, m * k)
ix0 = pkg.Index(r, profile=True); k0 = ix0.search_keys(q).clone(); torch.cuda.synchronize(); print("fp32 filter", ix0.stats()); ix0.close()
ix = pkg.Index(r, profile=True, filter_bf16=True)
for _ in range(3):
    k1 = ix.search_keys(q); torch.cuda.synchronize()
    st = ix.stats()
print("bf16 filter", st)
print("keys equal:", bool(torch.equal(k0, k1)))
