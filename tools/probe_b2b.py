import sys, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as graft
pkg = graft.load_package()
for (m, n) in [(1024, 1048576), (16384, 65536)]:
    q = torch.empty((m, 128), dtype=torch.float32, device="cuda"); r = torch.empty((n, 128), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
    ix = pkg.Index(r, path="mfma", profile=True)
    for _ in range(5): ix.search_keys(q)
    torch.cuda.synchronize(); ix.stats()
    for _ in range(32): ix.search_keys(q)
    torch.cuda.synchronize(); st = ix.stats()
    tf = 2.0 * m * n * 128 / (st["filter_ms"] * 1e-3) / 1e12
    print(m, n, "back-to-back: filter %.3f ms = %.1f TF (%.1f %%), total %.3f" % (st["filter_ms"], tf, tf / 1.573, st["total_ms"]))
