mkdir -p gpurun_out/r03e
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_largest_ref_count_through_the_filter --deselect tests/test_gpu_parity.py::test_largest_ref_count_int32_boundary --deselect tests/test_gpu_parity.py::test_largest_query_count_int32_boundary > gpurun_out/r03e/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03e/tests.log
tail -5 gpurun_out/r03e/tests.log
for i in 1 2; do for v in prod ahead2; do
  echo "== $v" >> gpurun_out/r03e/depths.txt
  NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_depths.py 2>&1 | grep "points" >> gpurun_out/r03e/depths.txt
done; done
sh tools/ab5.sh prod ahead2 > gpurun_out/r03e/ab5.txt 2>&1
for v in prod k1c_nw16w256 k1c_nw16w128; do NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_k1c.py 2>&1 | grep "k=" >> gpurun_out/r03e/k1c.txt; done
NNS_FILTER_CLOCK=1 NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_diag.so python - > gpurun_out/r03e/diag16.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
for (k, m, n) in [(16, 1024, 1048576), (128, 1024, 1048576)]:
    q = torch.empty((m, k), dtype=torch.float32, device="cuda"); r = torch.empty((n, k), dtype=torch.float32, device="cuda")
    pkg.fill_uniform(q, 1, 0); pkg.fill_uniform(r, 2, 0)
    for path in ("auto", "mfma_perref"):
        ix = pkg.Index(r, path=path, profile=True)
        keys = torch.empty(m, dtype=torch.int64, device="cuda")
        print("====", k, m, n, path, flush=True)
        for _ in range(3): ix.search_keys(q, keys)
        torch.cuda.synchronize()
        print(ix.stats(), flush=True)
        ix.close()
PY
cat gpurun_out/r03e/depths.txt gpurun_out/r03e/ab5.txt
