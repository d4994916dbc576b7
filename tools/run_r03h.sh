mkdir -p gpurun_out/r03h
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_largest_ref_count_through_the_filter --deselect tests/test_gpu_parity.py::test_largest_ref_count_int32_boundary --deselect tests/test_gpu_parity.py::test_largest_query_count_int32_boundary > gpurun_out/r03h/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03h/tests.log
tail -5 gpurun_out/r03h/tests.log
for i in 1 2; do for v in prod notop2; do
  echo "== $v" >> gpurun_out/r03h/streams.txt
  NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_streams.py --c3 2>&1 | grep "k=" >> gpurun_out/r03h/streams.txt
done; done
cat gpurun_out/r03h/streams.txt
