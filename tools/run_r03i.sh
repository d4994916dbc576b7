mkdir -p gpurun_out/r03i
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_largest_ref_count_through_the_filter --deselect tests/test_gpu_parity.py::test_largest_ref_count_int32_boundary --deselect tests/test_gpu_parity.py::test_largest_query_count_int32_boundary > gpurun_out/r03i/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03i/tests.log
tail -5 gpurun_out/r03i/tests.log
./tools/ubench/mfma_k4 > gpurun_out/r03i/mfma_k4.txt 2>&1
cat gpurun_out/r03i/mfma_k4.txt
python tools/probe_streams.py 2>&1 | grep "k=" > gpurun_out/r03i/streams.txt
cat gpurun_out/r03i/streams.txt
