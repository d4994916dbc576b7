mkdir -p gpurun_out/r03n
python -m pytest tests -m gpu -x -q -k "k32_tile or golden or ragged or recipe or filter_sorted or overflow" > gpurun_out/r03n/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03n/tests.log
tail -3 gpurun_out/r03n/tests.log
for i in 1 2; do for v in prod noseedahead; do
  echo "== $v" >> gpurun_out/r03n/ab.txt
  NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_depths.py --k16 2>&1 | grep points >> gpurun_out/r03n/ab.txt
  NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_streams.py 2>&1 | grep "k= 16" >> gpurun_out/r03n/ab.txt
done; done
cat gpurun_out/r03n/ab.txt
