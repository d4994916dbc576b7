mkdir -p gpurun_out/r03g
python -m pytest tests -m gpu -x -q -k "k1024 or error_model or rounding_model or deep_tile or dimensionality_beyond or k512" > gpurun_out/r03g/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03g/tests.log
tail -5 gpurun_out/r03g/tests.log
for v in prod k768nw4; do
  echo "== $v" >> gpurun_out/r03g/deep.txt
  NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_depths.py --deep 2>&1 | grep -E "points|exact" >> gpurun_out/r03g/deep.txt
done
for v in negbase negabl16; do
  echo "== $v" >> gpurun_out/r03g/deep_ablate.txt
  NNS_DIAG_FILTER_ONLY=1 NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_depths.py --nw4 2>&1 | grep "points" >> gpurun_out/r03g/deep_ablate.txt
done
cat gpurun_out/r03g/deep.txt gpurun_out/r03g/deep_ablate.txt
