# round 3, session 2: the 512-deep tile on eight waves (prod) against the four-wave form (k512nw4), tests first
mkdir -p gpurun_out/r03t
python -m pytest tests -m gpu -x -q -k "k512 or error_model or rounding_model or deep_tile or bf16_filter or lane_threshold or overflow" > gpurun_out/r03t/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03t/tests.log
tail -n 4 gpurun_out/r03t/tests.log
for i in 1 2; do
  for v in prod k512nw4; do
    if [ $v = prod ]; then L=""; else L=$PWD/nns-cuda_amd/libnns_var_$v.so; fi
    NNS_LIB_PATH=$L python tools/probe_depths.py --k512 2>&1 | grep -E "points k=512" | sed "s/^/$v /" | tee -a gpurun_out/r03t/k512_ab.txt
  done
done
