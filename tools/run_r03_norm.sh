# norm pieces by one wave per slot (prod) against every wave (normall = -DNNS_F_NORM_ONE=0): parity first, then same-device A/B
mkdir -p gpurun_out/r03n
python -m pytest tests -m gpu -x -q -k "filter or headline or c4 or c5 or bf16 or k512 or k1024 or deep_tile or k32 or k64 or k256 or sorted or overflow or ragged or short_randomised or lane_threshold or error_model" > gpurun_out/r03n/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03n/tests.log
tail -n 4 gpurun_out/r03n/tests.log
bash tools/ab_streams.sh prod normall 2>&1 | tee gpurun_out/r03n/ab_streams.txt
bash tools/ab.sh prod normall 2>&1 | tee gpurun_out/r03n/ab_c3_2.txt
for i in 1 2; do for v in prod normall; do NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_depths.py --k16 2>&1 | grep -E "points" | sed "s/^/$v /" | cut -c1-120 | tee -a gpurun_out/r03n/ab_k16.txt; done; done
