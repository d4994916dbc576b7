# fp32 operators: a slot's ring DMA pieces issued back to back (prod) against one piece per step (noburst = -DNNS_F_DMA_BURST=0): parity, then A/B
mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -x -q -k "filter or headline or c4 or k32 or k64 or k256 or sorted or overflow or ragged or short_randomised or golden or midsize or offset" > gpurun_out/r03b/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03b/tests.log
tail -n 3 gpurun_out/r03b/tests.log
bash tools/ab.sh prod noburst 2>&1 | tee gpurun_out/r03b/ab_c3_2.txt
for i in 1 2; do for v in prod noburst; do
  NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_depths.py 2>&1 | grep -E "f32  points k=( 16| 32| 64|128|256)" | sed "s/^/$v /" | cut -c1-125 | tee -a gpurun_out/r03b/ab_f32_depths.txt
done; done
for i in 1 2; do for v in prod noburst; do
  NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_streams.py 2>&1 | grep -E "^k=" | grep -v "k=  3" | sed "s/^/$v /" | cut -c1-140 | tee -a gpurun_out/r03b/ab_streams.txt
done; done
