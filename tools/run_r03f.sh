mkdir -p gpurun_out/r03f
bash tools/collect_k1b.sh r03 > gpurun_out/r03f/k1b_collect.log 2>&1
# deep-tile ablations: what the one-wave-per-SIMD tiles lose to DMA issue (16), to the ring sync (1), to both (17)
for v in diag abl16 abl1 abl17; do
  echo "== $v" >> gpurun_out/r03f/deep_ablate.txt
  NNS_DIAG_FILTER_ONLY=1 NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_depths.py --nw4 2>&1 | grep "points" >> gpurun_out/r03f/deep_ablate.txt
done
# K1a: relaxed counter vs acq_rel counter (same device)
sh tools/abc2.sh prod k1acq > gpurun_out/r03f/ab_k1a_counter.txt 2>&1
cat gpurun_out/r03f/deep_ablate.txt gpurun_out/r03f/ab_k1a_counter.txt
tail -60 gpurun_out/r03f/k1b_collect.log
