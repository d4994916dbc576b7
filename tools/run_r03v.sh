# slow-path statistics + in-kernel clock of the deep tiles (diagnostic build)
mkdir -p gpurun_out/r03v
NNS_FILTER_CLOCK=1 NNS_DIAG_FILTER_ONLY=1 NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_diag0.so python tools/probe_depths.py --nw4 2>&1 | grep -E "points|nns\]" | cut -c1-250 | tail -n 12 | tee gpurun_out/r03v/deep_clock.txt
