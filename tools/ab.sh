# A/B of filter builds on one device: tools/ab.sh name1 name2 ...
for i in 1 2; do
  for v in "$@"; do
    NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_filter.py --reps 3 2>&1 | grep "run " | sed "s/^/$v /" | cut -c1-80
  done
done
