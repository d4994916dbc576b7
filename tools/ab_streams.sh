# same-device A/B of library variants on the short-stream shapes: tools/ab_streams.sh name1 name2 ...
VARS="$@"
for i in 1 2; do
  for v in $VARS; do
    for shape in "1024 1048576 128" "1024 1048576 16" "16384 65536 128"; do
      read M N K <<< "$shape"
      NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_filter.py --m $M --n $N --k $K --reps 5 2>&1 | grep "run " | sed "s/^/$v ${M}x${N}x${K} /" | cut -c1-110
    done
  done
done
