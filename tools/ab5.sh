# same-device A/B of library variants on C5 (bf16): tools/ab5.sh name1 name2 ...
for i in 1 2 3; do
  for v in "$@"; do
    NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c5 step %.2f ms filter %.2f ms frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done
