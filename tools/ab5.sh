# A/B of filter builds on C5 (bf16): tools/ab5.sh name1 name2 ...
for i in 1 2; do
  for v in "$@"; do
    NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c5 filter %.3f ms frac %.4f' % (d['config']['stage_ms']['filter_ms'], d['roofline']['frac']))"
  done
done
