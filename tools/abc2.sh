# same-device A/B of library variants on C2 (bench step + exact-path HIP-event time): tools/abc2.sh name1 name2 ...
for i in 1 2; do
  for v in "$@"; do
    NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python bench.py --workload c2 --steps 300 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c2 step %.2f us exact %.2f us frac %.3f' % (d['ms_per_step']*1e3, d['config']['stage_ms']['exact_ms']*1e3, d['roofline']['frac']))"
  done
done
