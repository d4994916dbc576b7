for i in 1 2; do
  for v in "$@"; do
    NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python bench.py --workload c2 --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c2 step %.2f us exact %.2f us' % (d['ms_per_step']*1e3, d['config']['stage_ms']['exact_ms']*1e3))"
  done
done
