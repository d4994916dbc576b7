#!/bin/bash
# L2 behaviour of the filter: TCC hit/miss/request counters for one workload (c3 | c5)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-c5}
OUT=$R/gpurun_out/l2_$W
mkdir -p $OUT
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  D=$OUT/$(echo $C | cut -d" " -f1)
  ( cd $R && timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $D.log 2>&1 )
  echo "pmc $W $C exit $?"
done
cd $R && python3 tools/pmc_traffic.py $OUT/l2.json $W filter_kernel $OUT/TCC_HIT_sum $OUT/TCC_REQ_sum $OUT/TCC_EA0_RDREQ_sum | tail -30
