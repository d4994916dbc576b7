#!/bin/bash
# Collect the round's evidence on the GPU box: bench lines, rocprofv3 kernel stats, PMC passes.
# usage (from the repo root, on the GPU box): bash tools/collect_profiles.sh <tag>
# (two parts, each within one gpurun call: part 1 = bench lines, kernel stats, PMC passes; part 2 = the rest)
TAG=${1:-r03}
PART=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$PART" != "2" ]; then
# the driver-shaped default run: C3 headline + the also block (C1, C2, C5)
timeout -k 10 400 python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_default.json
echo "bench default: $(cut -c1-160 $OUT/bench_default.json)"
for W in c3 c2 c5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python3 bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-also > $OUT/stats_$W.log 2>&1
  cp $OUT/stats_$W/*/*kernel_stats.csv $OUT/kernel_stats_$W.csv 2>/dev/null
  echo "stats $W exit $?"
done
for W in c3 c5 c2; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    D=$OUT/pmc_${W}_$(echo $C | cut -d" " -f1)
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-also > $D.log 2>&1
    echo "pmc $W $C exit $?"
  done
done
fi
if [ "$PART" = "1" ]; then ls $OUT; exit 0; fi
# K4 / K3: does the vector work co-execute with the matrix pipe?  (SQ_INSTS_VALU counts MFMAs too: 1.47 "VALU per MFMA"
# on C5 = 0.47 other vector instructions per MFMA.)
for W in c5 c3; do
  D=$OUT/pmc_${W}_COEXEC
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-also > $D.log 2>&1
  echo "pmc $W COEXEC exit $?"
done
# the exact driver command under the profiler (headline + also block in one process)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 bench.py > $OUT/stats_default.log 2>&1
cp $OUT/stats_default/*/*kernel_stats.csv $OUT/kernel_stats_default.csv 2>/dev/null
timeout -k 10 300 ./nns-cuda_amd/nns_driver --repeat 3 > $OUT/driver.txt 2>&1
timeout -k 10 300 python tools/wholecall_c3.py 2>&1 | grep -v amdgpu > $OUT/wholecall_c3.txt
timeout -k 10 300 python tools/probe_depths.py 2>&1 | grep -v amdgpu > $OUT/depths.txt
timeout -k 10 300 python tools/probe_depths.py --deep 2>&1 | grep -v amdgpu > $OUT/depths_deep.txt
timeout -k 10 200 python tools/probe_shapes.py 2>&1 | grep -v amdgpu > $OUT/shapes.txt
timeout -k 10 300 python tools/probe_streams.py 2>&1 | grep -v amdgpu > $OUT/streams.txt
cat $OUT/depths.txt
ls $OUT
