#!/bin/bash
# PMC passes over the C2 bench (exact VALU path): instruction mix, waits, HBM traffic.  On the GPU box.
# usage: bash tools/pmc_c2.sh <tag>
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_pmc_c2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 bench.py --workload c2 --steps 20 --warmup 2 --no-cpu-baseline --no-also > $OUT/pass$i.log 2>&1
  echo "pass $i ($C): exit $?"
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for d in sorted(glob.glob("$OUT/pass*/")):
    for path in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(path)):
            k = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kern, cn), (s, c) in acc.items():
            out.setdefault(kern, {})[cn] = s / c
json.dump(out, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
for kern, v in out.items():
    print(kern, {a: round(b, 1) for a, b in v.items()})
PY
