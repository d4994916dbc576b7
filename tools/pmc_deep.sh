#!/bin/bash
# PMC passes over the one-wave-per-SIMD bf16 tiles (512- and 1024-deep): where an MFMA's cycles go.  On the GPU box.
# usage: bash tools/pmc_deep.sh <tag>   (NNS_LIB_PATH selects a variant library)
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_pmc_deep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 tools/probe_depths.py --nw4 > $OUT/pass$i.log 2>&1
  echo "pass $i ($C): exit $?"
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for d in sorted(glob.glob("$OUT/pass*/")):
    for path in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(path)):
            if "filter_kernel" not in row["Kernel_Name"]:
                continue
            k = (row["Kernel_Name"][:90], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kern, cn), (s, c) in acc.items():
            out.setdefault(kern, {})[cn] = s / c
json.dump(out, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
for kern, v in out.items():
    print(kern, {a: round(b, 1) for a, b in v.items()})
PY
