#!/bin/bash
# rocprofv3 evidence for the HBM-bound few-query shapes (main.cu:39-42 and 1 x 1 M x 16 / x 128):
# kernel-trace stats (average kernel duration) and, in separate passes, FETCH_SIZE / WRITE_SIZE.  On the GPU box.
# usage: bash tools/collect_k1b.sh <tag>
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_k1b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in "16 1 65536" "16 1 1048576" "128 1 1048576" "3 1 1048576" "3 1 65536" "16 4 1048576"; do
  N=$(echo $S | tr ' ' x)
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$N -- python3 tools/run_fewq.py $S 50 > $OUT/stats_$N.log 2>&1
  cp $OUT/stats_$N/*/*kernel_stats.csv $OUT/kernel_stats_$N.csv 2>/dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${N}_$C -- python3 tools/run_fewq.py $S 10 > $OUT/pmc_${N}_$C.log 2>&1
  done
  echo "shape $S done"
done
python3 - <<PY
import csv, glob, json, collections, os
out = {}
for shape in ("16x1x65536", "16x1x1048576", "128x1x1048576", "3x1x1048576", "3x1x65536", "16x4x1048576"):
    k, m, n = (int(v) for v in shape.split("x"))
    ent = {"algorithmic_bytes": n * k * 4 + m * k * 4 + m * 8}
    st = "$OUT/kernel_stats_%s.csv" % shape
    if os.path.exists(st):
        for row in csv.DictReader(open(st)):
            if "exact_" in row["Name"]:
                ent.setdefault("kernels", {})[row["Name"][:70]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for path in glob.glob("$OUT/pmc_%s_%s/**/*counter_collection.csv" % (shape, cn), recursive=True):
            for row in csv.DictReader(open(path)):
                if "exact_" in row["Kernel_Name"] and row["Counter_Name"] == cn:
                    acc[row["Kernel_Name"][:70]][0] += float(row["Counter_Value"]); acc[row["Kernel_Name"][:70]][1] += 1
        for kern, (s, c) in acc.items():
            v = s / c * 1024
            ent.setdefault("pmc", {}).setdefault(kern, {})[cn + "_bytes"] = v * (2 if cn == "FETCH_SIZE" else 1)   # gfx950: FETCH_SIZE x2 (16 B / lane streams)
    for kern, kv in ent.get("kernels", {}).items():
        kv["algorithmic_TBps"] = ent["algorithmic_bytes"] / kv["avg_ns"] / 1e3
    out[shape] = ent
json.dump(out, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
PY
