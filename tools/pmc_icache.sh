cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_icache; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/probe_depths.py > $OUT/p1.log 2>&1
echo exit $?
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "filter_kernel" not in row["Kernel_Name"]: continue
        k = (row["Kernel_Name"][25:60], row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
out = collections.defaultdict(dict)
for (kern, cn), (s, c) in acc.items(): out[kern][cn] = s / c
for kern, v in out.items(): print(kern, {a: round(b) for a, b in v.items()})
PY
tail -3 $OUT/p1.log
