# K1f at C2: which kernel runs and how long (rocprofv3 kernel stats)
mkdir -p gpurun_out/r03z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03z/c2 -- python3 bench.py --workload c2 --steps 200 --warmup 20 --no-also --no-cpu-baseline > gpurun_out/r03z/c2.log 2>&1
python3 - <<'PY'
import csv, glob
for p in glob.glob("gpurun_out/r03z/c2/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(p)))[:6]:
        print(row["Name"][:70], row["Calls"], row["AverageNs"], row["MinNs"], row["MaxNs"])
PY
