#!/bin/bash
# The clock a kernel actually ran at: GRBM_GUI_ACTIVE (shader-clock cycles the GPU was busy) / the dispatch's duration, for
# every filter launch of a probe (default: tools/probe_streams.py, the 1024-query short streams).  On the GPU box.
# usage: bash tools/pmc_clock.sh <tag> [probe.py args...]
TAG=${1:-r03}; shift
PROBE=${1:-tools/probe_streams.py}; shift
OUT=gpurun_out/${TAG}_clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pass -- python3 $PROBE "$@" > $OUT/pass.log 2>&1
echo "pmc exit $?"
python3 - <<PY
import csv, glob, collections
cnt = {}
for p in glob.glob("$OUT/pass/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[row["Dispatch_Id"]] = (row["Kernel_Name"], float(row["Counter_Value"]))
acc = collections.OrderedDict()
for p in glob.glob("$OUT/pass/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        d = row["Dispatch_Id"]
        if d not in cnt or "filter_kernel" not in cnt[d][0] and "lowdim" not in cnt[d][0] and "exact_lane" not in cnt[d][0]:
            continue
        dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        key = (cnt[d][0][:70], round(dur / 1000, -1 if dur > 2e5 else 0))
        a = acc.setdefault(cnt[d][0][:70] + " wg=" + row.get("Workgroup_Size", "?") + " grid=" + row.get("Grid_Size", "?"), [0.0, 0.0, 0])
        a[0] += cnt[d][1]; a[1] += dur; a[2] += 1
with open("$OUT/clock.txt", "w") as f:
    for k, (c, dur, n) in acc.items():
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        line = f"{k}: {n} launches, {dur / n / 1000:.1f} us each, GRBM_GUI_ACTIVE / 8 / duration = {c / 8 / dur:.3f} GHz"
        print(line); f.write(line + "\n")
PY
