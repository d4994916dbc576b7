# round 3, session 2: where the deep tiles' time goes — ablated filter builds (-DNNS_DIAG -DNNS_FILTER_ABLATE=bits: 1 no ring
# sync, 2 no epilogue, 16 no DMA issue; results are wrong, the filter runs alone) and the L2 counters of the same kernels
mkdir -p gpurun_out/r03u
for v in diag0 diag1 diag2 diag16 diag19; do
  NNS_DIAG_FILTER_ONLY=1 NNS_LIB_PATH=$PWD/nns-cuda_amd/libnns_var_$v.so python tools/probe_depths.py --nw4 2>&1 | grep -E "points" | cut -c1-110 | sed "s/^/$v /" | tee -a gpurun_out/r03u/deep_ablate.txt
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  D=gpurun_out/r03u/$(echo $C | cut -d" " -f1)
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 tools/probe_depths.py --nw4 > $D.log 2>&1
  echo "pmc $C exit $?"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob("gpurun_out/r03u/TCC*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "filter_kernel" not in row["Kernel_Name"]:
            continue
        k = (row["Kernel_Name"][23:60], row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
with open("gpurun_out/r03u/l2_deep.txt", "w") as f:
    for (kern, cn), (s, c) in sorted(acc.items()):
        line = f"{kern} {cn} {s / c:.4g} per launch ({c} launches)"
        print(line); f.write(line + "\n")
PY
