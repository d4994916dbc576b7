#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one counter set per pass) into profiles/*.json.

usage: pmc_traffic.py <out.json> <key> <kernel-substring> <pass_dir> [<pass_dir> ...]

Per MI355X_MICROARCH.md (HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE come
from separate passes (TCC slots), are in KiB, and on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced read stream -> doubled here.  Values
are averaged per launch of the named kernel.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out_path, key, needle = sys.argv[1:4]
    sums = defaultdict(float)
    counts = defaultdict(int)
    for d in sys.argv[4:]:
        # gpurun merges results into the same local directory call after call: only the newest
        # collection of each pass counts
        found = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        for path in found[-1:]:
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if needle not in row.get("Kernel_Name", ""):
                        continue
                    name = row["Counter_Name"]
                    sums[name] += float(row["Counter_Value"])
                    counts[name] += 1
    per_launch = {n: sums[n] / counts[n] for n in sums}
    res = {"kernel": needle, "launches": {n: counts[n] for n in counts}, "raw_per_launch": per_launch}
    if "FETCH_SIZE" in per_launch:
        res["fetch_bytes"] = per_launch["FETCH_SIZE"] * 1024 * 2   # gfx950 x2 correction (16 B/lane streams)
    if "WRITE_SIZE" in per_launch:
        res["write_bytes"] = per_launch["WRITE_SIZE"] * 1024
    if "fetch_bytes" in res:
        res["hbm_bytes"] = res["fetch_bytes"] + res.get("write_bytes", 0.0)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per_launch and "GRBM_GUI_ACTIVE" in per_launch:
        res["mfma_busy_per_gui_active"] = per_launch["SQ_VALU_MFMA_BUSY_CYCLES"] / per_launch["GRBM_GUI_ACTIVE"]
    data = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            data = json.load(f)
    data[key] = res
    with open(out_path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
