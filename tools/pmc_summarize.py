#!/usr/bin/env python3
"""Fold the rocprofv3 outputs of tools/profile_headline.sh into one JSON summary (per-launch averages of
every counter for the kernels whose name contains MATCH, plus the kernel-trace average duration).

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  On gfx950 FETCH_SIZE counts 64-byte requests as
32 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section): both the raw and the corrected (x2
on the fetch side) figures are given.
"""
import csv
import glob
import json
import os
import sys


def main():
    out, match = sys.argv[1], sys.argv[2]
    per = {}
    for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if match not in row["Kernel_Name"]:
                    continue
                key = (row["Counter_Name"], row["Dispatch_Id"])
                acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
        names = {k[0] for k in acc}
        for nm in names:
            vals = [v for (c, _), v in acc.items() if c == nm]
            per[nm] = sum(vals) / len(vals)
    dur = None
    for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if match in row["Name"]:
                    dur = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "pct": float(row["Percentage"])}
    res = {"kernel_match": match, "counters_per_launch": per, "kernel_trace": dur}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        res["hbm_bytes_per_launch_raw"] = (per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
        res["hbm_bytes_per_launch"] = (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in per:
        res["l2_hit_rate"] = per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
