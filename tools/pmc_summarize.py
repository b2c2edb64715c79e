#!/usr/bin/env python3
"""Fold the rocprofv3 outputs of tools/profile_workload.sh into one JSON summary (per-launch averages of
every counter for the kernels whose name contains MATCH, plus the kernel-trace average duration).

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reads
exactly half the bytes of a WIDE COALESCED STREAMING read (128-byte requests tallied at 64 B) and "other access widths
are uncalibrated: calibrate on a known byte count in your own access pattern".  Pass the known read volume of the kernel
as the third argument (bytes per launch) and the summary says which rule fits; for the scalar-multiplication kernels -
lane-divergent 64-byte table blocks, 61.9 of them per unit - the raw figure matches to ~1 %, so `hbm_bytes_per_launch`
is raw FETCH + WRITE and the x2 figure is kept beside it as `hbm_bytes_per_launch_x2_streaming_rule`.
"""
import csv
import glob
import json
import os
import sys


def main():
    out, match = sys.argv[1], sys.argv[2]
    known_read = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] not in ("", "-") else None
    workload = sys.argv[4] if len(sys.argv) > 4 else None
    commit = sys.argv[5] if len(sys.argv) > 5 else None
    per = {}
    clocks = []
    for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc, span = {}, {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if match not in row["Kernel_Name"]:
                    continue
                key = (row["Counter_Name"], row["Dispatch_Id"])
                acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
                # effective clock of THIS dispatch in THIS pass: the counter sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp") and float(row["End_Timestamp"]) > float(row["Start_Timestamp"]):
                    span[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        clocks += [acc[("GRBM_GUI_ACTIVE", d)] / 8.0 / ns for d, ns in span.items()]
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
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    # provenance: which workload of bench.py, the commit the tree was at (passed in: the GPU box has no .git) and the hash of
    # the kernel sources that were profiled - bench.py recomputes the hash and says whether the sources have changed since
    res = {"workload": workload, "commit": commit, "csrc_sha16": bench.csrc_sha16(workload), "kernel_match": match, "counters_per_launch": per, "kernel_trace": dur}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        raw = (per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
        x2 = (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
        res["hbm_bytes_per_launch_x2_streaming_rule"] = x2
        res["hbm_bytes_per_launch"] = raw
        if known_read:
            f = per["FETCH_SIZE"] * 1024
            res["fetch_calibration"] = {"known_read_bytes": known_read, "fetch_size_bytes": f, "ratio_raw": f / known_read, "ratio_x2": 2 * f / known_read,
                                        "rule": "raw" if abs(f / known_read - 1) < abs(2 * f / known_read - 1) else "x2"}
            if res["fetch_calibration"]["rule"] == "x2":
                res["hbm_bytes_per_launch"] = x2
    if clocks:
        res["effective_clock_ghz"] = sum(clocks) / len(clocks)
    if "TCC_HIT_sum" in per:
        res["l2_hit_rate"] = per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
