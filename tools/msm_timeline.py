#!/usr/bin/env python3
"""Timeline of the last MSM call in a rocprofv3 kernel trace (CSV): start / end of every kernel relative to the call's
first kernel, and the queue it ran on - shows what the window-group pipeline overlaps.
    python tools/msm_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = from the last digits_kernel on
last = max(i for i, r in enumerate(rows) if "digits_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
end = 0
prev = ""
for r in rows[last:]:
    if end and "msm::finish_kernel" in prev:
        break
    prev = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    end = max(end, e)
    name = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].replace("void ", "")[:28]
    print("%9.3f %9.3f %8.3f ms  q%-3s %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), name))
print("span %.3f ms" % (end / 1e6))
