#!/usr/bin/env python3
"""Fold the counter CSV of tools/ct_evidence.py: per kernel, the counters of its consecutive dispatches (scalar sets
"all ones", "all sevens", random).  python tools/ct_summarize.py gpurun_out/ct > profiles/r03_ct_counters.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        rows += list(csv.DictReader(f))
per = defaultdict(lambda: defaultdict(float))       # (dispatch, kernel) -> counter -> value
for r in rows:
    per[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0])][r["Counter_Name"]] += float(r["Counter_Value"])
by_kernel = defaultdict(list)
for (d, k), c in sorted(per.items()):
    by_kernel[k].append((d, c))
for k, lst in by_kernel.items():
    if not any(t in k for t in ("mul_gen_ref", "lincomb_ref", "mul_kernel", "mul_fast", "mul_wide", "mul_ct_kernel", "sign_finish")):
        continue
    names = sorted(lst[0][1])
    # one kernel can serve several experiments of tools/ct_evidence.py (e.g. the secp256k1 reference schedule: k G-multiples of
    # G, then the ECDH sets on random points): dispatches further than 3 ids apart are different experiments
    runs = [[lst[0]]]
    for prev, cur in zip(lst, lst[1:]):
        if cur[0] - prev[0] > 3:
            runs.append([])
        runs[-1].append(cur)
    print(k)
    for run in runs:
        for d, c in run:
            print("   dispatch %4d  " % d + "  ".join("%s=%d" % (nm, c[nm]) for nm in names))
        same = all(all(c[nm] == run[0][1][nm] for nm in names) for _, c in run)
        spread = max((max(c[nm] for _, c in run) - min(c[nm] for _, c in run)) / max(1.0, max(c[nm] for _, c in run)) for nm in names)
        print("   -> %d scalar sets, counters identical: %s%s" % (len(run), same, "" if same else "  (largest relative spread %.1e)" % spread))
