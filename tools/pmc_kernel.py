#!/usr/bin/env python3
"""Averages every rocprofv3 --pmc counter over the dispatches of kernels whose name contains <substring>.
usage: pmc_kernel.py <dir-with-*_counter_collection.csv> <substring>"""
import csv, glob, collections, os, sys
acc = collections.defaultdict(float); n = collections.Counter(); dur = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Counter_Name"], r["Dispatch_Id"]) not in seen:
                seen.add((r["Counter_Name"], r["Dispatch_Id"])); n[r["Counter_Name"]] += 1
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    print(f"{k:32s} {acc[k] / max(n[k], 1):16.1f}   ({n[k]} dispatches)")
if dur:
    print(f"{'avg duration under PMC [us]':32s} {sum(dur) / len(dur):16.1f}")
