#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: mean counter value per (kernel, counter)."""
import csv, glob, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mh::" not in k:
            continue
        short = k.split("(")[0].replace("void mh::", "")
        acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[short].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("==", f)
    for (k, c), v in sorted(acc.items()):
        print("  %-45s %-24s n=%-3d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
    for k, v in dur.items():
        print("  %-45s duration_ns mean=%.0f" % (k, sum(v) / len(v)))
