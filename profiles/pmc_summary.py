#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per launch, per kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(x in k for x in ("k_fill", "k_sieve", "k_format", "k_error")):
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} launches={len(v):4d} mean={sum(v)/len(v):16.1f}")
