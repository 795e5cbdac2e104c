#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            short = name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
            if "anonymous" in name and "at::native" not in name:
                short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "at::native" in k or "rocclr" in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:24s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
