#!/usr/bin/env python
"""Aggregates rocprofv3 --pmc counter_collection CSVs: mean of every counter per kernel name."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")
        short = k.split("(")[0][-70:]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    if "conv_gemm" not in k and len(sys.argv) < 3:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:36s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
