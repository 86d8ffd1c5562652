#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0][-48:]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in rows for c in rows[k]})
print("kernel".ljust(50), *[n[-22:].rjust(23) for n in names])
for k, d in rows.items():
    print(k.ljust(50), *[f"{sum(d[n]) / max(1, len(d[n])):23.0f}" for n in names], f" n={len(next(iter(d.values())))}")
