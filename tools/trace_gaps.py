#!/usr/bin/env python3
"""Per-step kernel-time vs wall-time from a rocprofv3 kernel_trace.csv (graph-replay run)."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# steps are delimited by ddpm kernels
steps, cur = [], []
for s, e, n in rows:
    cur.append((s, e, n))
    if n.startswith("ddpm_step"):
        steps.append(cur); cur = []
steps = steps[20:-5]
busy = [sum(e - s for s, e, _ in st) for st in steps]
wall = [st[-1][1] - st[0][0] for st in steps]
gaps = [sum(max(0, st[i + 1][0] - st[i][1]) for i in range(len(st) - 1)) for st in steps]
import statistics as S
print(f"steps analysed: {len(steps)}, kernels/step: {S.median(len(st) for st in steps)}")
print(f"median per step: wall(first start..last end) {S.median(wall)/1e3:.1f} us, sum kernel durations {S.median(busy)/1e3:.1f} us, sum idle gaps {S.median(gaps)/1e3:.1f} us")
per = collections.defaultdict(list)
for st in steps:
    acc = collections.Counter()
    for s, e, n in st: acc[n] += e - s
    for n, v in acc.items(): per[n].append(v)
# kernels that run in every analysed step (per-chain preparation kernels and copies appear in a few steps only)
for n, v in sorted(per.items(), key=lambda kv: -S.median(kv[1])):
    if len(v) * 2 < len(steps):
        continue
    print(f"  {n[-46:]:46s} {S.median(v)/1e3:8.1f} us/step")
