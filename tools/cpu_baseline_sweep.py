#!/usr/bin/env python3
"""How many threads should the CPU baseline of bench.py use on the GPU box?  Times the CPU oracle on the B = 256 workload at
8 / 16 / 32 / 64 threads (3 reverse steps each after a warm-up step) and prints one JSON object.
    python tools/cpu_baseline_sweep.py > profiles/r03/cpu_baseline_sweep.json"""
import json, os, sys
import yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from shapemol_amd import synth
cfg = yaml.safe_load(open(bench.TRAIN_YML))["model"]
bb = synth.synthetic_batch(256, seed=2021)
out = {"os_cpu_count": os.cpu_count(), "cpu_affinity": len(os.sched_getaffinity(0)), "workload": "B=256 (5541 atoms), 3 reverse steps", "runs": []}
for th in (8, 16, 32, 64):
    dt, used = bench.cpu_baseline(cfg, bb, 3, th)
    out["runs"].append({"threads": used, "s_per_step": round(dt, 4), "molecules_per_s": round(256 / (1000 * dt), 4)})
    print(out["runs"][-1], file=sys.stderr, flush=True)
out["fastest_threads"] = min(out["runs"], key=lambda r: r["s_per_step"])["threads"]
print(json.dumps(out, indent=1))
