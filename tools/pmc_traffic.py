#!/usr/bin/env python3
"""Per-kernel-class HBM traffic from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE):
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [lib_sha16] > pmc_traffic.json
Counters are in KB per dispatch (summed over the XCDs by rocprofv3); mean per launch per class."""
import collections, csv, json, re, sys

CLASSES = [("edge_fused_kernel", "false", "edge_x2h"), ("edge_fused_kernel", "true", "edge_h2x"),
           ("node_chain16_kernel", "", "node_chain"), ("node_chain6_kernel", "", "node_chain"), ("node_chain_kernel", "", "node_chain"),
           ("node_linear16_kernel", "", "node_pre"), ("node_linear6_kernel", "", "node_pre"), ("node_linear_kernel", "", "node_pre"),
           ("node_prologue16_kernel", "", "node_prologue"), ("node_prologue6_kernel", "", "node_prologue"), ("vn_stats_kernel", "", "vn_stats"),
           ("vn_apply_kernel", "", "vn_apply"), ("knn_kernel", "", "knn"), ("edge_weight_kernel", "", "edge_weight"),
           ("graph_kernel", "", "graph"), ("combine32_kernel", "", "edge_combine"), ("ddpm_step", "", "ddpm")]


def classify(name):
    if "x2h_chain16_kernel" in name:
        return "edge_x2h_chain"
    m = re.search(r"edge(?:_fused|16|16_loop|_stream)_kernel<\d+, \d+, (false|true)", name)      # third template argument: H2X
    if m:
        return "edge_h2x" if m.group(1) == "true" else "edge_x2h"
    for key, flag, cls in CLASSES:
        if key in name and not flag:
            return cls
    return None


def mean_per_class(path, counter):
    acc = collections.defaultdict(list)
    per_dispatch = collections.defaultdict(float)
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            key = r.get("Dispatch_Id") or r.get("Correlation_Id")
            per_dispatch[key] += float(r["Counter_Value"])
            meta[key] = r["Kernel_Name"]
    for key, val in per_dispatch.items():
        cls = classify(meta[key])
        if cls:
            acc[cls].append(val)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, n = mean_per_class(sys.argv[1], "FETCH_SIZE")
write, _ = mean_per_class(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager launches); mean per "
                "launch; counters are in KB (x1024 = bytes). Per MI355X_MICROARCH.md the gfx950 FETCH_SIZE under-reports wide "
                "coalesced reads by 2x; the gather pattern here is uncalibrated, so both the raw and the doubled figure are "
                "given; hbm_bytes_per_launch = 2 x fetch + write.",
       "lib_sha16": (sys.argv[3] if len(sys.argv) > 3 else None),      # the library build these were measured on (bench.py nulls `traffic` for any other)
       "kernels": {}}
for k in fetch:
    fb, wb = fetch[k] * 1024, write.get(k, 0.0) * 1024
    out["kernels"][k] = {"fetch_bytes_raw": int(fb), "fetch_bytes_x2": int(2 * fb), "write_bytes": int(wb),
                         "hbm_bytes_per_launch": int(2 * fb + wb), "launches_sampled": n[k]}
json.dump(out, sys.stdout, indent=1)
print()
