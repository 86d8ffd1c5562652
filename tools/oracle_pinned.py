#!/usr/bin/env python3
"""CPU-side evidence for the pinned free-running gate (tests/test_gpu_parity.py::test_chain_b256_s1000_free_run_pinned_golden):
the CPU ORACLE (a second float32 implementation, torch-CPU) free-running over the reference's B = 256 x 1000 chain with its kNN
choice pinned to the reference's at the same (step, atom) pairs the GPU test pins (tests/golden/chain_b256_s1000_pins.npz).  With
the only discontinuity of the path removed, what is left is the growth of float32 rounding differences along the chain -- the
floor any float32 implementation of this chain sits on.  CPU only, ~15-25 minutes.

    python tools/oracle_pinned.py [--case b256|b1024] [--threads 6] [--thr 5e-4] -> profiles/r04/oracle_pinned_<case>.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import T, golden, oracle_model, synth  # noqa: E402
from oracle import shapemol_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--thr", type=float, default=1.0)
    ap.add_argument("--case", default="b256", choices=["b256", "b1024"], help="b1024: ~1.5 CPU-hours")
    ap.add_argument("--pins", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    a.pins = a.pins or os.path.join(ROOT, "tests", "golden", f"chain_{a.case}_s1000_pins.npz")
    a.out = a.out or os.path.join(ROOT, "profiles", "r04", f"oracle_pinned_{a.case}.json")
    torch.set_num_threads(a.threads)
    c = golden(f"chain_{a.case}_s1000_hash.npz")
    if a.case == "b256":
        ct = golden("chain_b256_s1000_tail_hash.npz")
        tail = (int(ct["first_step"]), int(ct["every"]), ct["pos_traj_tail"])
    else:
        tail = (int(c["tail_first"]), int(c["tail_every"]), c["pos_traj_tail"])
    pins = np.load(a.pins)
    keep = pins["margin"] < a.thr
    p_step, p_atom, p_nbr = pins["step"][keep].astype(np.int64), pins["atom"][keep].astype(np.int64), pins["nbr"][keep].astype(np.int64)
    order = np.argsort(p_step, kind="stable")
    p_step, p_atom, p_nbr = p_step[order], p_atom[order], p_nbr[order]
    B, S, seed, every = int(c["B"]), int(c["S"]), int(c["seed"]), int(c["every"])
    first = np.searchsorted(p_step, np.arange(S + 1))
    bb = synth.synthetic_batch(B, seed=seed, max_atoms=38)
    n = len(bb["batch"])
    off = np.concatenate([[0], np.cumsum(bb["counts"])])
    sd, dm, _, _ = oracle_model()
    real_knn, calls, changed = O.knn_edges, [0], [0]

    def pinned_knn(x, batch, k):
        src, dst = real_knn(x, batch, k)
        s = calls[0]
        calls[0] += 1
        lo, hi = first[s], first[s + 1]
        if hi > lo:
            src = src.clone()
            # edges are grouped by centre, k per atom here (every molecule has more than k atoms)
            assert src.numel() == n * k
            for e in range(lo, hi):
                i = int(p_atom[e])
                new = torch.from_numpy(p_nbr[e])
                if not torch.equal(torch.sort(src[i * k:(i + 1) * k])[0], torch.sort(new)[0]):
                    changed[0] += 1
                src[i * k:(i + 1) * k] = new
        return src, dst

    O.knn_edges = pinned_knn
    t0 = time.time()
    try:
        r = O.sample_chain(sd, dm, T(bb["init_pos"]), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]), S,
                           lambda s: synth.step_noise(n, 15, s, seed=seed), keep_traj=True)
    finally:
        O.knn_edges = real_knn
    pos_traj = torch.stack(r["pos_traj"]).numpy()
    v_traj = torch.stack(r["v_traj"]).numpy()
    mol = lambda e: np.array([e[off[b]:off[b + 1]].max() for b in range(B)])  # noqa: E731
    out = {"what": "CPU oracle (float32), free-running, kNN pinned to the reference's where its choice is fragile", "pins": int(keep.sum()), "thr": a.thr,
           "neighbour_sets_the_pins_changed": changed[0], "B": B, "steps": S, "n_atoms": n, "threads": a.threads, "seconds": round(time.time() - t0, 1), "per_snapshot": []}
    for j in range((S - 1) // every + 1):
        e = mol(np.abs(pos_traj[j * every].astype(np.float64) - c["pos_traj_sub"][j]).max(-1))
        out["per_snapshot"].append({"step": j * every, "max": float(e.max()), "median": float(np.median(e)), "n_over_1e-4": int((e > 1e-4).sum()),
                                    "atom_type_mismatches": int((v_traj[j * every] != c["v_traj_sub"][j]).sum())})
    f0, ev = tail[0], tail[1]
    for i in range(len(tail[2])):
        e = mol(np.abs(pos_traj[f0 + i * ev].astype(np.float64) - tail[2][i]).max(-1))
        out["per_snapshot"].append({"step": f0 + i * ev, "max": float(e.max()), "median": float(np.median(e)), "n_over_1e-4": int((e > 1e-4).sum())})
    e = mol(np.abs(r["pos"].numpy().astype(np.float64) - c["pos"]).max(-1))
    out["end"] = {"step": S - 1, "max": float(e.max()), "median": float(np.median(e)), "n_over_1e-4": int((e > 1e-4).sum()),
                  "atom_type_mismatches": int((r["v"].numpy() != c["v"]).sum()), "worst_molecules": np.argsort(-e)[:8].tolist(),
                  "their_errors": [float(x) for x in np.sort(e)[::-1][:8]]}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out["end"]), flush=True)
    print(json.dumps(out["per_snapshot"][-8:]), flush=True)


if __name__ == "__main__":
    main()
