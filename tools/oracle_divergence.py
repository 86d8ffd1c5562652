#!/usr/bin/env python3
"""Does a SECOND float32 implementation on the CPU hold the reference's trajectory over a full-length chain?  (CPU only.)

The CPU oracle (torch-CPU float32, within 5e-6 per forward of the reference) runs free over the reference's
B = 256 x 1000 chain (tests/golden/chain_b256_s1000_hash.npz, same hash noise) and is compared with the reference's
snapshots: per snapshot the max / median molecule error and the number of molecules beyond 1e-4, per diverged molecule the
first snapshot beyond 1e-4 and the smallest relative gap between its k-th and (k+1)-th neighbour distances on the way.
This is the CPU-side evidence for the windowed full-length gate: kNN selection is discontinuous, so
no second float32 implementation stays within 1e-4 over the free run, whatever the device.

    python tools/oracle_divergence.py [--threads 6] [--steps 1000] -> profiles/r03/oracle_divergence_b256.json   (~11-20 CPU-minutes)
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
from tools_knn import knn_margin_rel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03", "oracle_divergence_b256.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    c = golden("chain_b256_s1000_hash.npz")
    B, S, seed, every = int(c["B"]), min(int(c["S"]), a.steps), int(c["seed"]), int(c["every"])
    bb = synth.synthetic_batch(B, seed=seed, max_atoms=38)
    n = len(bb["batch"])
    off = np.concatenate([[0], np.cumsum(bb["counts"])])
    sd, dm, _, _ = oracle_model()
    t0 = time.time()
    r = O.sample_chain(sd, dm, T(bb["init_pos"]), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]), S,
                       lambda s: synth.step_noise(n, 15, s, seed=seed), keep_traj=True)
    pos_traj = torch.stack(r["pos_traj"]).numpy()
    v_traj = torch.stack(r["v_traj"]).numpy()
    mol = lambda e: np.array([e[off[b]:off[b + 1]].max() for b in range(B)])  # noqa: E731
    out = {"what": "CPU oracle (float32) free-running over the reference's chain", "fixture": "chain_b256_s1000_hash.npz", "B": B, "steps": S, "n_atoms": n,
           "threads": a.threads, "seconds": round(time.time() - t0, 1), "per_snapshot": [], "diverged": []}
    n_snap = (S - 1) // every + 1
    mol_err = np.zeros((n_snap, B))
    for j in range(n_snap):
        e = mol(np.abs(pos_traj[j * every].astype(np.float64) - c["pos_traj_sub"][j]).max(-1))
        mol_err[j] = e
        out["per_snapshot"].append({"step": j * every, "max": float(e.max()), "median": float(np.median(e)), "n_over_1e-4": int((e > 1e-4).sum()),
                                    "atom_type_mismatches": int((v_traj[j * every] != c["v_traj_sub"][j]).sum())})
    if S == int(c["S"]):
        e = mol(np.abs(r["pos"].numpy().astype(np.float64) - c["pos"]).max(-1))
        out["end"] = {"step": S - 1, "max": float(e.max()), "median": float(np.median(e)), "n_over_1e-4": int((e > 1e-4).sum()),
                      "atom_type_mismatches": int((r["v"].numpy() != c["v"]).sum())}
        mol_err = np.concatenate([mol_err, e[None]], 0)
    for b in np.where((mol_err > 1e-4).any(0))[0]:
        first = int(np.where(mol_err[:, b] > 1e-4)[0][0])
        lo, hi = max(0, (first - 1) * every), min(S, first * every)
        states = [bb["init_pos"][off[b]:off[b + 1]]] if lo == 0 else []
        states += [pos_traj[s, off[b]:off[b + 1]] for s in range(max(lo - 1, 0), hi)]
        out["diverged"].append({"mol": int(b), "atoms": int(bb["counts"][b]), "first_snapshot_over_1e-4": min(first * every, S - 1),
                                "err_there": float(mol_err[first, b]), "min_knn_margin_rel_in_window": float(min(knn_margin_rel(x, 8) for x in states)),
                                "end_err": float(mol_err[-1, b])})
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in out if k not in ("per_snapshot",)}, indent=1))
    print(json.dumps(out["per_snapshot"]))


if __name__ == "__main__":
    main()
