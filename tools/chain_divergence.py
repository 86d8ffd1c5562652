#!/usr/bin/env python3
"""Where and why a long chain leaves the reference's trajectory (GPU box; diagnostic for the full-length gate).

Runs the B=256 x 1000 chain of tests/golden/chain_b256_s1000_hash.npz, reports per molecule the first snapshot at
which its coordinates differ from the reference's by more than 1e-4, and, for those molecules, the smallest
gap between the k-th and (k+1)-th neighbour distance seen in the steps before (float64, from the GPU trajectory):
kNN is discontinuous, so a neighbour flip caused by 1e-7-level rounding differences is the expected mechanism."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import T, golden, hash_noise, hip_model, synth  # noqa: E402

DEV = "cuda:0"


def knn_margin(x, k=8):  # same as tests/tools_knn.py
    """min over atoms of (d2_{k+1} - d2_k) / d2_k for one molecule (float64); inf if the molecule has <= k+1 atoms."""
    n = len(x)
    if n <= k + 1:
        return np.inf
    d = ((x[:, None, :].astype(np.float64) - x[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    s = np.sort(d, 1)
    return float(((s[:, k] - s[:, k - 1]) / s[:, k - 1]).min())


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "chain_b256_s1000_hash.npz"
    c = golden(name)
    B, S, seed, every = int(c["B"]), int(c["S"]), int(c["seed"]), int(c["every"])
    bb = synth.synthetic_batch(B, seed=seed, max_atoms=38)
    n = len(bb["batch"])
    eps, u = hash_noise(n, S, seed)
    m = hip_model()
    r = m.sample_diffusion(T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV).view(B, -1),
                           num_steps=S, center_pos_mode="none", noise=(T(eps, DEV), T(u, DEV)))
    pos_traj = torch.stack(r["pos_traj"]).numpy()
    off = np.concatenate([[0], np.cumsum(bb["counts"])])
    snaps = pos_traj[::every]
    err = np.abs(snaps.astype(np.float64) - c["pos_traj_sub"]).max(-1)          # (n_snap, N)
    mol_err = np.stack([err[:, off[b]:off[b + 1]].max(1) for b in range(B)], 1)   # (n_snap, B)
    end_err = np.abs(r["pos"].cpu().numpy().astype(np.float64) - c["pos"]).max(-1)
    mol_end = np.array([end_err[off[b]:off[b + 1]].max() for b in range(B)])
    out = {"fixture": name, "n_mols": B, "n_atoms": n, "per_snapshot": [], "diverged": []}
    for s in range(len(snaps)):
        e = mol_err[s]
        out["per_snapshot"].append({"step": s * every, "max": float(e.max()), "median": float(np.median(e)),
                                    "n_over_1e-4": int((e > 1e-4).sum()), "n_over_1e-5": int((e > 1e-5).sum())})
    out["end"] = {"max": float(mol_end.max()), "median": float(np.median(mol_end)), "n_over_1e-4": int((mol_end > 1e-4).sum()),
                  "max_of_the_rest": float(mol_end[mol_end <= 1e-4].max())}
    bad = np.where((mol_err > 1e-4).any(0) | (mol_end > 1e-4))[0]
    for b in bad:
        over = np.where(mol_err[:, b] > 1e-4)[0]
        first = int(over[0]) if len(over) else len(snaps)
        lo, hi = max(0, (first - 1) * every), min(S, first * every)
        margins = [knn_margin(pos_traj[s, off[b]:off[b + 1]]) for s in range(lo, hi)]
        # the state that enters step s is pos_traj[s - 1] (init_pos for s = 0)
        out["diverged"].append({"mol": int(b), "atoms": int(bb["counts"][b]), "first_snapshot_over_1e-4": first * every,
                                "min_knn_margin_rel_in_window": float(min(margins)) if margins else None,
                                "end_err": float(mol_end[b])})
    allm = [min(knn_margin(pos_traj[s, off[b]:off[b + 1]]) for s in range(0, S, 7)) for b in range(0, B, 8)]
    out["typical_min_margin_sampled"] = float(np.median(allm))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "chain_divergence.json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("end", "diverged", "typical_min_margin_sampled")}, indent=1))
    print(json.dumps(out["per_snapshot"]))


if __name__ == "__main__":
    main()
