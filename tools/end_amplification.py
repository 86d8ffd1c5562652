#!/usr/bin/env python3
"""Why the free-running chain's error grows ~8x in its last 50 steps ("What's weak").

Hypothesis: nothing amplifies in the kernels; the bulk of the molecules inherits, through the train-mode batch-norm
(its statistics are sums over ALL atoms of the batch, models/shape_vn_layers.py:50-61), a share of the displacement of the
few molecules that left the reference's trajectory after a kNN flip -- and in the last steps the posterior mean hands the
network's x0 estimate through with weight c0 -> 1 (molopt_score_model.py:400-404), so that share becomes visible.  Then the
bulk error at the end is proportional to the LARGEST displacement in the batch, whatever the backend.

Test: the last window of the reference's B = 256 x 1000 chain (from its state after step 950, fixture
chain_b256_s1000_hash.npz, 49 steps, same noise), run
  (a) from the reference's state as it is:            error of every molecule vs the reference's end state,
  (b) with ONE molecule displaced rigidly by d:       error of the OTHER 255 molecules, for d = 1e-3, 1e-2, 1e-1 Angstrom.
Reports per run the median / max error of the untouched molecules every 10 steps (fixture chain_b256_s1000_tail_hash.npz)
and at the end.

    python tools/end_amplification.py --backend oracle [--threads 6]     (CPU, ~4 min per run)
    python tools/end_amplification.py --backend hip [--opt edge_bf16=1 --opt node_f16=0]     (GPU box)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import T, golden, hash_noise, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["oracle", "hip"], required=True)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--mol", type=int, default=94, help="the molecule to displace (94: the first one to flip in the free runs)")
    ap.add_argument("--disp", type=float, nargs="*", default=[0.0, 1e-3, 1e-2, 1e-1])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    c, ct = golden("chain_b256_s1000_hash.npz"), golden("chain_b256_s1000_tail_hash.npz")
    B, S, seed, every = int(c["B"]), int(c["S"]), int(c["seed"]), int(c["every"])
    bb = synth.synthetic_batch(B, seed=seed, max_atoms=38)
    n = len(bb["batch"])
    off = np.concatenate([[0], np.cumsum(bb["counts"])])
    j = len(c["pos_traj_sub"]) - 1
    s0 = j * every + 1                     # first reverse step of the window (the snapshot is the state after step j * every)
    ns = S - s0
    eps = np.stack([synth.step_noise(n, 15, s, seed=seed)[0] for s in range(s0, S)])
    u = np.stack([synth.step_noise(n, 15, s, seed=seed)[1] for s in range(s0, S)])
    tail_first, tail_every = int(ct["first_step"]), int(ct["every"])
    others = np.ones(B, bool); others[a.mol] = False
    if a.backend == "oracle":
        from util import oracle_model
        from oracle import shapemol_oracle as O
        torch.set_num_threads(a.threads)
        sd, dm, _, _ = oracle_model()

        def run(pos0):
            r = O.sample_chain(sd, dm, T(pos0), T(c["v_traj_sub"][j].astype(np.int64)), T(bb["batch"]), T(bb["shape"]), ns,
                               lambda s: (eps[s], u[s]), keep_traj=True, first_step=s0)
            return torch.stack(r["pos_traj"]).numpy()
    else:
        from util import hip_model
        m = hip_model()
        for kv in a.opt:
            k, v = kv.split("=")
            m.set_option(k, int(v))

        def run(pos0):
            r = m.sample_diffusion(T(pos0, "cuda:0"), T(c["v_traj_sub"][j].astype(np.int64), "cuda:0"), T(bb["batch"], "cuda:0"),
                                   T(bb["shape"], "cuda:0").view(B, -1), num_steps=ns, center_pos_mode="none",
                                   noise=(T(eps, "cuda:0"), T(u, "cuda:0")), first_step=s0)
            return torch.stack(r["pos_traj"]).numpy()
    mol = lambda e: np.array([e[off[b]:off[b + 1]].max() for b in range(B)])  # noqa: E731
    res = {"backend": a.backend + ("" if not a.opt else " " + " ".join(a.opt)), "window": [s0, S - 1], "displaced_mol": a.mol, "runs": []}
    for d in a.disp:
        pos0 = c["pos_traj_sub"][j].copy()
        pos0[off[a.mol]:off[a.mol + 1]] += np.float32(d) * np.array([0.6, -0.64, 0.48], np.float32)      # rigid shift of length d
        traj = run(pos0)
        rec = {"displacement": d, "steps": []}
        for k in range(len(ct["pos_traj_tail"])):
            st = tail_first + k * tail_every
            e = mol(np.abs(traj[st - s0].astype(np.float64) - ct["pos_traj_tail"][k]).max(-1))
            rec["steps"].append({"step": st, "median_others": float(np.median(e[others])), "max_others": float(e[others].max()), "displaced": float(e[a.mol])})
        e = mol(np.abs(traj[-1].astype(np.float64) - c["pos"]).max(-1))
        rec["end"] = {"step": S - 1, "median_others": float(np.median(e[others])), "max_others": float(e[others].max()),
                      "n_others_over_1e-4": int((e[others] > 1e-4).sum()), "displaced": float(e[a.mol])}
        res["runs"].append(rec)
        print(json.dumps(rec), flush=True)
    out = a.out or os.path.join(ROOT, "gpurun_out" if a.backend == "hip" else os.path.join("profiles", "r03"),
                                f"end_amplification_{a.backend}{'_' + '_'.join(a.opt).replace('=', '') if a.opt else ''}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
