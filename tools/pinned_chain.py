#!/usr/bin/env python3
"""Developer tool: the free-running B = 256 (or 1024) x 1000 chain with the reference's fragile kNN choices pinned, error per snapshot.
    python tools/pinned_chain.py [--case b256|b1024] [--mode exact|f16x2] [--thr 5e-4] [--pins tests/golden/chain_b256_s1000_pins.npz]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import T, golden, hash_noise, hip_model, synth  # noqa: E402

DEV = "cuda:0"
ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="exact")
ap.add_argument("--thr", type=float, default=1.0)
ap.add_argument("--pins", default="")
ap.add_argument("--case", default="b256", choices=["b256", "b1024"])
a = ap.parse_args()
c = golden(f"chain_{a.case}_s1000_hash.npz")
if a.case == "b256":
    ct = golden("chain_b256_s1000_tail_hash.npz")
    tail = (int(ct["first_step"]), int(ct["every"]), ct["pos_traj_tail"])
else:
    tail = (int(c["tail_first"]), int(c["tail_every"]), c["pos_traj_tail"])
pins = np.load(a.pins or os.path.join(ROOT, f"tests/golden/chain_{a.case}_s1000_pins.npz"))
m = hip_model()
if a.mode == "exact":
    m.set_option("edge_bf16", 2); m.set_option("node_f16", 0)
else:
    m.set_option("node_f16", 1); m.set_option("edge_bf16", 3)
B, S, seed, every = int(c["B"]), int(c["S"]), int(c["seed"]), int(c["every"])
bb = synth.synthetic_batch(B, seed=seed, max_atoms=38)
eps, u = hash_noise(len(bb["batch"]), S, seed)
k = pins["margin"] < a.thr
if k.any():
    m.set_knn_pins(pins["step"][k], pins["atom"][k], pins["nbr"][k])
print(f"mode {a.mode}, pins {int(k.sum())} (margin < {a.thr:g})", flush=True)
r = m.sample_diffusion(T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV).view(B, -1), num_steps=S,
                       center_pos_mode="none", noise=(T(eps, DEV), T(u, DEV)), use_graph=True)
pos = torch.stack(r["pos_traj"]).numpy()
vt = torch.stack(r["v_traj"]).numpy()
off = np.concatenate([[0], np.cumsum(bb["counts"])])


def mol_err(p, q):
    e = np.abs(p.astype(np.float64) - q).max(-1)
    return np.array([e[off[b]:off[b + 1]].max() for b in range(B)])


for i, st in enumerate(range(0, S, every)):
    me = mol_err(pos[st], c["pos_traj_sub"][i])
    print(f"  step {st:4d}: max {me.max():.2e} median {np.median(me):.2e} mols>1e-4 {int((me > 1e-4).sum())} v-mismatch {int((vt[st] != c['v_traj_sub'][i]).sum())}")
f0, ev = tail[0], tail[1]
for i in range(len(tail[2])):
    st = f0 + i * ev
    me = mol_err(pos[st], tail[2][i])
    print(f"  step {st:4d}: max {me.max():.2e} median {np.median(me):.2e} mols>1e-4 {int((me > 1e-4).sum())} worst mol {int(me.argmax())}")
me = mol_err(r["pos"].cpu().numpy(), c["pos"])
print(f"  end      : max {me.max():.2e} median {np.median(me):.2e} mols>1e-4 {int((me > 1e-4).sum())} worst mols {np.argsort(-me)[:8].tolist()} {np.sort(me)[::-1][:8]}")
