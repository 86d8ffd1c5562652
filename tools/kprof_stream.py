#!/usr/bin/env python3
"""Developer tool: where the waves of one streaming edge launch (sm_edge_stream.h) spend their time, per role.  Needs the attribution
build (build.sh --variant prof -DSM_STREAM_PROF=1):  SHAPEMOL_LIB=prof python tools/kprof_stream.py --sel 1   # 1 edge_x2h, 2 edge_h2x
Every wave accumulates (100 MHz counter) its prologue, its first unit / weight load, the work of its rounds, the waits for its
gathers (producers) and the waits at the round barriers, and writes the sums once at its end -- the timed launch is within a few per
cent of the plain one (the number printed last against rocprofv3's)."""
import argparse
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd import ScorePosNet3D, synth  # noqa: E402
from shapemol_amd.runtime import ChainRunner  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sel", type=int, default=1)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--chain", type=int, default=200)
ap.add_argument("--knn", type=int, default=0)
ap.add_argument("--atoms", default="")
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
if a.knn:
    cfg["knn"] = a.knn
m = ScorePosNet3D(cfg, 15)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()})
m = m.to("cuda:0")
kw = {}
if a.atoms:
    lo, hi = (int(v) for v in a.atoms.split(","))
    kw = dict(min_atoms=lo, max_atoms=hi)
bb = synth.synthetic_batch(a.batch, seed=2021, **kw)
r = ChainRunner(m, len(bb["batch"]), a.batch, a.chain, keep_traj=False)
r.load_batch(bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"])
r.run(a.chain); r.synchronize()
m.set_option("kstamp_sel", a.sel)
r.run(a.chain); r.synchronize()          # the sums of the LAST step's launch survive
st = m.debug_read("kstamps", (4096 * 16, 8), np.uint64).astype(np.int64)
NW = 12
rows = np.nonzero(st[:, 0] > 0)[0]
st, wave = st[rows], rows % NW
t0 = st[:, 0].min()
print(f"waves: {len(st)} ({len(st) // NW} workgroups, {int(np.median(st[:, 7]))} rounds); microseconds, median over the role's waves [max]")
print(f"  {'role':16s} {'start':>7s} {'prologue':>9s} {'1st unit':>9s} {'rounds: work':>13s} {'gather wait':>12s} {'barrier wait':>13s} {'total':>7s}")
for role, sel in (("consumers", wave < 8), ("key producers", (wave >= 8) & ((wave - 8) % 2 == 0)), ("value producers", (wave >= 8) & ((wave - 8) % 2 == 1))):
    s_ = st[sel]
    med = lambda c: np.median(s_[:, c]) / 100.0
    mx = lambda c: s_[:, c].max() / 100.0
    print(f"  {role:16s} {np.median(s_[:, 0] - t0) / 100.0:7.2f} {med(1):9.2f} {med(2):9.2f} {med(3):13.2f} {med(4):12.2f} {med(5):13.2f} {med(6):7.2f} [{mx(6):.2f}]")
print(f"  launch: {((st[:, 0] + st[:, 6]).max() - t0) / 100.0:.2f} us from the first wave's start to the last wave's last barrier")
