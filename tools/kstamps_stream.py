#!/usr/bin/env python3
"""Developer tool: in-kernel timeline of one launch of the streaming edge kernels (sm_edge_stream.h), per role.  Needs a --stamps
build:  SHAPEMOL_LIB=stamps python tools/kstamps_stream.py --sel 1      # 1 edge_x2h, 2 edge_h2x (layer 0)
Stamps (100 MHz counter): 0 start, 1 prologue barrier passed, 2 first round produced (barrier), 3 / 4 / 5 this wave's work of rounds
0 / 1 / 2 done (before the round's barrier), 6 last round's barrier passed."""
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
ap.add_argument("--chain", type=int, default=24)
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
m = ScorePosNet3D(cfg, 15)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()})
m = m.to("cuda:0")
m.set_option("edge_bf16", 2)
m.set_option("node_f16", 0)
bb = synth.synthetic_batch(a.batch, seed=2021)
r = ChainRunner(m, len(bb["batch"]), a.batch, a.chain, keep_traj=False)
r.load_batch(bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"])
r.run(a.chain); r.synchronize()
m.set_option("kstamp_sel", a.sel)
r.run(a.chain); r.synchronize()          # the stamps of the LAST step's launch survive
st = m.debug_read("kstamps", (4096 * 16, 8), np.uint64).astype(np.int64)
NW = 12
rows = np.nonzero(st[:, 0] > 0)[0]
st, wave = st[rows], rows % NW
t0 = st[:, 0].min()
print(f"waves stamped: {len(st)} ({len(st) // NW} workgroups); times in us relative to the earliest start")
for role, sel in (("consumers", wave < 8), ("key producers", (wave >= 8) & ((wave - 8) % 2 == 0)), ("value producers", (wave >= 8) & ((wave - 8) % 2 == 1))):
    s_ = st[sel]
    line = []
    for k in range(8):
        col = s_[:, k]; col = col[col > 0]
        line.append("   -  " if len(col) == 0 else f"{np.median((col - t0) / 100.0):6.2f}")
    print(f"  {role:16s} median stamp 0..7: " + " ".join(line))
    line = []
    for k in range(8):
        col = s_[:, k]; col = col[col > 0]
        line.append("   -  " if len(col) == 0 else f"{((col - t0) / 100.0).max():6.2f}")
    print(f"  {'':16s}    max stamp 0..7: " + " ".join(line))
print(f"  launch: {(st.max() - t0) / 100.0:.2f} us from first start to last stamp")
