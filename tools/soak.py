#!/usr/bin/env python3
"""Stability check on the GPU box: the same 1000-step chain (same seed, device Philox noise) repeated at several batch sizes
must give bit-identical results every time, and the library's status flags must stay clear.
    python tools/soak.py [--reps 6]"""
import argparse, hashlib, os, sys, time
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd import ScorePosNet3D, synth
from shapemol_amd.runtime import ChainRunner
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=6); a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
m = ScorePosNet3D(cfg, 15)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()})
m = m.to("cuda:0")
ok = True
for B in (256, 64, 1024, 256):
    bb = synth.synthetic_batch(B, seed=2021)
    r = ChainRunner(m, len(bb["batch"]), B, 1000, keep_traj=False)
    r.load_batch(bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"])
    digests = []
    t0 = time.time()
    for _ in range(a.reps):
        r.run(1000, seed=77)
        r.synchronize()                      # raises on any status flag
        h = hashlib.sha256(r.out_pos.cpu().numpy().tobytes() + r.out_v.cpu().numpy().tobytes()).hexdigest()[:16]
        digests.append(h)
    same = len(set(digests)) == 1
    ok &= same
    print(f"B={B:5d} atoms={len(bb['batch']):6d}: {a.reps} x 1000 steps in {time.time() - t0:.1f} s, digest {digests[0]} {'identical every time' if same else 'DIFFERENT: ' + str(digests)}", flush=True)
    r.close()
    del r
print("soak OK" if ok else "soak FAILED")
sys.exit(0 if ok else 1)
