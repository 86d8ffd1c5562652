#!/usr/bin/env python3
"""Developer tool (GPU box): how an idle gap in front of a short chain changes its time (the device's clocks fall within a
millisecond of idling and need ~20 ms of load to return).  200 steps of load, a host sleep of X ms, then 20 timed steps."""
import os, sys, time
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd import ScorePosNet3D, synth
from shapemol_amd.runtime import ChainRunner
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
m = ScorePosNet3D(cfg, 15); m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()}); m = m.to("cuda:0")
bb = synth.synthetic_batch(256, seed=2021)
r = ChainRunner(m, len(bb["batch"]), 256, 200, keep_traj=True)
r.load_batch(bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"])
r.run(200); r.synchronize()
for gap_ms in (0.0, 0.2, 0.5, 1.0, 2.0, 5.0, 20.0, 100.0):
    ts = []
    for rep in range(5):
        r.run(200); r.synchronize()
        t_end = time.perf_counter() + gap_ms * 1e-3
        while time.perf_counter() < t_end:
            pass
        t0 = time.perf_counter()
        r.run(20); r.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"idle gap {gap_ms:6.1f} ms -> 20 steps in {np.median(ts):.3f} ms ({np.median(ts) / 20:.4f} ms/step; min {min(ts):.3f}, max {max(ts):.3f})", flush=True)
