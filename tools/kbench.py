#!/usr/bin/env python3
"""Developer tool: per-kernel-class launch times (HIP events, eager) + graph-replay step time.
    python tools/kbench.py [--batch 256] [--steps 20] [--opt edge_waves=8] ..."""
import argparse, os, sys, time
if '--stamps-build' in sys.argv:
    os.environ['SHAPEMOL_STAMPS'] = '1'; sys.argv.remove('--stamps-build')
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd import ScorePosNet3D, synth
from shapemol_amd.runtime import ChainRunner

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--graph-steps", type=int, default=200)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--k", type=int, default=8)
ap.add_argument("--atoms", type=str, default="")
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
cfg["knn"] = a.k
m = ScorePosNet3D(cfg, 15)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()})
m = m.to("cuda:0")
rng = tuple(int(x) for x in a.atoms.split(",")) if a.atoms else None
bb = synth.synthetic_batch(a.batch, seed=2021, atoms_range=rng)
n = len(bb["batch"])
r = ChainRunner(m, n, a.batch, max(a.steps, a.graph_steps, 20), keep_traj=True)
r.load_batch(bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"])
for o in a.opt:
    k, v = o.split("=")
    m.set_option(k, int(v))
r.run(5); r.synchronize()
prof = r.profile(a.steps); r.synchronize()
tot = 0.0
for k, (ms, cnt) in prof.items():
    tot += ms
    print(f"{k:12s} {ms / a.steps * 1e3:9.1f} us/step  {cnt // a.steps:3d} launches/step  {ms / cnt * 1e3:8.2f} us/launch")
print(f"sum(events) {tot / a.steps * 1e3:.1f} us/step   atoms={n}")
r.run(20); r.synchronize()
t0 = time.perf_counter(); r.run(a.graph_steps); r.synchronize(); dt = time.perf_counter() - t0
import numpy as np
m.set_option("stamps", 1)
r.run(a.graph_steps); r.synchronize()
st = m.debug_read("stamps", (1024, 2), np.uint64)[:a.graph_steps].astype(np.int64)
d = np.diff(st, axis=0)
print(f"shader clock during chain: median {np.median(d[:, 0] / d[:, 1]) * 100:.0f} MHz  (step {np.median(d[:, 1]) / 100:.1f} us by realtime counter)")
m.set_option("stamps", 0)
print(f"graph replay: {dt / a.graph_steps * 1e3:.4f} ms/step -> {a.batch / (1000 * dt / a.graph_steps):.1f} molecules/s")
