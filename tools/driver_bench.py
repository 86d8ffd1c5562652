#!/usr/bin/env python3
"""Developer tool: end-to-end time of the sampling driver (shapemol_amd.sampling.sample_diffusion_ligand) for one
shape condition: chain + trajectory D2H + per-molecule unbatching, against the chain alone.
    python tools/driver_bench.py [--samples 256] [--batch 256] [--steps 1000]"""
import argparse, os, sys, time
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd import ScorePosNet3D, synth
from shapemol_amd.synth import moses_atom_prior
from shapemol_amd.sampling import sample_diffusion_ligand

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=256); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=1000)
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
m = ScorePosNet3D(cfg, 15)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()})
m = m.to("cuda:0")
nums, p = moses_atom_prior()
rs = np.random.RandomState(0)
shape_emb = synth.synthetic_batch(1, seed=5)["shape"][0]
fn = lambda n: rs.choice(nums, size=n, p=p).tolist()  # noqa: E731
sample_diffusion_ligand(m, shape_emb, min(a.samples, a.batch), a.batch, num_steps=20, sample_num_atoms="size", sample_func=fn, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = sample_diffusion_ligand(m, shape_emb, a.samples, a.batch, num_steps=a.steps, sample_num_atoms="size", sample_func=fn, seed=2)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"driver end to end: {a.samples} molecules, {a.steps} steps, batch {a.batch}: {dt:.3f} s -> "
      f"{a.samples / dt * (1000 / a.steps):.1f} molecules/s (1000-step equivalent); per-batch times {['%.3f' % t for t in out[6]]}")
