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
ap.add_argument("--samples", type=int, default=1024); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--pipeline", type=int, nargs="*", default=[1, 2], help="batches in flight (1 = sequential driver)")
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
m = ScorePosNet3D(cfg, 15)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()})
m = m.to("cuda:0")
nums, p = moses_atom_prior()
rs = np.random.RandomState(0)
shape_emb = synth.synthetic_batch(1, seed=5)["shape"][0]
fn = lambda n: rs.choice(nums, size=n, p=p).tolist()  # noqa: E731
import json
res = {"samples": a.samples, "batch": a.batch, "steps": a.steps, "runs": []}
for depth in a.pipeline:
    sample_diffusion_ligand(m, shape_emb, min(a.samples, 2 * a.batch), a.batch, num_steps=20, sample_num_atoms="size", sample_func=fn, seed=1, pipeline=depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sample_diffusion_ligand(m, shape_emb, a.samples, a.batch, num_steps=a.steps, sample_num_atoms="size", sample_func=fn, seed=2, pipeline=depth)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = sum(x.nbytes for k in (2, 3, 4, 5, 7, 8) for x in out[k])
    rate = a.samples / dt * (1000 / a.steps)
    res["runs"].append({"pipeline": depth, "seconds": round(dt, 3), "molecules_per_s_1000step_equiv": round(rate, 1), "trajectory_bytes_delivered": nbytes,
                        "per_batch_s": [round(t, 3) for t in out[6]]})
    print(f"driver end to end, pipeline={depth}: {a.samples} molecules, {a.steps} steps, batch {a.batch}: {dt:.3f} s -> {rate:.1f} molecules/s "
          f"(1000-step equivalent), {nbytes / 1e9:.2f} GB of trajectories delivered; per-batch times {['%.3f' % t for t in out[6]]}", flush=True)
    del out
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "driver_bench.json"), "w"), indent=1)
