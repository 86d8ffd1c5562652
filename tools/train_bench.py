#!/usr/bin/env python3
"""Developer tool: time of one training step's forward + backward on the device (get_diffusion_loss with autograd enabled,
shapemol_amd/training.py: MLP blocks in HIP, glue in torch device ops) against the validation-mode evaluation (HIP sampling
kernels, no gradients) on the same batch.
    python tools/train_bench.py [--batch 256] [--reps 5]"""
import argparse, json, os, sys, time
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd import ScorePosNet3D, synth

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
m = ScorePosNet3D(cfg, 15)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()})
m = m.to("cuda:0").train()
bb = synth.synthetic_batch(a.batch, seed=2021)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
args = (T(bb["init_pos"] * 1.5), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]).view(a.batch, -1))
t = torch.randint(0, 1000, (a.batch,), device="cuda:0")


def step():
    m.zero_grad(set_to_none=True)
    r = m.get_diffusion_loss(*args, time_step=t, eval_mode=False)
    r["loss"].backward()
    return float(r["loss"])


step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    loss = step()
torch.cuda.synchronize()
dt_train = (time.perf_counter() - t0) / a.reps
with torch.no_grad():
    m.get_diffusion_loss(*args, time_step=t, eval_mode=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps * 4):
        m.get_diffusion_loss(*args, time_step=t, eval_mode=True)
    torch.cuda.synchronize()
    dt_val = (time.perf_counter() - t0) / (a.reps * 4)
print(json.dumps({"batch": a.batch, "atoms": len(bb["batch"]), "train_step_fwd_bwd_ms": round(dt_train * 1e3, 2), "validation_eval_ms": round(dt_val * 1e3, 3),
                  "loss": loss, "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
