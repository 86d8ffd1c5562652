#!/usr/bin/env python3
"""Developer tool: where the host time of a training step goes (cProfile over 5 steps; the device runs behind)."""
import cProfile, os, pstats, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd import ScorePosNet3D, synth
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
m = ScorePosNet3D(cfg, 15)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()})
m = m.to("cuda:0").train()
B = 256
bb = synth.synthetic_batch(B, seed=2021)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
args = (T(bb["init_pos"] * 1.5), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]).view(B, -1))
t = torch.randint(0, 1000, (B,), device="cuda:0")


def step():
    m.zero_grad(set_to_none=True)
    m.get_diffusion_loss(*args, time_step=t, eval_mode=False)["loss"].backward()


step(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
