#!/usr/bin/env python3
"""Measurement of the widened rows (SURVEY.md section 8 f2, f3 and the validation half of f4) on the GPU box, with their CPU counterparts timed beside:
the frozen shape encoder (shapes/s, against the torch-CPU oracle) and the point-cloud guidance (us per guided step against
the reference's own algorithm: numpy + sklearn KD-tree on the host plus the D2H / H2D copies it needs).
    python tools/aux_bench.py > profiles/r02_final/aux_bench.json"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shapemol_amd  # noqa: E402
from shapemol_amd import synth  # noqa: E402
from util import hip_model  # noqa: E402

DEV = "cuda:0"
out = {}

# ---- shape encoder ------------------------------------------------------------------------------------------------
from oracle import shape_encoder_oracle as SE  # noqa: E402
enc = shapemol_amd.VN_DGCNN_Encoder(128, 32, 4, 20)
sd = synth.shape_encoder_state_dict(128, 32, 4, 17)
enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
enc = enc.to(DEV)
for B in (4, 32):
    pts = (synth.hash_normal((B, 512, 3), 401, 9) * np.array([1.5, 1.0, 0.6], np.float32)).astype(np.float32)
    x = torch.from_numpy(pts).to(DEV)
    enc(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        enc(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    rec = {"shapes": B, "points": 512, "ms": round(dt * 1e3, 3), "shapes_per_s": round(B / dt, 1)}
    if B == 4:
        torch.set_num_threads(16)
        sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
        SE.encode(sdt, torch.from_numpy(pts))
        t0 = time.perf_counter()
        for _ in range(3):
            SE.encode(sdt, torch.from_numpy(pts))
        dc = (time.perf_counter() - t0) / 3
        rec["cpu_oracle_ms"] = round(dc * 1e3, 1)
        rec["cpu_threads"] = torch.get_num_threads()
        # executed FLOPs: Gram tiles 4 x N^2 x 384 x 2 + point products 4 x N x 3 x 512 x 128 x 2 per shape (+ the edge stage, VALU)
        rec["gflop_per_shape"] = round((4 * 512 * 512 * 384 * 2 + 4 * 512 * 3 * 512 * 128 * 2) / 1e9, 3)
    out[f"shape_encoder_b{B}"] = rec

# ---- point-cloud guidance -----------------------------------------------------------------------------------------------
m = hip_model()
cloud = (synth.hash_normal((512, 3), 301, 5) * 1.2).astype(np.float64)
bb = synth.synthetic_batch(256, seed=2021)
n = len(bb["batch"])
pred = (synth.hash_normal((n, 3), 302, 5) * 1.6).astype(np.float32)
dev_pred = torch.from_numpy(pred).to(DEV)
m.pointcloud_shape_guidance((cloud, None, 0.2), dev_pred.clone(), seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    m.pointcloud_shape_guidance((cloud, None, 0.2), dev_pred.clone(), seed=1)
torch.cuda.synchronize()
t_call = (time.perf_counter() - t0) / 20          # includes the per-call cloud upload of the standalone entry point
# the reference's host algorithm on the same input (oracle restatement of it with a real KD-tree), plus the copies it needs
from sklearn.neighbors import KDTree  # noqa: E402
tree = KDTree(cloud)
def host_guidance(p_dev):
    p = np.array(p_dev.cpu())
    d, idx = tree.query(p, k=3)
    far = np.where(d.mean(1) > 0.2)[0]
    pts, pidx = p[far].astype(np.float64), idx[far]
    for _ in range(5):
        if len(far) == 0:
            break
        pts = pts - (np.random.random(len(far)) * 0.6 + 0.2)[:, None] * (pts - cloud[pidx].mean(1))
        d, idx = tree.query(pts, k=3)
        inside = d.mean(1) < 0.2
        p[far[inside]] = pts[inside]
        far, pts, pidx = far[~inside], pts[~inside], idx[~inside]
    p[far] = pts
    return torch.from_numpy(p).to(p_dev.device)
host_guidance(dev_pred)
t0 = time.perf_counter()
for _ in range(20):
    host_guidance(dev_pred)
torch.cuda.synchronize()
t_host = (time.perf_counter() - t0) / 20
# in-chain cost: guided vs unguided 200-step chains (all steps guided: grad_step below every t)
args = (torch.from_numpy(bb["init_pos"]).to(DEV), torch.from_numpy(bb["init_v"]).to(DEV), torch.from_numpy(bb["batch"]).to(DEV),
        torch.from_numpy(bb["shape"]).to(DEV).view(256, -1))
def chain(**kw):
    m.sample_diffusion(*args, num_steps=200, center_pos_mode="none", seed=3, return_traj=False, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.sample_diffusion(*args, num_steps=200, center_pos_mode="none", seed=3, return_traj=False, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 200
import contextlib, io  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    plain = chain()
    guided = chain(use_pointcloud_data=(cloud, None, 0.2), grad_step=0)
out["pointcloud_guidance"] = {"atoms": n, "cloud_points": 512, "standalone_call_us": round(t_call * 1e6, 1),
                              "host_reference_algorithm_us": round(t_host * 1e6, 1),
                              "chain_us_per_step_unguided": round(plain * 1e6, 1), "chain_us_per_step_guided": round(guided * 1e6, 1),
                              "in_chain_cost_us_per_guided_step": round((guided - plain) * 1e6, 1)}
# ---- validation loss (f4, the half validate() needs): ten time steps per batch as scripts/train_diffusion.py:178 ----------
from oracle import shapemol_oracle as O  # noqa: E402
from util import model_cfg  # noqa: E402
cfg = model_cfg()
sdn = synth.synthetic_state_dict(cfg, seed=7)
sdn.update(synth.running_stats(m.dims.L, m.dims.heads, 23))
mv = shapemol_amd.ScorePosNet3D(cfg, 15)
mv.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
mv = mv.to(DEV).eval()
pos0 = torch.from_numpy((synth.hash_normal((n, 3), 501, 41) * 1.5).astype(np.float32)).to(DEV)
ts = np.linspace(0, 999, 10).astype(int)
def validate_batch():
    tot = 0.0
    with torch.no_grad():
        for t in ts:
            r = mv.get_diffusion_loss(pos0, args[1], args[2], args[3], time_step=torch.full((256,), int(t), dtype=torch.long, device=DEV), eval_mode=True)
            tot += float(r["loss"])
    return tot
validate_batch()
t0 = time.perf_counter()
validate_batch()
torch.cuda.synchronize()
t_dev = time.perf_counter() - t0
sd_o, dm_o = O.state_dict_from_numpy(sdn), O.Dims(cfg)
noise = synth.hash_normal((n, 3), 502, 41); uu = synth.hash_uniform((n, 15), 503, 41)
torch.set_num_threads(16)
t0 = time.perf_counter()
O.diffusion_loss(sd_o, dm_o, pos0.cpu(), args[1].cpu(), args[2].cpu(), torch.from_numpy(bb["shape"]), torch.full((256,), 500, dtype=torch.long),
                 torch.from_numpy(noise), torch.from_numpy(uu), bn_eval=True)
t_cpu = time.perf_counter() - t0
out["validation_loss"] = {"molecules": 256, "atoms": n, "time_steps_per_batch": 10, "device_ms_per_batch": round(t_dev * 1e3, 2),
                          "device_ms_per_evaluation": round(t_dev * 1e2, 3), "cpu_oracle_ms_per_evaluation": round(t_cpu * 1e3, 1),
                          "cpu_threads": torch.get_num_threads(),
                          "note": "get_diffusion_loss(eval_mode=True) as validate() calls it, float(loss) read back after every evaluation as the script does"}
print(json.dumps(out, indent=1))
