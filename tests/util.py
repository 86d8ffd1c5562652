"""Shared helpers of the test-suite (inputs, oracle handles, the HIP model)."""
import json
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
TRAIN_YML = os.path.join(ROOT, "config", "training",
                         "dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")

from shapemol_amd import synth  # noqa: E402
from oracle import shapemol_oracle as O  # noqa: E402


def model_cfg(**overrides):
    cfg = yaml.safe_load(open(TRAIN_YML))["model"]
    cfg.update(overrides)
    return cfg


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


_cache = {}


def oracle_model(seed=7, **overrides):
    key = ("o", seed, json.dumps(overrides, sort_keys=True))
    if key not in _cache:
        cfg = model_cfg(**overrides)
        sdn = synth.synthetic_state_dict(cfg, seed=seed)
        _cache[key] = (O.state_dict_from_numpy(sdn), O.Dims(cfg), cfg, sdn)
    return _cache[key]


def hip_model(seed=7, **overrides):
    """ScorePosNet3D on cuda:0 with the synthetic weights of `seed`."""
    key = ("h", seed, json.dumps(overrides, sort_keys=True))
    if key not in _cache:
        import shapemol_amd
        cfg = model_cfg(**overrides)
        m = shapemol_amd.ScorePosNet3D(cfg, 15)
        sdn = synth.synthetic_state_dict(cfg, seed=seed)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
        _cache[key] = m.to("cuda:0")
    return _cache[key]


def T(a, device=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device) if device else t


def maxabs(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    b = b.detach().cpu().numpy() if hasattr(b, "detach") else np.asarray(b)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.size else 0.0


def hash_noise(n, steps, seed, c=15):
    eps, u = zip(*[synth.step_noise(n, c, s, seed=seed) for s in range(steps)])
    return np.stack(eps), np.stack(u)


def record(test, **values):
    """Append the measured errors of a parity test to gpurun_out/parity_errors.jsonl (copied to profiles/ per round),
    so that drift between rounds is visible even while the assertions stay green."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_errors.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=test, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in values.items()})) + "\n")
    except OSError:
        pass
