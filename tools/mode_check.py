#!/usr/bin/env python3
"""Quick A/B of the precision modes on one GPU: forward outputs of one mode against another (and optionally the CPU oracle) at a
given batch size.  tools/ only.

    python tools/mode_check.py --batch 256 --a edge_bf16=2,node_f16=0 --b edge_bf16=1,node_f16=0 [--oracle]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, T, hip_model, maxabs, oracle_model, synth  # noqa: E402

DEV = "cuda:0"


def opts(spec):
    return {k: int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--a", default="edge_bf16=2,node_f16=0")
    ap.add_argument("--b", default="edge_bf16=1,node_f16=0")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--knn", type=int, default=0)
    a = ap.parse_args()
    kw = dict(knn=a.knn) if a.knn else {}
    m = hip_model(**kw)
    bb = synth.synthetic_batch(a.batch, seed=13, max_atoms=38)
    t = (synth.hash_u24(a.batch, 9, 13) % 1000).astype(np.int64)
    args = (T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV), T(t, DEV))
    outs = {}
    for tag, spec in (("a", a.a), ("b", a.b)):
        o = opts(spec)
        for k, v in o.items():
            m.set_option(k, v)
        with torch.no_grad():
            outs[tag] = {k: v.cpu() for k, v in m(*args).items()}
        m.check_status()
    keys = ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v")
    print(f"B = {a.batch}, {len(bb['batch'])} atoms: [{a.a}] vs [{a.b}]:", {k: maxabs(outs['a'][k], outs['b'][k]) for k in keys}, flush=True)
    if a.oracle:
        sd, dm, _, _ = oracle_model(**kw)
        ref = O.score(sd, dm, T(bb["init_pos"]), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]), T(t))
        for tag in ("a", "b"):
            print(f"  [{a.a if tag == 'a' else a.b}] vs oracle:", {k: maxabs(outs[tag][k], ref[k]) for k in keys}, flush=True)


if __name__ == "__main__":
    main()
