#!/usr/bin/env python3
"""Developer tool: in-kernel phase timeline of one launch (needs the --stamps build).
    SHAPEMOL_STAMPS=1 python tools/kstamps.py --sel 1     # 0 node_pre, 1 edge_x2h, 2 edge_h2x"""
import argparse, os, sys
os.environ["SHAPEMOL_STAMPS"] = "1"
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd import ScorePosNet3D, synth
ap = argparse.ArgumentParser(); ap.add_argument("--sel", type=int, default=1); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--waves", type=int, default=0); ap.add_argument("--chain", type=int, default=0)
ap.add_argument("--opt", action="append", default=[]); ap.add_argument("--k", type=int, default=8); ap.add_argument("--atoms", type=str, default="")
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(ROOT, "config/training/dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")))["model"]
cfg["knn"] = a.k
m = ScorePosNet3D(cfg, 15); m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(cfg, 7).items()}); m = m.to("cuda:0")
bb = synth.synthetic_batch(a.batch, seed=2021, atoms_range=(tuple(int(x) for x in a.atoms.split(",")) if a.atoms else None))
for o in a.opt:
    k_, v_ = o.split("="); m.set_option(k_, int(v_))
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
args = (T(bb["init_pos"]), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]), T(np.full(a.batch, 500, np.int64)))
m(*args); m.set_option("edge_waves", a.waves)
for _ in range(3): m(*args)
torch.cuda.synchronize()
m.set_option("kstamp_sel", a.sel)
if a.chain > 0:      # stamp the selected launch of the LAST step of a running chain (clocks and caches as in production)
    from shapemol_amd.runtime import ChainRunner
    r = ChainRunner(m, len(bb["batch"]), a.batch, a.chain, keep_traj=False)
    r.load_batch(bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"])
    r.run(a.chain); r.synchronize()
else:
    m(*args); torch.cuda.synchronize()
st = m.debug_read("kstamps", (4096 * 16, 8), np.uint64).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
print(f"waves stamped: {len(st)}; all times in us relative to the earliest stamp 0")
names = {0: ["start", "weights requested, rows loaded + split -> LDS", "barrier", "all tiles' products + stores issued", "stores drained", "-", "-", "-"],
         1: ["start", "K image in LDS (barrier 1)", "key: rbf + gathered rows summed", "key: GEMM1 (fp32 MFMA)", "key: LayerNorm+ReLU", "key: split", "key: GEMM2+softmax+alpha stores issued", "V image swapped (2 barriers)"]}
names[1] = ["start", "both images in LDS (barrier)", "key: hidden fragments ready", "value: hidden fragments ready", "key: GEMM2 + softmax done", "value: GEMM2 + sums + stores issued", "(serial build) image DMA landed", "(serial build) nbr + x loaded"]      # sm_edge16.h
names[1][6] = "(looping launches) top of the wave's LAST job"
names[2] = names[1][:6] + ["VN-linear of the wave's atoms", "workgroup barrier (then batch sums -> atomics)"]      # h2x
names[3] = ["start", "W1 + [att|h] fragments staged (barrier)", "GEMM1 -> pre (barrier)", "normalise (barrier)", "GEMM2 + h' (barrier)", "follow GEMM1s (barrier)", "normalise x2 (barrier)", "follow GEMM2s + stores drained"]
names[4] = ["start", "span, coordinates, distances -> LDS (drained)", "barrier (weights staged)", "rank loop + neighbour row", "weight MLP: first Linear (fp32 MFMA)", "LayerNorm", "dot, sigmoid, stores drained", "-"]     # graph_kernel
names[5] = ["start", "embedding -> fragments, weights requested", "barrier", "query GEMM1 + 4H per-node products issued", "barrier", "normalise (barrier)", "query GEMM2 + stores drained", "-"]     # node_prologue16_kernel
nm = names[a.sel if a.sel in names else 1]
for k in range(8):
    col = st[:, k]; col = col[col > 0]
    if len(col) == 0: continue
    rel = (col - t0) / 100.0
    print(f"  stamp {k} {nm[k]:32s} min {rel.min():7.2f}  median {np.median(rel):7.2f}  max {rel.max():7.2f}")
if (st[:, 6] > st[:, 1]).any() and a.sel in (1, 2):      # looping launch: phases of the last job, from its own start
    ok = st[:, 6] > st[:, 1]
    for k, lab in ((2, "key hidden"), (3, "value hidden"), (4, "key GEMM2 + softmax"), (5, "value GEMM2 + stores")):
        prev = st[ok, 6] if k == 2 else st[ok, k - 1]
        print(f"  last job: {lab:24s} median {np.median((st[ok, k] - prev) / 100.0):6.2f} us")
    print(f"  last job: whole               median {np.median((st[ok, 5] - st[ok, 6]) / 100.0):6.2f} us; launch {((st[:, 5].max() - t0) / 100.0):.1f} us")
d = np.diff(st, axis=1) / 100.0
print("  median phase lengths (us):", " ".join(f"{np.median(d[:, k][st[:, k + 1] > 0]):.2f}" if (st[:, k + 1] > 0).any() else "-" for k in range(7)))
