#!/usr/bin/env python3
"""Round-2 golden fixtures at the BASELINE.json configuration sizes, from the reference itself.

Same harness as make_golden.py (the reference's own ``models`` package is imported in the build
container and filled with the hash weights of ``shapemol_amd.synth``); kept in a second script
because these take tens of CPU-minutes while make_golden.py's set regenerates in a few.

    python tests/golden/make_golden_r2.py b256      # configs[1]: B=256 x 1000 steps   (20-47 min)
    python tests/golden/make_golden_r2.py b1024     # configs[2]: B=1024 x 50 steps    (~6 min)
    python tests/golden/make_golden_r2.py k32       # configs[4]: <=80 atoms, k=32, L=8, B=64
    python tests/golden/make_golden_r2.py guide     # point-cloud shape guidance: the function alone and inside a chain
    python tests/golden/make_golden_r2.py se        # frozen shape encoder (VN_DGCNN_Encoder), 3 clouds of 512 points
    python tests/golden/make_golden_r2.py loss      # get_diffusion_loss as validate() calls it, module in eval and in train mode
    python tests/golden/make_golden_r2.py all
    python tests/golden/make_golden_r2.py grad          # round 3: gradients of the training loss (B = 12)
    python tests/golden/make_golden_r2.py retall        # round 3: forward(return_all=True)
    python tests/golden/make_golden_r2.py b256_tail     # round 3: the last 49 steps of the b256 chain in 10-step snapshots (minutes)
    python tests/golden/make_golden_r2.py b1024_s1000   # round 3: configs[2]/[3] per-GPU batch at full length (~3 CPU-hours)
    python tests/golden/make_golden_r2.py center        # round 3: sample_diffusion(center_pos_mode='center') on off-centre molecules (seconds)
    python tests/golden/make_golden_r2.py b256_pins     # round 4: the b256 chain again, recording the reference's neighbour lists where the k-th / (k+1)-th choice is fragile
    python tests/golden/make_golden_r2.py b1024_pins    # round 4: the same for the B = 1024 x 1000 chain (~3.5 CPU-hours)

Noise is the hash noise of synth.step_noise (a pure function of (seed, step)), so the fixtures
hold only the states: end pos / v, snapshots, and the first steps (so that the CPU suite can
check the oracle against a few steps without a half-hour run).
Reference entry points: /root/reference/models/molopt_score_model.py:286-320 (forward),
:533-697 (sample_diffusion).
"""
import contextlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from shapemol_amd import synth  # noqa: E402

t_ = G.t_


def chain(model, tag, B, S, seed, every, head, atoms_range=None, max_atoms=None, tail=None):
    """tail=(first, every): also keep the states after reverse steps first, first + every, ... (finer windows at the end of a
    full-length chain, where the posterior passes the network's x0 estimate through almost unchanged)."""
    bb = synth.synthetic_batch(B, seed=seed, atoms_range=atoms_range, max_atoms=max_atoms)
    n = len(bb["batch"])
    eps, u = zip(*[synth.step_noise(n, 15, s, seed=seed) for s in range(S)])
    t0 = time.time()
    with G.fed_noise(list(eps), list(u)), contextlib.redirect_stdout(open(os.devnull, "w")):
        r = model.sample_diffusion(t_(bb["init_pos"]), t_(bb["init_v"]), t_(bb["batch"]),
                                   t_(bb["shape"]).view(B, -1), num_steps=S, center_pos_mode="none")
    del eps, u
    pos_traj = torch.stack(r["pos_traj"]).numpy()
    v_traj = torch.stack(r["v_traj"]).numpy()
    extra = {}
    if tail is not None:
        extra = dict(tail_first=tail[0], tail_every=tail[1], pos_traj_tail=pos_traj[tail[0]::tail[1]],
                     v_traj_tail=v_traj[tail[0]::tail[1]].astype(np.int8))
    np.savez_compressed(
        os.path.join(HERE, f"chain_{tag}_hash.npz"), B=B, S=S, seed=seed, every=every, head=head,
        counts=bb["counts"], pos=r["pos"].numpy(), v=r["v"].numpy(),
        pos_traj_sub=pos_traj[::every], v_traj_sub=v_traj[::every].astype(np.int8),
        pos_traj_head=pos_traj[:head], v_traj_head=v_traj[:head].astype(np.int8),
        pos0_first=r["pos_cond_traj"][0].numpy(), v0_first=r["v0_traj"][0].numpy(),
        vt_last=r["vt_traj"][-1].numpy(), **extra)
    print(f"chain {tag}: N = {n}, {S} steps in {time.time() - t0:.0f} s", flush=True)


def center_fixture(model):
    """sample_diffusion(center_pos_mode='center') (molopt_score_model.py:52-60,547,675-684) on molecules moved away from the
    origin: 6 molecules, 30 steps, hash noise; every trajectory kept (small)."""
    B, S, seed = 6, 30, 41
    bb = synth.synthetic_batch(B, seed=seed)
    n = len(bb["batch"])
    shift = (synth.hash_normal((B, 3), 911, seed) * 4.0).astype(np.float32)
    init = (bb["init_pos"] + shift[bb["batch"]]).astype(np.float32)
    eps, u = zip(*[synth.step_noise(n, 15, s, seed=seed) for s in range(S)])
    with G.fed_noise(list(eps), list(u)), contextlib.redirect_stdout(open(os.devnull, "w")):
        r = model.sample_diffusion(t_(init), t_(bb["init_v"]), t_(bb["batch"]), t_(bb["shape"]).view(B, -1), num_steps=S,
                                   center_pos_mode="center")
    np.savez_compressed(os.path.join(HERE, "chain_center_b6_s30_hash.npz"), B=B, S=S, seed=seed, init_pos=init, pos=r["pos"].numpy(),
                        v=r["v"].numpy(), pos_traj=torch.stack(r["pos_traj"]).numpy(), v_traj=torch.stack(r["v_traj"]).numpy().astype(np.int8),
                        pos_cond_traj=torch.stack(r["pos_cond_traj"]).numpy())
    print(f"center: N = {n}, {S} steps, |pos| max {float(r['pos'].abs().max()):.2f}", flush=True)


def chain_tail(model, src_tag, tag, max_atoms=None, every=10):
    """The last window of a committed full-length chain in finer snapshots WITHOUT re-running the whole chain: the reference's
    loop runs t = num_timesteps - 1 ... num_timesteps - num_steps (models/molopt_score_model.py:558) and indexes every
    schedule table and the time embedding by t itself, so with ``model.num_timesteps`` lowered to the number of steps
    left it runs exactly the chain's last steps from the committed state.  Self-check: the end state must reproduce the
    full chain's end state bit for bit."""
    c = np.load(os.path.join(HERE, f"chain_{src_tag}_hash.npz"))
    B, S, seed, ev = int(c["B"]), int(c["S"]), int(c["seed"]), int(c["every"])
    bb = synth.synthetic_batch(B, seed=seed, max_atoms=max_atoms)
    n = len(bb["batch"])
    j = len(c["pos_traj_sub"]) - 1
    done = j * ev + 1                                   # reverse steps 0 .. j * ev are behind the last snapshot
    left = S - done
    eps, u = zip(*[synth.step_noise(n, 15, s, seed=seed) for s in range(done, S)])
    T_full = model.num_timesteps
    model.num_timesteps = left
    try:
        with G.fed_noise(list(eps), list(u)), contextlib.redirect_stdout(open(os.devnull, "w")):
            r = model.sample_diffusion(t_(c["pos_traj_sub"][j]), t_(c["v_traj_sub"][j].astype(np.int64)), t_(bb["batch"]),
                                       t_(bb["shape"]).view(B, -1), num_steps=left, center_pos_mode="none")
    finally:
        model.num_timesteps = T_full
    same = np.array_equal(r["pos"].numpy(), c["pos"]) and np.array_equal(r["v"].numpy(), c["v"])
    print(f"chain_tail {tag}: steps {done}..{S - 1}; end state reproduces the full chain bit for bit: {same}", flush=True)
    assert same, "the resumed tail must reproduce the committed end state (same BLAS thread count as the full chain?)"
    pos_traj = torch.stack(r["pos_traj"]).numpy()
    v_traj = torch.stack(r["v_traj"]).numpy()
    # state after reverse step done + i is pos_traj[i]; keep the steps that are multiples of `every`
    first = (-done) % every
    np.savez_compressed(os.path.join(HERE, f"chain_{tag}_hash.npz"), src=src_tag, first_step=done + first, every=every,
                        pos_traj_tail=pos_traj[first::every], v_traj_tail=v_traj[first::every].astype(np.int8))


def chain_pins(model, src_tag, tag, max_atoms=None, thr=5e-4):
    """Re-run a committed full-length chain and record the reference's kNN choice wherever it is FRAGILE: centre atoms whose
    k-th and (k+1)-th candidates lie within a relative margin `thr` in squared distance (models/uni_transformer.py:446-473;
    one graph per score evaluation, :499).  A second float32 implementation whose coordinates differ in the last bits picks
    the other candidate there and leaves the reference's trajectory for good; with the choice pinned
    at exactly these (step, atom) pairs a free-running chain can be held to the reference over all 1000 steps.  Sparse table
    (step, atom, nbr[k], margin): KBs.  Self-check: the end state must reproduce the committed chain bit for bit."""
    import models.uni_transformer as ut
    c = np.load(os.path.join(HERE, f"chain_{src_tag}_hash.npz"))
    B, S, seed = int(c["B"]), int(c["S"]), int(c["seed"])
    bb = synth.synthetic_batch(B, seed=seed, max_atoms=max_atoms)
    n = len(bb["batch"])
    counts = [int(x) for x in bb["counts"]]
    real_knn = ut.knn_graph
    rec = dict(step=[], atom=[], nbr=[], margin=[], calls=0, min_margin=[])

    def spy(x, k, batch=None, **kw):
        e = real_knn(x, k=k, batch=batch, **kw)
        step, start, lo = rec["calls"], 0, np.inf
        for cnt in counts:
            if cnt - 1 > k:
                p = x[start:start + cnt]
                d = p[:, None, :] - p[None, :, :]
                d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]        # the stand-in's arithmetic (make_golden.knn_graph)
                d2 = (d2 + d[..., 2] * d[..., 2]).clone()
                d2.fill_diagonal_(float("inf"))
                val, order = torch.sort(d2, dim=1, stable=True)
                m = ((val[:, k] - val[:, k - 1]) / val[:, k - 1]).numpy()
                lo = min(lo, float(m.min()))
                for a in np.nonzero(m < thr)[0]:
                    rec["step"].append(step)
                    rec["atom"].append(start + int(a))
                    rec["nbr"].append((order[a, :k] + start).numpy().astype(np.int32))
                    rec["margin"].append(float(m[a]))
            start += cnt
        rec["min_margin"].append(lo)
        rec["calls"] += 1
        return e

    eps, u = zip(*[synth.step_noise(n, 15, s, seed=seed) for s in range(S)])
    t0 = time.time()
    ut.knn_graph = spy
    try:
        with G.fed_noise(list(eps), list(u)), contextlib.redirect_stdout(open(os.devnull, "w")):
            r = model.sample_diffusion(t_(bb["init_pos"]), t_(bb["init_v"]), t_(bb["batch"]),
                                       t_(bb["shape"]).view(B, -1), num_steps=S, center_pos_mode="none")
    finally:
        ut.knn_graph = real_knn
    same = np.array_equal(r["pos"].numpy(), c["pos"]) and np.array_equal(r["v"].numpy(), c["v"])
    print(f"chain_pins {tag}: {rec['calls']} graphs, {len(rec['step'])} fragile (step, atom) pairs below {thr:g}, "
          f"{time.time() - t0:.0f} s; end state reproduces the committed chain bit for bit: {same}", flush=True)
    # the pins are recorded against THIS run's edges: verify the pinned rows are what the reference's graph held
    np.savez_compressed(os.path.join(HERE, f"chain_{tag}_pins.npz"), src=src_tag, thr=thr, reproduces_committed_chain=same,
                        step=np.asarray(rec["step"], np.int16), atom=np.asarray(rec["atom"], np.int16),
                        nbr=np.stack(rec["nbr"]).astype(np.int16) if rec["nbr"] else np.zeros((0, 8), np.int16),
                        margin=np.asarray(rec["margin"], np.float32), min_margin_per_step=np.asarray(rec["min_margin"], np.float32))
    assert same, "the re-run must reproduce the committed end state (same BLAS thread count as the full chain?)"


class GuideRecorder:
    """Stands in for the sklearn KD-tree handed to the reference's pointcloud_shape_guidance
    (models/molopt_score_model.py:699-740) and for np.random.random during the call: answers every query with a real
    KDTree and mirrors the function's control flow to learn WHICH atoms receive each uniform draw (the function draws
    one value per currently-far atom, in index order).  Yields the dense table draws[step][iteration][atom]."""

    def __init__(self, cloud, radius, n_atoms):
        from sklearn.neighbors import KDTree
        self.tree, self.radius, self.n = KDTree(cloud), radius, n_atoms
        self.steps, self.far, self.it, self.fresh = [], None, 0, True

    def query(self, x, k=3):
        d, i = self.tree.query(x, k=k)
        m = d.mean(1)
        if self.fresh:                       # first query of a call: all atoms
            assert len(x) == self.n
            self.steps.append(np.full((5, self.n), 0.5))
            self.far, self.it = np.where(m > self.radius)[0], 0
            self.fresh = len(self.far) == 0
        else:                                # re-check of the pulled atoms
            self.far = self.far[~(m < self.radius)]
            self.it += 1
            self.fresh = len(self.far) == 0 or self.it == 5
        return d, i

    def random(self, n):
        assert not self.fresh and n == len(self.far)
        u = self._rs.random_sample(n)
        self.steps[-1][self.it, self.far] = u
        return u

    @contextlib.contextmanager
    def active(self, seed):
        self._rs = np.random.RandomState(seed)
        real_random, real_cuda = np.random.random, torch.Tensor.cuda
        np.random.random = self.random
        torch.Tensor.cuda = lambda t, *a, **k: t          # the reference hard-codes .cuda() (:738); this harness runs on the CPU
        try:
            yield self
        finally:
            np.random.random, torch.Tensor.cuda = real_random, real_cuda


def guidance_fixtures():
    from models.molopt_score_model import pointcloud_shape_guidance
    cloud = (synth.hash_normal((512, 3), 301, 5) * 1.2).astype(np.float64)
    radius = 0.2
    # (A) the function alone: atoms scattered around and beyond the cloud
    n = 300
    pred = (synth.hash_normal((n, 3), 302, 5) * 1.6).astype(np.float32)
    rec = GuideRecorder(cloud, radius, n)
    with rec.active(77):
        out = pointcloud_shape_guidance((cloud, rec, radius), torch.from_numpy(pred.copy()))
    np.savez_compressed(os.path.join(HERE, "guidance_fn.npz"), cloud=cloud, radius=radius, pred=pred, out=out.numpy(),
                        draws=rec.steps[0])
    print("guidance_fn: moved atoms", int((out.numpy() != pred).any(1).sum()), "of", n, flush=True)
    # (B) inside a chain: B = 4, 20 reverse steps, guided while t > 990
    model, _ = G.load_reference_model()
    G.synthetic_load(model, seed=7)
    B, S, seed, grad_step = 4, 20, 21, 990
    bb = synth.synthetic_batch(B, seed=seed)
    na = len(bb["batch"])
    eps, u = zip(*[synth.step_noise(na, 15, s, seed=seed) for s in range(S)])
    rec = GuideRecorder(cloud, radius, na)
    with G.fed_noise(list(eps), list(u)), rec.active(78), contextlib.redirect_stdout(open(os.devnull, "w")):
        r = model.sample_diffusion(t_(bb["init_pos"]), t_(bb["init_v"]), t_(bb["batch"]), t_(bb["shape"]).view(B, -1), num_steps=S,
                                   center_pos_mode="none", use_pointcloud_data=(cloud, rec, radius), grad_step=grad_step)
    draws = np.full((S, 5, na), 0.5)
    draws[:len(rec.steps)] = np.stack(rec.steps)          # guided steps come first (t = 999 ... grad_step + 1)
    np.savez_compressed(os.path.join(HERE, "chain_guided_b4_s20.npz"), B=B, S=S, seed=seed, grad_step=grad_step, cloud=cloud,
                        radius=radius, draws=draws, guided_steps=len(rec.steps), pos=r["pos"].numpy(), v=r["v"].numpy(),
                        pos_traj=torch.stack(r["pos_traj"]).numpy(), v_traj=torch.stack(r["v_traj"]).numpy(),
                        pos_cond_traj=torch.stack(r["pos_cond_traj"]).numpy())
    print("chain_guided: guided steps", len(rec.steps), "of", S, flush=True)


def shape_encoder_fixture():
    from models.shape_pointcloud_modelAE import VN_DGCNN_Encoder
    hidden, latent, layers, k = 128, 32, 4, 20                     # config.model of trained_models/se_model.pt
    enc = VN_DGCNN_Encoder(hidden, latent, layers, k)              # stays in train mode, as utils/shape.py:226-238 leaves it
    sd = synth.shape_encoder_state_dict(hidden, latent, layers, seed=17)
    def load(mod, prefix):
        mod.map_to_feat.weight.data = t_(sd[prefix + ".map_to_feat.weight"])
        mod.batchnorm.bn.weight.data = t_(sd[prefix + ".batchnorm.bn.weight"])
        mod.batchnorm.bn.bias.data = t_(sd[prefix + ".batchnorm.bn.bias"])
        mod.map_to_dir.weight.data = t_(sd[prefix + ".map_to_dir.weight"])
    load(enc.conv_pos, "conv_pos")
    for i, blk in enumerate(enc.blocks):
        load(blk, f"blocks.{i}")
    load(enc.conv_c, "conv_c")
    B, N = 3, 512
    pts = (synth.hash_normal((B, N, 3), 401, 9) * np.array([1.5, 1.0, 0.6], np.float32)).astype(np.float32)   # anisotropic blobs
    with torch.no_grad():
        z = enc(t_(pts).unsqueeze(1))
    np.savez_compressed(os.path.join(HERE, "shape_encoder.npz"), points=pts, latent=z.numpy(), hidden=hidden, latent_dim=latent,
                        layers=layers, k=k, seed=17)
    print("shape_encoder:", tuple(z.shape), "max |z|", float(z.abs().max()), flush=True)


@contextlib.contextmanager
def fed_normal(draws):
    """Replace Tensor.normal_() (the in-place draw of get_diffusion_loss, molopt_score_model.py:461) by given values."""
    real = torch.Tensor.normal_
    it = iter(draws)
    def normal_(t, *a, **k):
        t.copy_(torch.from_numpy(next(it)).to(t.dtype))
        return t
    torch.Tensor.normal_ = normal_
    try:
        yield
    finally:
        torch.Tensor.normal_ = real


def loss_fixture():
    """get_diffusion_loss(eval_mode=True, time_step=...) as validate() calls it (scripts/train_diffusion.py:168-192): once
    with the module in eval mode (running batch-norm statistics, set to non-trivial values) and once in train mode."""
    model, _ = G.load_reference_model()
    G.synthetic_load(model, seed=7)
    rs = synth.running_stats(len(model.refine_net.base_block), 16, seed=23)
    sd = model.state_dict()
    for k, v in rs.items():
        assert k in sd, k
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd, strict=True)
    B, seed = 12, 41
    bb = synth.synthetic_batch(B, seed=seed)
    n = len(bb["batch"])
    tt = (synth.hash_u24(B, 88, 3) % 1000).astype(np.int64)
    tt[0], tt[1], tt[2] = 0, 0, 999              # the decoder-NLL branch (t = 0) and the last step
    pos0 = (synth.hash_normal((n, 3), 501, seed) * 1.5).astype(np.float32)     # "clean" molecules
    noise = synth.hash_normal((n, 3), 502, seed)
    u = synth.hash_uniform((n, 15), 503, seed)
    rec = {}
    for mode in ("eval", "train"):
        model.eval() if mode == "eval" else model.train()
        with torch.no_grad(), fed_normal([noise]), G.fed_noise(None, [u]), contextlib.redirect_stdout(open(os.devnull, "w")):
            r = model.get_diffusion_loss(t_(pos0), t_(bb["init_v"]), t_(bb["batch"]), t_(bb["shape"]).view(B, -1), time_step=t_(tt), eval_mode=True)
        for k in ("loss_pos", "loss_v", "loss", "ligand_pos_perturbed", "ligand_v_perturbed", "pred_ligand_pos", "pred_ligand_v", "ligand_v_recon"):
            rec[f"{mode}_{k}"] = r[k].numpy()
        print(f"loss[{mode}]: loss {float(r['loss']):.6f}  pos {float(r['loss_pos']):.6f}  v {float(r['loss_v']):.6f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "diffusion_loss_b12.npz"), B=B, seed=seed, t=tt, counts=bb["counts"], pos=pos0,
                        running_stats_seed=23, **rec)


def grad_sample_index(key, numel, k=48):
    """The entries of a parameter's gradient kept in the fixture: k positions drawn by a hash of the key (all of them for
    tensors of up to 256 entries).  Shared with tests/ (imported from here)."""
    if numel <= 256:
        return np.arange(numel)
    import zlib
    return np.sort(np.unique(synth.hash_u24(k, zlib.crc32(key.encode()) % 100003, 61) % numel))


def grad_fixture():
    """The training step's backward (scripts/train_diffusion.py:135-147: results['loss'].backward() in train mode): gradients
    of the loss of diffusion_loss_b12.npz's inputs (train-mode batch-norm) with respect to every parameter.  2.66 M values are
    not a fixture: per parameter the L2 norm, the sum and a hashed sample of entries (everything for tensors <= 256)."""
    model, _ = G.load_reference_model()
    G.synthetic_load(model, seed=7)
    model.train()
    f = np.load(os.path.join(HERE, "diffusion_loss_b12.npz"))
    B, seed = int(f["B"]), int(f["seed"])
    bb = synth.synthetic_batch(B, seed=seed)
    n = len(bb["batch"])
    noise = synth.hash_normal((n, 3), 502, seed)
    u = synth.hash_uniform((n, 15), 503, seed)
    with fed_normal([noise]), G.fed_noise(None, [u]), contextlib.redirect_stdout(open(os.devnull, "w")):
        r = model.get_diffusion_loss(t_(f["pos"]), t_(bb["init_v"]), t_(bb["batch"]), t_(bb["shape"]).view(B, -1), time_step=t_(f["t"]), eval_mode=True)
    assert abs(float(r["loss"]) - float(f["train_loss"])) < 1e-6 * abs(float(f["train_loss"]))     # the committed forward
    r["loss"].backward()
    rec, total = {}, 0.0
    names = []
    for k, p in model.named_parameters():
        names.append(k)
        if p.grad is None:
            rec[f"has_{k}"] = False
            continue
        g = p.grad.detach().double().numpy().reshape(-1)
        idx = grad_sample_index(k, g.size)
        rec[f"has_{k}"] = True
        rec[f"norm_{k}"] = float(np.sqrt((g * g).sum()))
        rec[f"sum_{k}"] = float(g.sum())
        rec[f"val_{k}"] = g[idx].astype(np.float32)
        total += float((g * g).sum())
    np.savez_compressed(os.path.join(HERE, "grad_b12.npz"), loss=float(r["loss"]), loss_pos=float(r["loss_pos"]), loss_v=float(r["loss_v"]),
                        total_grad_norm=float(np.sqrt(total)), names=np.array(names), **rec)
    print(f"grad: loss {float(r['loss']):.6f}, {len(names)} parameters, {sum(1 for k in names if rec['has_' + k])} with gradients, "
          f"total norm {np.sqrt(total):.6f}", flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    # float32 reductions of the CPU BLAS depend on the thread count: every fixture records the count it was made with
    # (the 47-minute B=256 chain ran on 6 threads beside a build, the k=32 set on 3), and a regeneration uses the same
    threads = {"b256_pins": 6, "b1024_pins": 5, "b256": 6, "b1024": 8, "k32": 3, "guide": 8, "se": 8, "loss": 8, "b256_tail": 6, "b1024_s1000": 5, "b512_k32": 6, "grad": 8}
    def use_threads(task):
        torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", threads[task])))
    torch.set_num_threads(8)
    if what in ("b256", "b1024", "b256_tail", "b1024_s1000", "b256_pins", "b1024_pins", "all"):
        model, _ = G.load_reference_model()
        G.synthetic_load(model, seed=7)
        if what == "b256_pins":          # (not part of "all": a second 20-47 minute run of the b256 chain) round 4
            use_threads("b256_pins")
            chain_pins(model, "b256_s1000", "b256_s1000", max_atoms=38)
        if what == "b1024_pins":         # (not part of "all": ~3.5 CPU-hours)
            use_threads("b1024_pins")
            chain_pins(model, "b1024_s1000", "b1024_s1000", max_atoms=38)
        if what == "b256_tail":          # (not part of "all": derived from the committed b256 chain, minutes)
            use_threads("b256_tail")
            chain_tail(model, "b256_s1000", "b256_s1000_tail", max_atoms=38)
        if what == "b1024_s1000":        # (not part of "all": ~3 CPU-hours) configs[2]/[3] at full length
            use_threads("b1024_s1000")
            chain(model, "b1024_s1000", 1024, 1000, 15, every=50, head=2, max_atoms=38, tail=(960, 10))
        if what in ("b1024", "all"):
            use_threads("b1024")
            chain(model, "b1024_s50", 1024, 50, 14, every=10, head=2, max_atoms=38)
        if what in ("b256", "all"):
            use_threads("b256")
            chain(model, "b256_s1000", 256, 1000, 13, every=50, head=4, max_atoms=38)
    if what in ("center", "all"):        # round 3
        torch.set_num_threads(8)
        model, _ = G.load_reference_model()
        G.synthetic_load(model, seed=7)
        center_fixture(model)
    if what in ("loss", "all"):
        use_threads("loss")
        loss_fixture()
    if what in ("retall", "all"):        # round 3: forward(..., return_all=True) on the inputs of forward_b4.npz (molopt_score_model.py:312-319)
        torch.set_num_threads(8)
        model, _ = G.load_reference_model()
        G.synthetic_load(model, seed=7)
        f = np.load(os.path.join(HERE, "forward_b4.npz"))
        with torch.no_grad():
            out = model(t_(f["pos"]), t_(f["v"]), t_(f["batch"]), t_(f["shape"]), time_step=t_(f["tmix_t"]), return_all=True)
        assert np.array_equal(out["pred_ligand_pos"].numpy(), f["tmix_pred_ligand_pos"])        # same evaluation as the committed fixture
        np.savez_compressed(os.path.join(HERE, "forward_b4_return_all.npz"), n_layer_entries=len(out["layer_pred_ligand_pos"]),
                            **{f"layer_pos_{i}": x.numpy() for i, x in enumerate(out["layer_pred_ligand_pos"])},
                            **{f"layer_v_{i}": x.numpy() for i, x in enumerate(out["layer_pred_ligand_v"])})
        print("return_all:", len(out["layer_pred_ligand_pos"]), "entries", flush=True)
    if what in ("grad", "all"):          # round 3: the backward of the training step
        use_threads("grad")
        grad_fixture()
    if what in ("se", "all"):
        use_threads("se")
        G.install_stand_ins()
        shape_encoder_fixture()
    if what in ("guide", "all"):
        use_threads("guide")
        G.install_stand_ins()
        guidance_fixtures()
    if what in ("k32", "all"):
        use_threads("k32")
        # configs[4] analogue: 40-80 atom molecules, knn = 32, full depth
        m2, _ = G.load_reference_model(dict(knn=32))
        G.synthetic_load(m2, seed=9)
        B = 64
        bb = synth.synthetic_batch(B, seed=35, atoms_range=(40, 80))
        tt = (synth.hash_u24(B, 78, 1) % 1000).astype(np.int64)
        with torch.no_grad():
            out = m2(t_(bb["init_pos"]), t_(bb["init_v"]), t_(bb["batch"]), t_(bb["shape"]), time_step=t_(tt))
        np.savez_compressed(os.path.join(HERE, "forward_k32_b64.npz"), t=tt, counts=bb["counts"],
                            **{k: o.numpy() for k, o in out.items()})
        print("forward k32_b64: N =", len(bb["batch"]), flush=True)
        chain(m2, "k32_b64_s20", B, 20, 36, every=5, head=2, atoms_range=(40, 80))


if __name__ == "__main__":
    main()
