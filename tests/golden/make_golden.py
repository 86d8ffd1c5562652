#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the reference itself.

Runs ONLY in the build container (it needs /root/reference); nothing here is imported by
the product, the tests or the bench.  It imports the reference's own ``models`` package
(SURVEY.md Appendix A), fills it with the deterministic synthetic weights of
``shapemol_amd.synth`` and records inputs/outputs of

  * ScorePosNet3D.forward              (/root/reference/models/molopt_score_model.py:286-320)
  * ScorePosNet3D.sample_diffusion     (/root/reference/models/molopt_score_model.py:533-697)
  * the per-layer taps needed to debug kernels (hooks on the reference's own sub-modules)
  * the state-dict key/shape list and the schedule tables.

Third-party packages that the reference imports but that are absent offline are supplied as
minimal stand-ins implementing their *published* semantics (torch_scatter 2.0.9,
torch_geometric 2.3.0 knn_graph, easydict); these are the "parity unpinned" boundary named in
DESIGN.md -- everything inside the reference's own files is executed as shipped.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import os
import sys
import json
import types
import contextlib

sys.dont_write_bytecode = True
os.environ.setdefault("TQDM_DISABLE", "1")
import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
from shapemol_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------------------
# stand-ins for absent third-party packages (published semantics only)
# ----------------------------------------------------------------------------------------
class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def _bcast(index, src, dim):
    shape = [1] * src.dim()
    shape[dim] = -1
    return index.view(shape).expand_as(src)


def scatter_sum(src, index, dim=0, out=None, dim_size=None):
    n = int(index.max()) + 1 if dim_size is None else dim_size
    shape = list(src.shape)
    shape[dim] = n
    return torch.zeros(shape, dtype=src.dtype).index_add_(dim, index, src)


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    s = scatter_sum(src, index, dim, dim_size=dim_size)
    cnt = scatter_sum(torch.ones_like(index, dtype=src.dtype), index, 0, dim_size=s.shape[dim]).clamp(min=1)
    shape = [1] * s.dim()
    shape[dim] = -1
    return s / cnt.view(shape)


def scatter_softmax(src, index, dim=0, eps=1e-12):
    idx = _bcast(index, src, dim)
    n = int(index.max()) + 1
    shape = list(src.shape)
    shape[dim] = n
    mx = torch.full(shape, float("-inf"), dtype=src.dtype).scatter_reduce(dim, idx, src, "amax", include_self=True)
    ex = (src - mx.gather(dim, idx)).exp()
    den = torch.zeros(shape, dtype=src.dtype).scatter_add_(dim, idx, ex) + eps
    return ex / den.gather(dim, idx)


def knn_graph(x, k, batch=None, loop=False, flow="source_to_target", **kw):
    """k nearest neighbours inside each batch id, self excluded; edges grouped by centre i,
    neighbours ascending by (squared distance, index); returns [src=j, dst=i].
    Squared distance = (dx*dx + dy*dy) + dz*dz with each operation rounded to float32
    (the nanoflann L2 accumulation order used by torch_cluster's CPU path)."""
    assert flow == "source_to_target" and not loop
    n = x.shape[0]
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long)
    src, dst = [], []
    counts = torch.bincount(batch)
    start = 0
    for c in counts.tolist():
        p = x[start:start + c]
        d = p[:, None, :] - p[None, :, :]
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
        d2 = d2 + d[..., 2] * d[..., 2]
        d2 = d2.clone()
        d2.fill_diagonal_(float("inf"))
        kk = min(k, c - 1)
        if kk > 0:
            order = torch.sort(d2, dim=1, stable=True)[1][:, :kk]
            dst.append(torch.arange(c).repeat_interleave(kk) + start)
            src.append(order.reshape(-1) + start)
        start += c
    if not src:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(src), torch.cat(dst)])


def install_stand_ins():
    m = types.ModuleType("torch_scatter")
    m.scatter_sum, m.scatter_mean, m.scatter_softmax = scatter_sum, scatter_mean, scatter_softmax
    sys.modules["torch_scatter"] = m
    g = types.ModuleType("torch_geometric")
    gn = types.ModuleType("torch_geometric.nn")
    gn.knn_graph = knn_graph

    def radius_graph(*a, **k):
        raise NotImplementedError("radius_graph is never called by the reference path")
    gn.radius_graph = radius_graph
    g.nn = gn
    sys.modules["torch_geometric"] = g
    sys.modules["torch_geometric.nn"] = gn
    e = types.ModuleType("easydict")
    e.EasyDict = EasyDict
    sys.modules["easydict"] = e
    sys.modules["openbabel"] = types.ModuleType("openbabel")
    sys.modules["openbabel.openbabel"] = types.ModuleType("openbabel.openbabel")
    sys.path.insert(0, REF)
    import utils  # namespace package of the reference
    cg = types.ModuleType("utils.covalent_graph")

    def connect_covalent_graph(*a, **k):
        raise NotImplementedError("unreachable under cutoff_mode: knn")
    cg.connect_covalent_graph = connect_covalent_graph
    sys.modules["utils.covalent_graph"] = cg
    utils.covalent_graph = cg


def load_reference_model(overrides=None):
    install_stand_ins()
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        from models.molopt_score_model import ScorePosNet3D
    cfg_path = os.path.join(REF, "config/training",
                            "dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")
    cfg = EasyDict(yaml.safe_load(open(cfg_path)))
    for k, v in (overrides or {}).items():
        cfg.model[k] = v
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        model = ScorePosNet3D(cfg.model, ligand_atom_feature_dim=15)
    return model, cfg  # stays in train mode, as scripts/sample_diffusion.py leaves it


def classify(key, tensor, model):
    """(kind, fan_in) for synth.fill_state_dict."""
    leaf = key.rsplit(".", 1)[-1]
    if leaf in ("running_mean", "running_var"):
        return leaf, 0
    if leaf == "num_batches_tracked":
        return "counter", 0
    if leaf == "offset" or "." not in key:
        return "const", 0
    mod = model.get_submodule(key.rsplit(".", 1)[0])
    if isinstance(mod, (torch.nn.LayerNorm, torch.nn.BatchNorm1d)):
        return ("norm_weight" if leaf == "weight" else "norm_bias"), 0
    assert isinstance(mod, torch.nn.Linear), (key, type(mod))
    return leaf, mod.in_features


def synthetic_load(model, seed):
    sd = model.state_dict()
    spec = {k: (tuple(v.shape), *classify(k, v, model)) for k, v in sd.items()}
    filled = synth.fill_state_dict(spec, seed=seed)
    new = {k: (torch.from_numpy(filled[k]) if k in filled else v.clone()) for k, v in sd.items()}
    model.load_state_dict(new, strict=True)
    return spec


@contextlib.contextmanager
def fed_noise(eps_list, u_list, record=None):
    """Replace torch.randn_like / torch.rand_like for the duration of a reference call.
    If eps_list/u_list are given they are consumed in order; otherwise torch's own RNG is
    used and the draws are recorded."""
    real_randn, real_rand = torch.randn_like, torch.rand_like
    it_e, it_u = iter(eps_list or []), iter(u_list or [])

    def randn_like(t, **k):
        r = torch.from_numpy(next(it_e)).to(t.dtype) if eps_list is not None else real_randn(t, **k)
        if record is not None:
            record["eps"].append(r.numpy().copy())
        return r

    def rand_like(t, **k):
        r = torch.from_numpy(next(it_u)).to(t.dtype) if u_list is not None else real_rand(t, **k)
        if record is not None:
            record["u"].append(r.numpy().copy())
        return r
    torch.randn_like, torch.rand_like = randn_like, rand_like
    try:
        yield
    finally:
        torch.randn_like, torch.rand_like = real_randn, real_rand


def tapped_forward(model, pos, v, batch, shape, t):
    """One reference forward with hooks on the reference's own sub-modules."""
    taps = {}
    hooks = []
    rn = model.refine_net

    def grab(name, which="out"):
        def fn(mod, inp, out):
            val = out if which == "out" else inp[0]
            taps[name] = val.detach().numpy().copy()
        return fn
    hooks.append(rn.edge_pred_layer.register_forward_hook(grab("ew_logit")))
    hooks.append(rn.invariant_shape_layer.register_forward_hook(grab("invar_shape")))
    for l, blk in enumerate(rn.base_block):
        hooks.append(blk.x2h_layers[0].register_forward_hook(grab(f"h_{l}")))
        hooks.append(blk.h2x_layers[0].register_forward_hook(grab(f"dx_{l}")))
        hooks.append(blk.h2x_layers[0].shape_linear.batchnorm.bn.register_forward_hook(grab(f"bn_in_{l}", "in")))
    import models.uni_transformer as ut
    real_knn = ut.knn_graph

    def spy(*a, **k):
        e = real_knn(*a, **k)
        taps["edge_index"] = e.numpy().copy()
        return e
    ut.knn_graph = spy
    try:
        with torch.no_grad():
            out = model(pos, v, batch, shape, time_step=t)
    finally:
        ut.knn_graph = real_knn
        for h in hooks:
            h.remove()
    taps.update({k: o.numpy().copy() for k, o in out.items()})
    return taps


def t_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    torch.set_num_threads(8)
    model, cfg = load_reference_model()
    spec = synthetic_load(model, seed=7)
    sd = model.state_dict()

    # ---- (v) state-dict layout + schedule tables -------------------------------------
    layout = [[k, list(v.shape), str(v.dtype).replace("torch.", ""), spec[k][1], int(spec[k][2])]
              for k, v in sd.items()]
    json.dump(layout, open(os.path.join(HERE, "state_dict_layout.json"), "w"), indent=0)
    sched = {k: v.numpy() for k, v in sd.items() if "." not in k}
    np.savez_compressed(os.path.join(HERE, "schedules.npz"), **sched)
    print("state dict:", len(sd), "entries,", sum(v.numel() for v in sd.values()), "elements")

    # ---- (i)+(ii) one-forward goldens with taps, B=4 ----------------------------------
    b4 = synth.synthetic_batch(4, seed=2021)
    pos, v, batch, shape = t_(b4["init_pos"]), t_(b4["init_v"]), t_(b4["batch"]), t_(b4["shape"])
    fw = dict(pos=b4["init_pos"], v=b4["init_v"], batch=b4["batch"], shape=b4["shape"])
    for name, tvals in (("t999", [999] * 4), ("t500", [500] * 4), ("t0", [0] * 4), ("tmix", [3, 250, 640, 998])):
        taps = tapped_forward(model, pos, v, batch, shape, torch.tensor(tvals))
        fw[name + "_t"] = np.asarray(tvals, np.int64)
        keep = ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v") if name != "t999" else tuple(taps)
        for k in keep:
            fw[f"{name}_{k}"] = taps[k]
    np.savez_compressed(os.path.join(HERE, "forward_b4.npz"), **fw)
    print("forward_b4: N =", len(b4["batch"]))

    # ragged / degenerate sizes: molecules with 1, 2, 5, 9 and 30 atoms (deg < k for the small ones)
    counts = np.array([1, 2, 5, 9, 30, 3], np.int64)
    n = int(counts.sum())
    rg = dict(counts=counts, batch=np.repeat(np.arange(len(counts)), counts),
              pos=synth.hash_normal((n, 3), 201, 5) * 2.0, v=(synth.hash_u24(n, 202, 5) % 15).astype(np.int64),
              shape=synth.hash_normal((len(counts), 32, 3), 203, 5), t=np.array([10, 999, 400, 0, 77, 500], np.int64))
    with torch.no_grad():
        out = model(t_(rg["pos"]), t_(rg["v"]), t_(rg["batch"]), t_(rg["shape"]), time_step=t_(rg["t"]))
    rg.update({k: o.numpy() for k, o in out.items()})
    np.savez_compressed(os.path.join(HERE, "forward_ragged.npz"), **rg)

    # ---- (iii) config-1 analogue: B=4, 50 steps, torch RNG seed 2021, draws recorded ----
    # RNG call order of the reference driver (scripts/sample_diffusion.py:34,82,93,99):
    # np.random.choice -> torch.randn(N,3) -> rand_like(N,15) [init v] -> per step randn_like, rand_like
    torch.manual_seed(2021)
    np.random.seed(2021)
    nums, p = synth.moses_atom_prior()
    counts = np.asarray(np.random.choice(nums, 4, p=p).tolist(), np.int64)
    batch = torch.repeat_interleave(torch.arange(4), torch.from_numpy(counts))
    n = int(counts.sum())
    init_pos = torch.randn(n, 3)
    rec = {"eps": [], "u": []}
    from models.molopt_score_model import log_sample_categorical
    shape4 = t_(synth.hash_normal((4, 32, 3), 103, 2021))
    with fed_noise(None, None, rec), contextlib.redirect_stdout(open(os.devnull, "w")):
        init_v = log_sample_categorical(torch.zeros(n, 15))
        r = model.sample_diffusion(init_pos, init_v, batch, shape4.view(4, -1), num_steps=50, center_pos_mode="none")
    logp = torch.stack(r["vt_traj"]).numpy()
    top2 = np.sort(logp, -1)[..., -2:]
    np.savez_compressed(
        os.path.join(HERE, "chain_b4_s50_torchrng.npz"),
        counts=counts, batch=batch.numpy(), init_pos=init_pos.numpy(), init_v=init_v.numpy(),
        init_u=rec["u"][0], shape=shape4.numpy(), eps=np.stack(rec["eps"]), u=np.stack(rec["u"][1:]),
        pos=r["pos"].numpy(), v=r["v"].numpy(), pos_traj=torch.stack(r["pos_traj"]).numpy(),
        v_traj=torch.stack(r["v_traj"]).numpy(), v0_last=r["v0_traj"][-1].numpy(), vt_last=r["vt_traj"][-1].numpy(),
        pos_cond_last=r["pos_cond_traj"][-1].numpy(), v_cond_last=r["v_cond_traj"][-1].numpy(),
        min_top2_margin=np.float32((top2[..., 1] - top2[..., 0]).min()))
    print("chain_b4_s50: N =", n, "counts", counts.tolist())

    # ---- (iv) hash-noise chains: B=4 x 1000 steps (end state) and B=16 x 100 steps ------
    for tag, B, S, seed in (("b4_s1000", 4, 1000, 11), ("b16_s100", 16, 100, 12)):
        bb = synth.synthetic_batch(B, seed=seed)
        n = len(bb["batch"])
        eps, u = zip(*[synth.step_noise(n, 15, s, seed=seed) for s in range(S)])
        with fed_noise(list(eps), list(u)), contextlib.redirect_stdout(open(os.devnull, "w")):
            r = model.sample_diffusion(t_(bb["init_pos"]), t_(bb["init_v"]), t_(bb["batch"]),
                                       t_(bb["shape"]).view(B, -1), num_steps=S, center_pos_mode="none")
        every = max(S // 20, 1)
        np.savez_compressed(
            os.path.join(HERE, f"chain_{tag}_hash.npz"), B=B, S=S, seed=seed, every=every,
            pos=r["pos"].numpy(), v=r["v"].numpy(),
            pos_traj_sub=torch.stack(r["pos_traj"][::every]).numpy(),
            v_traj_sub=torch.stack(r["v_traj"][::every]).numpy(),
            vt_last=r["vt_traj"][-1].numpy())
        print("chain", tag, "N =", n)

    # ---- (vi) reduced-width model (H=32, 4 heads, L=2) and the k=32 stress variant -------
    for tag, ov, B, rng in (("small", dict(hidden_dim=32, n_heads=4, num_layers=2), 6, None),
                            ("k32", dict(knn=32, num_layers=2), 3, (40, 80))):
        m2, _ = load_reference_model(ov)
        synthetic_load(m2, seed=9)
        bb = synth.synthetic_batch(B, seed=33, atoms_range=rng)
        tt = (synth.hash_u24(B, 77, 1) % 1000).astype(np.int64)
        with torch.no_grad():
            out = m2(t_(bb["init_pos"]), t_(bb["init_v"]), t_(bb["batch"]), t_(bb["shape"]), time_step=t_(tt))
        np.savez_compressed(os.path.join(HERE, f"forward_{tag}.npz"), overrides=json.dumps(ov), t=tt,
                            **{k: bb[k] for k in ("counts", "batch", "init_pos", "init_v", "shape")},
                            **{k: o.numpy() for k, o in out.items()})
        print("forward", tag, "N =", len(bb["batch"]))


if __name__ == "__main__":
    main()
