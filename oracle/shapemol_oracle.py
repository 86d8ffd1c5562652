"""CPU ORACLE (test infrastructure, NOT product code) for the ShapeMol denoising hot path.

A functional torch-CPU float32 restatement of the reference algorithm, written against a
plain ``{key: tensor}`` state dict.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product (``shapemol_amd``)
never does and fails loudly when its HIP library is missing.

Pinned against the reference: ``tests/golden/*.npz`` were produced by importing the
reference's own ``models`` package in the build container (``tests/golden/make_golden.py``)
and ``tests/test_oracle_golden.py`` checks this file against them.  The third-party ops the
reference calls (torch_scatter 2.0.9 segment softmax / sum, torch_cluster 1.6.0 kNN through
torch_geometric 2.3.0) are absent offline and restated from their published semantics:
"parity unpinned" at that boundary only.

Reference map (paths relative to /root/reference):
  score()              models/molopt_score_model.py:286-320   ScorePosNet3D.forward
  _time_embedding()    models/molopt_score_model.py:154-166,247-252
  _refine()            models/uni_transformer.py:483-540       UniTransformerO2TwoUpdateGeneral.forward
  _invariant_shape()   models/uni_transformer.py:181-189
  knn_edges()          models/uni_transformer.py:466-468 (+ torch_geometric knn_graph semantics)
  _edge_weight()       models/uni_transformer.py:475-481
  _rbf()               models/common.py:19-28
  _mlp()               models/common.py:47-67
  _x2h()               models/uni_transformer.py:48-90
  _h2x()               models/uni_transformer.py:121-162
  _vn_linear_lrelu()   models/shape_vn_layers.py:95-110 with VNBatchNorm :50-61 (train-mode statistics)
  posterior / sampling models/molopt_score_model.py:64-68,98-113,323-404,533-697
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

RBF_CENTRES = (0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10)
VN_EPS = 1e-6
LEAK = 0.2


class Dims:
    """Model dimensions read from the `model` section of the training YAML."""

    def __init__(self, cfg, num_classes=15):
        g = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        self.H = int(g("hidden_dim"))
        self.heads = int(g("n_heads"))
        self.L = int(g("num_layers"))
        self.k = int(g("knn"))
        self.G = int(g("num_r_gaussian"))
        self.S = int(g("shape_dim"))
        self.temb = int(g("time_emb_dim"))
        self.C = int(num_classes)
        self.T = int(g("num_diffusion_timesteps"))
        assert int(g("num_blocks")) == 1 and int(g("edge_feat_dim")) == 0
        assert g("cutoff_mode") == "knn" and g("ew_net_type") == "global"
        assert g("v_mode") == "uniform" and g("shape_mode", "attention_residue") == "attention_residue"


# ------------------------------------------------------------------------------------------
# building blocks
# ------------------------------------------------------------------------------------------
def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _mlp(sd, p, x):
    """Linear -> LayerNorm(eps 1e-5) -> ReLU -> Linear."""
    y = _lin(sd, p + ".net.0", x)
    y = F.layer_norm(y, (y.shape[-1],), sd[p + ".net.1.weight"], sd[p + ".net.1.bias"], 1e-5)
    return _lin(sd, p + ".net.3", torch.relu(y))


def _rbf(d):
    """exp(-0.5 (d - mu_g)^2 / (mu_1 - mu_0)^2) over the 20 fixed centres; d: (E,1) or (E,)."""
    mu = torch.tensor(RBF_CENTRES, dtype=torch.float32)
    coeff = -0.5 / float(mu[1] - mu[0]) ** 2
    return torch.exp(coeff * (d.reshape(-1, 1) - mu.view(1, -1)) ** 2)


def knn_edges(x, batch, k):
    """Per-molecule k nearest neighbours, self excluded.  Returns (src=j, dst=i), grouped by
    centre i; inside a group ascending by (squared distance, index).  The squared distance is
    (dx*dx + dy*dy) + dz*dz with every operation rounded to float32."""
    counts = torch.bincount(batch).tolist()
    src, dst, start = [], [], 0
    for c in counts:
        p = x[start:start + c]
        d = p[:, None, :] - p[None, :, :]
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
        d2 = (d2 + d[..., 2] * d[..., 2]).clone()
        d2.fill_diagonal_(float("inf"))
        kk = min(k, c - 1)
        if kk > 0:
            nb = torch.sort(d2, dim=1, stable=True)[1][:, :kk]
            src.append(nb.reshape(-1) + start)
            dst.append(torch.arange(c).repeat_interleave(kk) + start)
        start += c
    if not src:
        z = torch.zeros(0, dtype=torch.long)
        return z, z
    return torch.cat(src), torch.cat(dst)


def _segment_softmax(logit, dst, n):
    """softmax over the incoming edges of each centre, independently per trailing column."""
    idx = dst.view(-1, 1).expand_as(logit)
    mx = torch.full((n, logit.shape[1]), float("-inf")).scatter_reduce(0, idx, logit, "amax", include_self=True)
    ex = torch.exp(logit - mx[dst])
    den = torch.zeros(n, logit.shape[1]).index_add_(0, dst, ex)
    return ex / den[dst]


def _segment_sum(val, dst, n):
    return torch.zeros((n,) + tuple(val.shape[1:])).index_add_(0, dst, val)


def _time_embedding(sd, dm, t):
    half = dm.temb // 2
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    arg = t[:, None] * freq[None, :]
    e = torch.cat((arg.sin(), arg.cos()), dim=-1)
    return _lin(sd, "time_emb.3", F.silu(_lin(sd, "time_emb.1", e)))


def _invariant_shape(sd, shape):
    m = shape.mean(dim=1)
    m = m / ((m * m).sum(-1, keepdim=True) + VN_EPS)
    inv = torch.einsum("bij,bj->bi", shape, m)
    return _mlp(sd, "refine_net.invariant_shape_layer.hidden_layer", inv)


def _edge_weight(sd, x, src, dst):
    d = torch.norm(x[dst] - x[src], p=2, dim=-1, keepdim=True)
    return torch.sigmoid(_mlp(sd, "refine_net.edge_pred_layer", _rbf(d)))


def _attention(q, k, dst, n, heads):
    """alpha[e, head] for q: (N,H) queries at the centre, k: (E,H) keys on the edges."""
    dh = q.shape[1] // heads
    logit = (q[dst].view(-1, heads, dh) * k.view(-1, heads, dh) / np.sqrt(dh)).sum(-1)
    return _segment_softmax(logit, dst, n)


def _x2h(sd, p, dm, h, rfeat, src, dst, inv_atom, e_w):
    n = h.shape[0]
    kv = torch.cat([rfeat, h[dst], h[src], inv_atom[dst]], -1)
    k = _mlp(sd, p + ".hk_func", kv)
    v = _mlp(sd, p + ".hv_func", kv) * e_w.view(-1, 1)
    q = _mlp(sd, p + ".hq_func", h)
    alpha = _attention(q, k, dst, n, dm.heads)
    dh = dm.H // dm.heads
    o = _segment_sum(alpha.unsqueeze(-1) * v.view(-1, dm.heads, dh), dst, n).view(n, dm.H)
    return _mlp(sd, p + ".node_output", torch.cat([o, h], -1)) + h


def _vn_linear_lrelu(sd, p, z, taps=None, bn_eval=False):
    """z: (N, Cin, 3) -> (N, Cout, 3); vector-neuron linear, batch norm of the vector norms (train mode: biased variance
    over all N atoms; bn_eval: the running statistics, as BatchNorm1d in eval mode -- shape_vn_layers.py:50-61 under
    module.eval()), eps 1e-5, affine, VN leaky ReLU."""
    wf, wd = sd[p + ".map_to_feat.weight"], sd[p + ".map_to_dir.weight"]
    pf = torch.einsum("oc,ncd->nod", wf, z)
    nrm = torch.sqrt((pf * pf).sum(2)) + VN_EPS
    if bn_eval:
        mean, var = sd[p + ".batchnorm.bn.running_mean"], sd[p + ".batchnorm.bn.running_var"]
    else:
        mean = nrm.mean(0)
        var = ((nrm - mean) ** 2).mean(0)
    if taps is not None:
        taps["bn_mean"], taps["bn_var"], taps["bn_in"] = mean, var, nrm
    nbn = (nrm - mean) / torch.sqrt(var + 1e-5) * sd[p + ".batchnorm.bn.weight"] + sd[p + ".batchnorm.bn.bias"]
    pf = pf / nrm.unsqueeze(2) * nbn.unsqueeze(2)
    d = torch.einsum("oc,ncd->nod", wd, z)
    dot = (pf * d).sum(2, keepdim=True)
    mask = (dot >= 0).float()
    dsq = (d * d).sum(2, keepdim=True)
    return LEAK * pf + (1 - LEAK) * (mask * pf + (1 - mask) * (pf - (dot / (dsq + VN_EPS)) * d))


def _h2x(sd, p, dm, h, x, rel_x, rfeat, src, dst, inv_atom, shape_atom, e_w, taps=None, bn_eval=False):
    n = h.shape[0]
    kv = torch.cat([rfeat, h[dst], h[src], inv_atom[dst]], -1)
    k = _mlp(sd, p + ".xk_func", kv)
    v = _mlp(sd, p + ".xv_func", kv) * e_w.view(-1, 1)
    v = v.unsqueeze(-1) * rel_x.unsqueeze(1)
    q = _mlp(sd, p + ".xq_func", h)
    alpha = _attention(q, k, dst, n, dm.heads)
    o = _segment_sum(alpha.unsqueeze(-1) * v, dst, n)                      # (N, heads, 3)
    z = torch.cat((x.unsqueeze(1), o, shape_atom), dim=1)                   # (N, 1+heads+S, 3)
    res = _vn_linear_lrelu(sd, p + ".shape_linear", z, taps, bn_eval).mean(dim=1)
    return o.mean(dim=1) + res


def _refine(sd, dm, h, x, batch, shape, taps=None, bn_eval=False):
    inv_atom = _invariant_shape(sd, shape)[batch]
    shape_atom = shape[batch]
    src, dst = knn_edges(x, batch, dm.k)
    e_w = _edge_weight(sd, x, src, dst)
    if taps is not None:
        taps["edge_index"] = torch.stack([src, dst])
        taps["e_w"] = e_w
    for l in range(dm.L):
        p = f"refine_net.base_block.{l}"
        rel_x = x[dst] - x[src]
        rfeat = _rbf(torch.norm(rel_x, p=2, dim=-1, keepdim=True))
        h = _x2h(sd, p + ".x2h_layers.0", dm, h, rfeat, src, dst, inv_atom, e_w)
        lt = {} if taps is not None else None
        dx = _h2x(sd, p + ".h2x_layers.0", dm, h, x, rel_x, rfeat, src, dst, inv_atom, shape_atom, e_w, lt, bn_eval)
        x = x + dx
        if taps is not None:
            taps[f"h_{l}"], taps[f"dx_{l}"] = h, dx
            taps[f"bn_in_{l}"] = lt["bn_in"]
    return h, x


def score_with_grad(sd, dm, pos, v, batch, shape, t, taps=None, bn_eval=False):
    """One score evaluation, recorded by autograd (the training step's forward; score() is the same under no_grad).  pos (N,3) f32, v (N,) i64, batch (N,) i64 sorted, shape (B,S,3)
    f32, t (B,) i64 -> dict(pred_ligand_pos (N,3), pred_ligand_h (N,H), pred_ligand_v (N,C)).
    bn_eval: the module after .eval() (running batch-norm statistics)."""
    onehot = F.one_hot(v, dm.C).float()
    feat = torch.cat([onehot, _time_embedding(sd, dm, t)[batch]], -1)
    h = _lin(sd, "ligand_atom_emb", feat)
    h, x = _refine(sd, dm, h, pos, batch, shape, taps, bn_eval)
    hv = F.softplus(_lin(sd, "v_inference.0", h)) - math.log(2.0)
    return {"pred_ligand_pos": x, "pred_ligand_h": h, "pred_ligand_v": _lin(sd, "v_inference.2", hv)}


def score(sd, dm, pos, v, batch, shape, t, taps=None, bn_eval=False):
    """score_with_grad without autograd (sampling, validation)."""
    with torch.no_grad():
        return score_with_grad(sd, dm, pos, v, batch, shape, t, taps, bn_eval)


# ------------------------------------------------------------------------------------------
# DDPM posterior step and chain
# ------------------------------------------------------------------------------------------
def _log_add_exp(a, b):
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def _mix_uniform(log_x, log_keep, log_drop, C):
    return _log_add_exp(log_x + log_keep, log_drop - np.log(C))


def gumbel_argmax(logits, u):
    g = -torch.log(-torch.log(u + 1e-30) + 1e-30)
    return (g + logits).argmax(dim=-1)


@torch.no_grad()
def posterior_step(sd, dm, pos, v, pred_pos, pred_v, batch, t, eps, u):
    """One reverse step given the network output.  Returns (pos_next, v_next, log_v0, log_post)."""
    tb = t[batch]
    c0 = sd["posterior_mean_c0_coef"][tb].unsqueeze(-1)
    ct = sd["posterior_mean_ct_coef"][tb].unsqueeze(-1)
    logvar = sd["posterior_logvar"][tb].unsqueeze(-1)
    nonzero = (1 - (t == 0).float())[batch].unsqueeze(-1)
    pos_next = (c0 * pred_pos + ct * pos) + nonzero * (0.5 * logvar).exp() * eps

    log_v0 = F.log_softmax(pred_v, dim=-1)
    log_vt = torch.log(F.one_hot(v, dm.C).float().clamp(min=1e-30))
    tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)[batch]
    a = _mix_uniform(log_v0, sd["log_alphas_cumprod_v"][tm1].unsqueeze(-1),
                     sd["log_one_minus_alphas_cumprod_v"][tm1].unsqueeze(-1), dm.C)
    b = _mix_uniform(log_vt, sd["log_alphas_v"][tb].unsqueeze(-1),
                     sd["log_one_minus_alphas_v"][tb].unsqueeze(-1), dm.C)
    un = a + b
    log_post = un - torch.logsumexp(un, dim=-1, keepdim=True)
    return pos_next, gumbel_argmax(log_post, u), log_v0, log_post


def _v_posterior(sd, dm, log_v0, log_vt, t, batch):
    """q(v_{t-1} | v_t, v_0) in log space (molopt_score_model.py:377-385)."""
    tb = t[batch]
    tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)[batch]
    a = _mix_uniform(log_v0, sd["log_alphas_cumprod_v"][tm1].unsqueeze(-1),
                     sd["log_one_minus_alphas_cumprod_v"][tm1].unsqueeze(-1), dm.C)
    b = _mix_uniform(log_vt, sd["log_alphas_v"][tb].unsqueeze(-1),
                     sd["log_one_minus_alphas_v"][tb].unsqueeze(-1), dm.C)
    un = a + b
    return un - torch.logsumexp(un, dim=-1, keepdim=True)


def _scatter_mean(val, batch, n_mols):
    out = torch.zeros((n_mols,) + val.shape[1:], dtype=val.dtype).index_add_(0, batch, val)
    cnt = torch.zeros(n_mols, dtype=val.dtype).index_add_(0, batch, torch.ones_like(batch, dtype=val.dtype)).clamp(min=1)
    return out / cnt.view(-1, *([1] * (val.dim() - 1)))


def diffusion_loss(sd, dm, pos, v, batch, shape, t, pos_noise, u, bn_eval=True, loss_v_weight=100.0, loss_weight_type="noise_level",
                   with_grad=False):
    """with_grad=False: as validate() evaluates it (no autograd).  with_grad=True: the training step's forward
    (scripts/train_diffusion.py:135-147; bn_eval=False there), so that result['loss'].backward() yields the gradients of every
    tensor of `sd` that requires grad -- the CPU check of the backward kernels (tests/golden/grad_b12.npz pins it)."""
    with torch.set_grad_enabled(bool(with_grad)):
        return _diffusion_loss(sd, dm, pos, v, batch, shape, t, pos_noise, u, bn_eval, loss_v_weight, loss_weight_type)


def _diffusion_loss(sd, dm, pos, v, batch, shape, t, pos_noise, u, bn_eval, loss_v_weight, loss_weight_type):
    """get_diffusion_loss with eval_mode=True and given time steps (molopt_score_model.py:447-531; the form validate() of
    scripts/train_diffusion.py:168-192 calls, module in eval mode): perturb positions and atom types at t (pos_noise (N,3)
    is the normal_() draw of :461, u (N,C) the rand_like of log_sample_categorical inside q_v_sample :366-374), one score
    evaluation, position MSE per molecule weighted by loss_pos_step_weight[t] (:506-518), atom-type KL / decoder NLL
    (compute_v_Lt :436-445).  center_pos_mode = none, v_mode = uniform (the shipped training configuration)."""
    n_mols = shape.shape[0]
    a_pos = sd["alphas_cumprod"][t][batch].unsqueeze(-1)
    pos_pert = a_pos.sqrt() * pos + (1.0 - a_pos).sqrt() * pos_noise
    log_v0 = torch.log(F.one_hot(v, dm.C).float().clamp(min=1e-30))
    tb = t[batch]
    log_qvt = _mix_uniform(log_v0, sd["log_alphas_cumprod_v"][tb].unsqueeze(-1),
                           sd["log_one_minus_alphas_cumprod_v"][tb].unsqueeze(-1), dm.C)
    v_pert = gumbel_argmax(log_qvt, u)
    log_vt = torch.log(F.one_hot(v_pert, dm.C).float().clamp(min=1e-30))
    out = score_with_grad(sd, dm, pos_pert, v_pert, batch, shape, t, bn_eval=bn_eval)      # (autograd follows the caller's mode)
    pred_pos, pred_v = out["pred_ligand_pos"], out["pred_ligand_v"]
    log_recon = F.log_softmax(pred_v, dim=-1)
    log_model = _v_posterior(sd, dm, log_recon, log_vt, t, batch)
    log_true = _v_posterior(sd, dm, log_v0, log_vt, t, batch)
    kl = (log_true.exp() * (log_true - log_model)).sum(dim=1)
    nll = -(log_v0.exp() * log_model).sum(dim=1)
    mask = (t == 0).float()[batch]
    kl_v = _scatter_mean(mask * nll + (1.0 - mask) * kl, batch, n_mols)
    loss_pos = _scatter_mean(((pred_pos - pos) ** 2).sum(-1), batch, n_mols)
    if loss_weight_type == "noise_level":
        loss_pos = torch.mean(sd["loss_pos_step_weight"][t] * loss_pos)
    else:
        loss_pos = torch.mean(loss_pos)
    loss_v = torch.mean(kl_v)
    return {"loss_pos": loss_pos, "loss_v": loss_v, "loss": loss_pos + loss_v * loss_v_weight, "x0": pos,
            "ligand_pos_perturbed": pos_pert, "ligand_v_perturbed": v_pert, "pred_ligand_pos": pred_pos,
            "pred_ligand_v": pred_v, "ligand_v_recon": F.softmax(pred_v, dim=-1)}


@torch.no_grad()
def sample_chain(sd, dm, init_pos, init_v, batch, shape, num_steps, noise_fn, keep_traj=True, guidance=None, bn_eval=False, first_step=0,
                 center=False):
    """Reverse chain t = T-1 ... T-num_steps with host-fed noise ``noise_fn(step) -> (eps, u)``
    (numpy or torch arrays; per step eps (N,3) first, then u (N,C), the reference's draw order).
    first_step = s0 resumes a chain at reverse step s0 (t = T-1-s0) from the given state; noise_fn still counts from 0.
    guidance = (cloud, radius, grad_step, draws (S,5,N)): point-cloud guidance of the predicted x0 while t > grad_step
    (/root/reference/models/molopt_score_model.py:583-586).
    center: center_pos_mode='center' (:52-60, :547, :675-684): the chain runs on coordinates centred per molecule, the offset
    returns onto pos and pos_traj."""
    B = int(batch.max()) + 1
    shape = shape.view(B, -1, 3)
    pos, v = init_pos, init_v
    offset = 0.0
    if center:
        cnt = torch.bincount(batch, minlength=B).to(pos.dtype).clamp(min=1)
        offset = (torch.zeros((B, 3), dtype=pos.dtype).index_add_(0, batch, pos) / cnt[:, None])[batch]
        pos = pos - offset
    out = {k: [] for k in ("pos_traj", "v_traj", "v0_traj", "vt_traj", "pos_cond_traj", "v_cond_traj")}
    for s, i in enumerate(reversed(range(dm.T - first_step - num_steps, dm.T - first_step))):
        t = torch.full((B,), i, dtype=torch.long)
        pr = score(sd, dm, pos, v, batch, shape, t, bn_eval=bn_eval)      # bn_eval: a module put in eval mode before sampling
        if guidance is not None and i > guidance[2]:
            pr["pred_ligand_pos"] = torch.from_numpy(pointcloud_shape_guidance(guidance[0], guidance[1], pr["pred_ligand_pos"].numpy(),
                                                                              guidance[3][s]))
        eps, u = noise_fn(s)
        eps, u = torch.as_tensor(eps), torch.as_tensor(u)
        pos, v, log_v0, log_post = posterior_step(sd, dm, pos, v, pr["pred_ligand_pos"], pr["pred_ligand_v"],
                                                  batch, t, eps, u)
        if keep_traj:
            out["pos_traj"].append(pos + offset); out["v_traj"].append(v.clone())
            out["v0_traj"].append(log_v0); out["vt_traj"].append(log_post)
            out["pos_cond_traj"].append(pr["pred_ligand_pos"]); out["v_cond_traj"].append(pr["pred_ligand_v"])
    out["pos"], out["v"] = pos + offset, v
    return out


def pointcloud_shape_guidance(cloud, radius, pred_pos, draws, k=3, ratio=0.2):
    """Point-cloud shape guidance of a predicted x0 (/root/reference/models/molopt_score_model.py:699-740), restated with a
    brute-force float64 nearest-point search in place of the sklearn KD-tree and with the uniform draws given per
    (iteration, atom) (``draws`` (5, N) float64: the value np.random.random() hands the atom in that iteration).
    pred_pos (N,3) float32 array; returns the guided float32 array."""
    cloud = np.asarray(cloud, np.float64)
    out = np.array(pred_pos, np.float32, copy=True)

    def query(x):
        d2 = ((x[:, None, :].astype(np.float64) - cloud[None, :, :]) ** 2).sum(-1)
        idx = np.argsort(d2, axis=1, kind="stable")[:, :k]
        return np.sqrt(np.take_along_axis(d2, idx, 1)), idx

    dists, idxs = query(out)
    far = np.where(dists.mean(1) > radius)[0]
    pts, pidx = out[far].astype(np.float64), idxs[far]
    for j in range(5):
        if len(far) == 0:
            break
        nearest = cloud[pidx].mean(1)
        scalar = (draws[j, far] * (0.8 - ratio) + ratio)[:, None]
        pts = pts - scalar * (pts - nearest)
        dists, idxs = query(pts)
        inside = dists.mean(1) < radius
        out[far[inside]] = pts[inside].astype(np.float32)
        far, pts, pidx = far[~inside], pts[~inside], idxs[~inside]
    out[far] = pts.astype(np.float32)          # still outside after five pulls: keep the last position (:733-735)
    return out


def state_dict_from_numpy(arrs):
    return {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in arrs.items()}
