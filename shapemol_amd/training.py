"""Training step on the device (the backward pass).

What the reference's ``scripts/train_diffusion.py:135-147`` needs from the model is ``get_diffusion_loss(...)['loss']``
with autograd recording, so that ``loss.backward()`` fills ``.grad`` of every parameter.  The sampling path of this package
is hand-written HIP without a backward; this module is the differentiable evaluation of the same network for the
training step:

  * every MLP block (``models/common.py:47-67``: Linear -> LayerNorm -> ReLU -> Linear; 58 of them per evaluation, ~95 % of
    the FLOPs of forward and backward) is ONE autograd node, :class:`HipMLP`, whose forward and backward are the HIP
    kernels of ``csrc/sm_train.h`` (fp32 MFMA products, deterministic reductions) behind ``shapemol_mlp_forward`` /
    ``shapemol_mlp_backward`` of the C ABI;
  * the four edge functions of every layer (key and value MLPs of x2h and h2x, 32 of the 58, whose input rows are the
    concatenation [r_e | h_i | h_j | s_i]) are :class:`HipEdgeMLP`: the concatenation is never formed, the first Linear is
    an edge term plus per-atom products as on the sampling path (``shapemol_edge_mlp_forward`` / ``_backward``);
  * the attention of every layer (logits, segment softmax over an atom's edges, weighted sum: 16 per evaluation) is ONE
    autograd node, :class:`HipSegAttention`, forward and backward in HIP (``seg_attention_kernel``: the softmax is
    recomputed in the backward, every gradient entry is written by exactly one thread -- no atomics, deterministic);
  * the coordinate update's vector-neuron block (VN-linear, train-mode batch-norm over the atoms, VN-leaky-ReLU, mean over
    the channels: 8 per evaluation) is ONE autograd node, :class:`HipVN` (``vn_*_kernel``; the batch statistics and the
    batch-norm's backward sums are two-pass float64 reductions in a fixed order);
  * what is left in torch device ops recorded by autograd: the kNN graph, the Gaussian smearing of the distances, the
    products with the edge weights and relative positions, the residual sums, the small Linears of the time / atom
    embedding and of the atom-type head (the gate for moving each is ``tests/golden/grad_b12.npz``, the reference's own
    gradients).

Reference semantics followed: ``models/molopt_score_model.py:286-320`` (forward), ``models/uni_transformer.py:48-90,
121-162,181-189,446-540`` (layers, graph, shape embedding), ``models/shape_vn_layers.py:41-61,95-110`` (VN batch-norm in
train mode, running statistics updated with momentum 0.1 as ``nn.BatchNorm1d`` does), ``models/common.py:19-28,39-45``.
There is no CPU path here either: tensors must live on a HIP device.
"""
import ctypes as C
import math

import torch
import torch.nn.functional as F

from . import _lib
from .spec import RBF_CENTRES

VN_EPS = 1e-6


def _p(t):
    return C.c_void_p(t.data_ptr())


class HipMLP(torch.autograd.Function):
    """y = W2 relu(LayerNorm(W1 x + b1)) + b2 on rows of x; forward and backward are HIP kernels (csrc/sm_train.h)."""

    @staticmethod
    def forward(ctx, x, w1, b1, gamma, beta, w2, b2):
        if not x.is_cuda:
            raise RuntimeError("HipMLP needs tensors on a HIP device (shapemol_amd has no CPU path)")
        x = x.contiguous().float()
        ws = [t.detach().contiguous().float() for t in (w1, b1, gamma, beta, w2, b2)]
        rows, k_in = x.shape
        hidden, n_out = ws[0].shape[0], ws[4].shape[0]
        y = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
        xhat = torch.empty((rows, hidden), dtype=torch.float32, device=x.device)
        rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
        act = torch.empty((rows, hidden), dtype=torch.float32, device=x.device)
        if rows > 0:
            with torch.cuda.device(x.device):
                rc = _lib.load().shapemol_mlp_forward(_p(x), rows, k_in, hidden, n_out, _p(ws[0]), _p(ws[1]), _p(ws[2]), _p(ws[3]),
                                                      _p(ws[4]), _p(ws[5]), _p(y), _p(xhat), _p(rstd), _p(act),
                                                      C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            _lib.check(rc, "shapemol_mlp_forward")
        ctx.save_for_backward(x, ws[0], ws[2], ws[3], ws[4], xhat, rstd, act)
        ctx.dims = (rows, k_in, hidden, n_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, gamma, beta, w2, xhat, rstd, act = ctx.saved_tensors
        rows, k_in, hidden, n_out = ctx.dims
        dev = x.device
        dy = dy.contiguous().float()
        z = lambda *s: (torch.empty if rows > 0 else torch.zeros)(s, dtype=torch.float32, device=dev)  # noqa: E731  (the kernels overwrite)
        dx, dw1, db1, dg, dbe, dw2, db2 = z(rows, k_in), z(hidden, k_in), z(hidden), z(hidden), z(hidden), z(n_out, hidden), z(n_out)
        if rows > 0:
            lib = _lib.load()
            n_work = lib.shapemol_mlp_backward_workspace(rows, k_in, hidden, n_out)
            work = torch.empty((n_work,), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.shapemol_mlp_backward(_p(x), _p(dy), rows, k_in, hidden, n_out, _p(w1), _p(gamma), _p(beta), _p(w2), _p(xhat),
                                               _p(rstd), _p(act), _p(dx) if ctx.needs_input_grad[0] else None, _p(dw1), _p(db1), _p(dg), _p(dbe),
                                               _p(dw2), _p(db2), _p(work), n_work, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(rc, "shapemol_mlp_backward")
        return (dx if ctx.needs_input_grad[0] else None), dw1, db1, dg, dbe, dw2, db2


class EdgeGraph:
    """The kNN graph of one evaluation as the kernels want it: edges (src = neighbour j, dst = centre i) grouped by centre with
    CSR offsets ``ptr``, and the same edges grouped by neighbour (``perm_src`` edge indices, ``ptr_src`` offsets) for the
    gradient of terms gathered from the neighbour."""

    def __init__(self, src, dst, ptr):
        n = ptr.numel() - 1
        self.src, self.dst, self.ptr = src.contiguous(), dst.contiguous(), ptr.contiguous()
        self.perm_src = torch.sort(src, stable=True)[1].contiguous()
        self.ptr_src = torch.zeros((n + 1,), dtype=torch.int64, device=src.device)
        self.ptr_src[1:] = torch.cumsum(torch.bincount(src, minlength=n), 0)
        self.n, self.E = n, src.numel()


class HipEdgeMLP(torch.autograd.Function):
    """The MLP block on the rows [r_e | h_i | h_j | s_i] of every edge e = (centre i, neighbour j) without forming them: the
    first Linear is an edge term plus per-atom products (csrc/train_ops.hip, shapemol_edge_mlp_*); forward and backward in HIP."""

    @staticmethod
    def forward(ctx, r, h, s, graph, w1, b1, gamma, beta, w2, b2):
        if not r.is_cuda:
            raise RuntimeError("HipEdgeMLP needs tensors on a HIP device (shapemol_amd has no CPU path)")
        r, h, s = r.contiguous().float(), h.contiguous().float(), s.contiguous().float()
        ws = [t.detach().contiguous().float() for t in (w1, b1, gamma, beta, w2, b2)]
        E, n = graph.E, graph.n
        kr, kn, ks, hidden, n_out = r.shape[1], h.shape[1], s.shape[1], ws[0].shape[0], ws[4].shape[0]
        if r.shape[0] != E or h.shape[0] != n or s.shape[0] != n or ws[0].shape[1] != kr + 2 * kn + ks:
            raise ValueError("HipEdgeMLP: r is one row per edge, h and s one row per atom, W1 has k_edge + 2 k_node + k_shape columns")
        new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=r.device)  # noqa: E731
        y, xhat, rstd, act, pd, ps = new(E, n_out), new(E, hidden), new(E), new(E, hidden), new(n, hidden), new(n, hidden)
        if E > 0:
            with torch.cuda.device(r.device):
                rc = _lib.load().shapemol_edge_mlp_forward(_p(r), _p(h), _p(s), _p(graph.dst), _p(graph.src), E, n, kr, kn, ks, hidden, n_out,
                                                           *[_p(t) for t in ws], _p(y), _p(xhat), _p(rstd), _p(act), _p(pd), _p(ps),
                                                           C.c_void_p(torch.cuda.current_stream(r.device).cuda_stream))
            _lib.check(rc, "shapemol_edge_mlp_forward")
        ctx.save_for_backward(r, h, s, ws[0], ws[2], ws[3], ws[4], xhat, rstd, act)
        ctx.graph, ctx.dims = graph, (E, n, kr, kn, ks, hidden, n_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        r, h, s, w1, gamma, beta, w2, xhat, rstd, act = ctx.saved_tensors
        E, n, kr, kn, ks, hidden, n_out = ctx.dims
        g, dev = ctx.graph, r.device
        dy = dy.contiguous().float()
        z = lambda *sh: (torch.empty if E > 0 else torch.zeros)(sh, dtype=torch.float32, device=dev)  # noqa: E731  (the kernels overwrite)
        dr, dh, ds = z(E, kr), z(n, kn), z(n, ks)
        dw1, db1, dg, dbe, dw2, db2 = z(hidden, kr + 2 * kn + ks), z(hidden), z(hidden), z(hidden), z(n_out, hidden), z(n_out)
        if E > 0:
            lib = _lib.load()
            n_work = lib.shapemol_edge_mlp_backward_workspace(E, n, kr, kn, ks, hidden, n_out)
            work = torch.empty((n_work,), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.shapemol_edge_mlp_backward(_p(r), _p(h), _p(s), _p(g.ptr), _p(g.perm_src), _p(g.ptr_src), _p(dy), E, n, kr, kn, ks, hidden, n_out,
                                                    _p(w1), _p(gamma), _p(beta), _p(w2), _p(xhat), _p(rstd), _p(act), _p(dr), _p(dh), _p(ds), _p(dw1), _p(db1),
                                                    _p(dg), _p(dbe), _p(dw2), _p(db2), _p(work), n_work, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(rc, "shapemol_edge_mlp_backward")
        return dr, dh, ds, None, dw1, db1, dg, dbe, dw2, db2


def _edge_mlp(P, prefix, r, h, s, graph):
    return HipEdgeMLP.apply(r, h, s, graph, P[prefix + ".net.0.weight"], P[prefix + ".net.0.bias"], P[prefix + ".net.1.weight"],
                            P[prefix + ".net.1.bias"], P[prefix + ".net.3.weight"], P[prefix + ".net.3.bias"])


class HipSegAttention(torch.autograd.Function):
    """out_i[h] = sum_e softmax_e(<q_i[h], k_e[h]> / sqrt(dh)) vals_e[h] over the incoming edges e of atom i (edges grouped by
    centre atom, ``ptr`` their CSR offsets); forward and backward are HIP kernels (csrc/sm_train.h, seg_attention_kernel)."""

    @staticmethod
    def forward(ctx, q, k, vals, ptr, heads):
        if not q.is_cuda:
            raise RuntimeError("HipSegAttention needs tensors on a HIP device (shapemol_amd has no CPU path)")
        q, k, vals = q.contiguous().float(), k.contiguous().float(), vals.contiguous().float()
        n, dh, width = q.shape[0], q.shape[1] // heads, vals.shape[2]
        if ptr.dtype != torch.int64 or ptr.numel() != n + 1 or k.shape[0] != vals.shape[0] or vals.shape[1] != heads:
            raise ValueError("HipSegAttention: ptr must be int64 of n_atoms + 1 entries, k and vals one row per edge")
        out = (torch.empty if k.shape[0] > 0 else torch.zeros)((n, heads, width), dtype=torch.float32, device=q.device)
        if n > 0 and k.shape[0] > 0:
            with torch.cuda.device(q.device):
                rc = _lib.load().shapemol_seg_attention_forward(_p(q), _p(k), _p(vals), _p(ptr), n, heads, dh, width, _p(out),
                                                                C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
            _lib.check(rc, "shapemol_seg_attention_forward")
        ctx.save_for_backward(q, k, vals, ptr)
        ctx.dims = (n, heads, dh, width)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, vals, ptr = ctx.saved_tensors
        n, heads, dh, width = ctx.dims
        dout = dout.contiguous().float()
        dq, dk, dvals = (torch.empty_like if k.shape[0] > 0 else torch.zeros_like)(q), torch.empty_like(k), torch.empty_like(vals)  # every entry is written
        if n > 0 and k.shape[0] > 0:
            with torch.cuda.device(q.device):
                rc = _lib.load().shapemol_seg_attention_backward(_p(q), _p(k), _p(vals), _p(ptr), _p(dout), n, heads, dh, width,
                                                                 _p(dq), _p(dk), _p(dvals),
                                                                 C.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
            _lib.check(rc, "shapemol_seg_attention_backward")
        return dq, dk, dvals, None, None


class HipVN(torch.autograd.Function):
    """mean over the output channels of VNLinearLeakyReLU (with VNBatchNorm) applied to [x_n | o3_n | shape_mol(n)]
    (models/shape_vn_layers.py:41-61,95-110; uni_transformer.py:157-160): forward and backward in HIP (csrc/sm_train.h, vn_*)."""

    @staticmethod
    def forward(ctx, x, o3, shape, batch, wf, wd, bn_w, bn_b, run_mean, run_var, training):
        if not x.is_cuda:
            raise RuntimeError("HipVN needs tensors on a HIP device (shapemol_amd has no CPU path)")
        if shape.requires_grad:
            raise NotImplementedError("HipVN: no gradient for the shape embedding (it comes from the frozen encoder)")
        x, o3, shape = x.contiguous().float(), o3.contiguous().float(), shape.contiguous().float()
        ws = [t.detach().contiguous().float() for t in (wf, wd, bn_w, bn_b)]
        n, rows_o, rows_s, ch = x.shape[0], o3.shape[1], shape.shape[1], ws[0].shape[0]
        if ws[0].shape[1] != 1 + rows_o + rows_s or run_mean.dtype != torch.float32 or not run_mean.is_contiguous() or not run_var.is_contiguous():
            raise ValueError("HipVN: the VN weights have 1 + rows_o + rows_s columns; running statistics are contiguous float32")
        new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=x.device)  # noqa: E731
        out, pf, dr, stats, nrm = new(n, 3), new(n, ch, 3), new(n, ch, 3), new(2, ch), new(n, ch)
        with torch.cuda.device(x.device):
            rc = _lib.load().shapemol_vn_forward(_p(x), _p(o3), _p(shape), _p(batch), n, rows_o, rows_s, ch, *[_p(t) for t in ws], _p(run_mean), _p(run_var),
                                                 int(bool(training)), _p(out), _p(pf), _p(dr), _p(stats), _p(nrm),
                                                 C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        _lib.check(rc, "shapemol_vn_forward")
        if training:      # the kernel wrote the running estimates through raw pointers: bump their version counters, so that whoever
            run_mean.add_(0); run_var.add_(0)      # keys a cache on (data_ptr, _version) -- the validation context of the model -- sees the update
        ctx.save_for_backward(x, o3, shape, batch, *ws, pf, dr, stats)
        ctx.dims = (n, rows_o, rows_s, ch, int(bool(training)))
        return out

    @staticmethod
    def backward(ctx, gout):
        x, o3, shape, batch, wf, wd, bn_w, bn_b, pf, dr, stats = ctx.saved_tensors
        n, rows_o, rows_s, ch, training = ctx.dims
        dev = x.device
        gout = gout.contiguous().float()
        new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)  # noqa: E731
        dx, do3, dwf, dwd, dg, db = new(n, 3), new(n, rows_o, 3), new(ch, 1 + rows_o + rows_s), new(ch, 1 + rows_o + rows_s), new(ch), new(ch)
        lib = _lib.load()
        n_work = lib.shapemol_vn_backward_workspace(n, rows_o, rows_s, ch)
        work = new(n_work)
        with torch.cuda.device(dev):
            rc = lib.shapemol_vn_backward(_p(x), _p(o3), _p(shape), _p(batch), n, rows_o, rows_s, ch, _p(wf), _p(wd), _p(bn_w), _p(bn_b), _p(pf), _p(dr),
                                          _p(stats), training, _p(gout), _p(dx), _p(do3), _p(dwf), _p(dwd), _p(dg), _p(db), _p(work), n_work,
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "shapemol_vn_backward")
        return dx, do3, None, None, dwf, dwd, dg, db, None, None, None


def _mlp(P, prefix, x):
    return HipMLP.apply(x, P[prefix + ".net.0.weight"], P[prefix + ".net.0.bias"], P[prefix + ".net.1.weight"], P[prefix + ".net.1.bias"],
                        P[prefix + ".net.3.weight"], P[prefix + ".net.3.bias"])


def _rbf(d):
    """Gaussian smearing with the reference's 20 fixed centres, coeff = -0.5 (models/common.py:19-28)."""
    mu = torch.tensor(RBF_CENTRES, dtype=torch.float32, device=d.device)
    return torch.exp(-0.5 * (d.view(-1, 1) - mu.view(1, -1)) ** 2)


def knn_edges(x, batch, k):
    """Per-molecule k nearest neighbours on the device (self excluded; ties by (squared distance, index), the squared
    distance evaluated as (dx*dx + dy*dy) + dz*dz like the sampling kernels): (src = j, dst = i), grouped by centre i, and the CSR offsets of the groups (``batch`` sorted, as the reference's collate emits it)."""
    n = x.shape[0]
    if n == 0:
        raise ValueError("knn_edges: empty batch")
    if bool((batch[1:] < batch[:-1]).any()):
        raise ValueError("knn_edges: the batch vector must be sorted (atoms of a molecule contiguous), as the sampling path requires")
    counts = torch.bincount(batch)
    B, M = counts.shape[0], int(counts.max())
    if M >= 65536:
        raise ValueError("knn_edges: molecules of 65536 atoms or more are not supported (the tie-breaking index takes 16 bits of the key)")
    first = torch.cumsum(counts, 0) - counts
    local = torch.arange(n, device=x.device) - first[batch]
    pad = torch.zeros((B, M, 3), dtype=x.dtype, device=x.device)
    pad[batch, local] = x.detach()
    valid = torch.zeros((B, M), dtype=torch.bool, device=x.device)
    valid[batch, local] = True
    d = pad[:, :, None, :] - pad[:, None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    inf = torch.tensor(float("inf"), device=x.device)
    d2 = torch.where(valid[:, None, :] & valid[:, :, None], d2, inf)
    d2 = torch.where(torch.eye(M, dtype=torch.bool, device=x.device)[None], inf, d2)
    # (B, M, k) local neighbour indices, ascending by (squared distance, index): topk on a key that breaks exact distance ties by
    # the index (float32 distances are non-negative: their bit patterns order like integers; one kernel instead of the ~70
    # merge passes of a stable sort of every row)
    key = (d2.view(torch.int32).to(torch.int64) << 16) | torch.arange(M, device=x.device, dtype=torch.int64)[None, None, :]
    order = torch.topk(key, min(k, M), dim=2, largest=False, sorted=True)[1]
    rank = torch.arange(order.shape[2], device=x.device)
    has = valid[:, :, None] & (rank[None, None, :] < (counts - 1).clamp(max=k)[:, None, None])
    src = (order + first[:, None, None])[has]
    dst = (torch.arange(M, device=x.device)[None, :, None] + first[:, None, None]).expand_as(order)[has]
    ptr = torch.zeros((n + 1,), dtype=torch.int64, device=x.device)
    ptr[1:] = torch.cumsum(has.sum(2)[valid], 0)          # atoms in batch order == (molecule, local index) order
    return src, dst, ptr


def _vn_update(P, B, p, x, o3, shape, batch, training):
    """mean_c VNLinearLeakyReLU([x | o3 | shape[batch]]) -> (N, 3); train mode also updates the running statistics."""
    out = HipVN.apply(x, o3, shape, batch, P[p + ".map_to_feat.weight"], P[p + ".map_to_dir.weight"], P[p + ".batchnorm.bn.weight"],
                      P[p + ".batchnorm.bn.bias"], B[p + ".batchnorm.bn.running_mean"], B[p + ".batchnorm.bn.running_var"], training)
    if training:
        B[p + ".batchnorm.bn.num_batches_tracked"].add_(1)
    return out


def _named_tensors(model):
    """name -> Parameter and name -> buffer maps of the model.  ``dict(model.named_parameters())`` walks the module tree (the
    model is a tree of ~400 small holders: 2.8 ms per call, a tenth of a training step); the (name, holder module, leaf)
    triples are kept on the model instead and the tensors looked up in their holders on every call, so that parameters or
    buffers replaced by ``.to()`` / ``load_state_dict(assign=True)`` are seen."""
    slots = model.__dict__.get("_tensor_slots")
    if slots is None:
        slots = ([], [])
        for prefix, mod in model.named_modules():
            dot = prefix + "." if prefix else ""
            slots[0].extend((dot + leaf, mod, leaf) for leaf in mod._parameters)
            slots[1].extend((dot + leaf, mod, leaf) for leaf in mod._buffers)
        model.__dict__["_tensor_slots"] = slots
    return ({name: mod._parameters[leaf] for name, mod, leaf in slots[0] if mod._parameters[leaf] is not None},
            {name: mod._buffers[leaf] for name, mod, leaf in slots[1] if mod._buffers[leaf] is not None})


def score_with_grad(model, pos, v, batch, shape, t):
    """One score evaluation recorded by autograd; same result dict as ``ScorePosNet3D.forward``."""
    dm = model.dims
    P, Bf = _named_tensors(model)
    lin = lambda p, x: F.linear(x, P[p + ".weight"], P[p + ".bias"])  # noqa: E731
    n = pos.shape[0]
    # time embedding (molopt_score_model.py:154-166,247-252) and atom embedding (:292-301)
    half = dm.temb // 2
    freq = torch.exp(torch.arange(half, device=pos.device, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    arg = t[:, None].float() * freq[None, :]
    temb = lin("time_emb.3", F.silu(lin("time_emb.1", torch.cat((arg.sin(), arg.cos()), dim=-1))))
    h = lin("ligand_atom_emb", torch.cat([F.one_hot(v, dm.C).float(), temb[batch]], -1))
    # invariant shape embedding (uni_transformer.py:181-189), graph, edge weights (:446-481)
    shape = shape.view(-1, dm.S, 3)
    m = shape.mean(dim=1)
    m = m / ((m * m).sum(-1, keepdim=True) + VN_EPS)
    inv_atom = _mlp(P, "refine_net.invariant_shape_layer.hidden_layer", torch.einsum("bij,bj->bi", shape, m))[batch]
    x = pos
    src, dst, ptr = knn_edges(x, batch, dm.k)
    graph = EdgeGraph(src, dst, ptr)
    e_w = torch.sigmoid(_mlp(P, "refine_net.edge_pred_layer", _rbf(torch.norm(x[dst] - x[src], p=2, dim=-1))))
    dh = dm.H // dm.heads
    for l in range(dm.L):
        p = f"refine_net.base_block.{l}"
        rel_x = x[dst] - x[src]
        rfeat = _rbf(torch.norm(rel_x, p=2, dim=-1))
        # x2h (uni_transformer.py:48-90)
        px = p + ".x2h_layers.0"
        val = (_edge_mlp(P, px + ".hv_func", rfeat, h, inv_atom, graph) * e_w.view(-1, 1)).view(-1, dm.heads, dh)
        o = HipSegAttention.apply(_mlp(P, px + ".hq_func", h), _edge_mlp(P, px + ".hk_func", rfeat, h, inv_atom, graph), val, ptr, dm.heads).view(n, dm.H)
        h = _mlp(P, px + ".node_output", torch.cat([o, h], -1)) + h
        # h2x (uni_transformer.py:121-162)
        ph = p + ".h2x_layers.0"
        val = (_edge_mlp(P, ph + ".xv_func", rfeat, h, inv_atom, graph) * e_w.view(-1, 1)).unsqueeze(-1) * rel_x.unsqueeze(1)
        o3 = HipSegAttention.apply(_mlp(P, ph + ".xq_func", h), _edge_mlp(P, ph + ".xk_func", rfeat, h, inv_atom, graph), val, ptr, dm.heads)   # (N, heads, 3)
        x = x + o3.mean(dim=1) + _vn_update(P, Bf, ph + ".shape_linear", x, o3, shape, batch, model.training)
    hv = F.softplus(lin("v_inference.0", h)) - math.log(2.0)
    return {"pred_ligand_pos": x, "pred_ligand_h": h, "pred_ligand_v": lin("v_inference.2", hv)}
