"""Drop-in host module for the reference's ``models/molopt_score_model.py`` hot path.

Same call surface as the reference (paths relative to the reference repository):
  * ``ScorePosNet3D(config, ligand_atom_feature_dim)``        models/molopt_score_model.py:171
    -- same state-dict keys (446 for the shipped config), so ``load_state_dict(ckpt['model'])``
       works unchanged (scripts/sample_diffusion.py:211-215)
  * ``.forward(ligand_pos_perturbed, ligand_v_perturbed, batch_ligand, ligand_shape, time_step, return_all)``
                                                              models/molopt_score_model.py:286-320
  * ``.sample_diffusion(init_ligand_pos, init_ligand_v, batch_ligand, ligand_shape, ...)``
                                                              models/molopt_score_model.py:533-697
  * ``log_sample_categorical(logits)``                        models/molopt_score_model.py:98-104

All arithmetic runs in libshapemol_hip.so (hand-written HIP for gfx950) through the C ABI of
``include/shapemol_hip.h``; torch only owns device memory and streams here.  The module's
``training`` flag selects the VN batch-norm's statistics as ``nn.BatchNorm1d`` does: train mode
(what the reference's sampling script runs in) normalises with the statistics of
the current batch, ``eval()`` (what ``validate()`` switches to) with the running statistics of
the state dict; the running statistics are read, never updated.
There is no CPU path: tensors must live on a HIP device and the library must be built.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .diffusion import build_schedule_tables
from .packing import pack_state_dict
from .spec import ModelDims, state_dict_spec, RBF_CENTRES

__all__ = ["ScorePosNet3D", "log_sample_categorical", "pointcloud_shape_guidance"]


class _Params(nn.Module):
    """A bare container; sub-modules and parameters are attached by name to reproduce the
    reference's state-dict keys without reproducing its classes."""


def _attach(root, key, tensor, kind):
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Params())
        mod = mod._modules[p]
    leaf = parts[-1]
    if kind in ("running_mean", "running_var", "counter") or leaf == "offset":
        mod.register_buffer(leaf, tensor)
    else:
        mod.register_parameter(leaf, nn.Parameter(tensor, requires_grad=(kind != "const")))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream_ptr(stream):
    return C.c_void_p(stream.cuda_stream)


def _check_device_tensor(name, t, dtype):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on a HIP device (shapemol_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t.contiguous()


# library defaults of the options the precision modes touch (shapemol_set_option; include/shapemol_hip.h)
DEFAULT_OPTIONS = {"edge_bf16": 2, "node_f16": 0, "feat_f16": 0}


class ScorePosNet3D(nn.Module):
    _accelerated = True      # shapemol_amd.sampling: this sample_diffusion takes the private host-buffer keyword

    def __init__(self, config, ligand_atom_feature_dim):
        super().__init__()
        self.config = config
        self.dims = ModelDims(config, ligand_atom_feature_dim)
        g = (config.get if hasattr(config, "get") else lambda k, d=None: getattr(config, k, d))
        self.denoise_type = g("denoise_type")
        self.model_mean_type = g("model_mean_type")
        self.loss_v_weight = g("loss_v_weight")
        self.loss_weight_type = g("loss_weight_type")
        self.v_mode = g("v_mode")
        self.v_net_type = g("v_net_type", "mlp") or "mlp"
        self.sample_time_method = g("sample_time_method")
        self.loss_pos_type = g("loss_pos_type")
        self.hidden_dim = self.dims.H
        self.num_classes = self.dims.C
        self.center_pos_mode = g("center_pos_mode")
        self.time_emb_dim = self.dims.temb
        self.refine_net_type = g("model_type")
        self.cond_mask_prob = g("cond_mask_prob")

        tables = build_schedule_tables(config)
        gen = torch.Generator().manual_seed(0)
        for key, (shape, kind, fan_in) in state_dict_spec(self.dims).items():
            if kind == "const":
                arr = np.asarray(RBF_CENTRES, np.float32) if key.endswith("offset") else tables[key]
                t = torch.from_numpy(np.array(arr, dtype=np.float32))
            elif kind == "counter":
                t = torch.zeros(shape, dtype=torch.long)
            elif kind in ("running_mean", "norm_bias"):
                t = torch.zeros(shape)
            elif kind in ("running_var", "norm_weight"):
                t = torch.ones(shape)
            else:   # Linear weight / bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in)), torch.nn.Linear's default range
                bound = 1.0 / np.sqrt(fan_in)
                t = (torch.rand(shape, generator=gen) * 2 - 1) * bound
            _attach(self, key, t, kind)
        self.num_timesteps = self.dims.T
        self._ctx = None
        self._ctx_key = None

    # ------------------------------------------------------------------ library context
    def _weights_key(self, device):
        return (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters()) \
            + tuple((b.data_ptr(), b._version) for n, b in self.named_buffers() if n.endswith(("running_mean", "running_var")))

    def _bn_running(self):
        """(mean, var) float32 [L][heads]: the batch-norm running statistics of the coordinate updates, layer by layer."""
        sd = self.state_dict()
        key = "refine_net.base_block.{}.h2x_layers.0.shape_linear.batchnorm.bn.running_{}"
        take = lambda which: np.ascontiguousarray(np.stack([sd[key.format(l, which)].detach().cpu().numpy() for l in range(self.dims.L)]), np.float32)
        return take("mean"), take("var")

    def _context(self, device, slot=0):
        """Create (or refresh after a weight change / device move) the library context.  slot > 0: additional contexts over the
        same weights (own workspace and captured graphs each), so that independent chains can be in flight side by side
        (shapemol_amd.sampling keeps two)."""
        key = self._weights_key(device)
        if slot != 0:
            extra = self.__dict__.setdefault("_ctx_extra", {})
            ent = extra.get(slot)
            if ent is not None and ent["key"] == key:
                return ent["ctx"]
            if ent is not None:
                _lib.load().shapemol_destroy(ent["ctx"])
            extra[slot] = {"ctx": self._new_context(device), "key": key, "bn_eval": False}
            return extra[slot]["ctx"]
        if self._ctx is not None and key == self._ctx_key:
            return self._ctx
        self._release(extra=False)
        self._ctx, self._ctx_key, self._bn_eval_set = self._new_context(device), key, False
        return self._ctx

    def _new_context(self, device):
        lib = _lib.load()
        d = self.dims
        cfg = _lib.Config(d.H, d.heads, d.L, d.k, d.G, d.S, d.S_latent, d.temb, d.C, d.T)
        packed = pack_state_dict(self.state_dict(), d.L)
        want = lib.shapemol_weight_count(C.byref(cfg))
        if packed.size != want:
            raise _lib.ShapeMolLibraryError(f"packed weight count {packed.size} != library's {want}")
        ctx = C.c_void_p()
        index = device.index if device.index is not None else torch.cuda.current_device()
        _lib.check(lib.shapemol_create(C.byref(cfg), packed.ctypes.data_as(C.c_void_p), packed.size, index, C.byref(ctx)),
                   "shapemol_create")
        mean, var = self._bn_running()
        _lib.check(lib.shapemol_set_bn_running(ctx, mean.ctypes.data_as(C.c_void_p), var.ctypes.data_as(C.c_void_p), mean.size),
                   "shapemol_set_bn_running")
        # options set through set_option apply to every context; replayed in a dependency-safe order (feat_f16 needs the f16
        # kernels selected first), and a context that cannot take them is destroyed, not leaked
        opts = self.__dict__.get("_options", {})
        order = sorted(opts, key=lambda k: {"edge_bf16": 0, "node_f16": 1, "feat_f16": 3}.get(k, 2))
        try:
            for name in order:
                _lib.check(lib.shapemol_set_option(ctx, name.encode(), int(opts[name])), "shapemol_set_option")
        except Exception:
            lib.shapemol_destroy(ctx)
            raise
        return ctx

    def _sync_bn_mode(self, ctx, slot=0):
        """module.eval() / .train() -> the library's batch-norm mode (running statistics / the batch's), as nn.BatchNorm1d
        switches with the module's flag (models/shape_vn_layers.py:45-61).  Sampling in the reference runs in train mode
        (the scripts never call .eval() before sample_diffusion); validate() switches to eval."""
        want = not self.training
        if slot != 0:
            ent = self.__dict__["_ctx_extra"][slot]
            if want != ent["bn_eval"]:
                _lib.check(_lib.load().shapemol_set_option(ctx, b"bn_eval", int(want)), "shapemol_set_option")
                ent["bn_eval"] = want
            return
        if want != getattr(self, "_bn_eval_set", False):
            _lib.check(_lib.load().shapemol_set_option(ctx, b"bn_eval", int(want)), "shapemol_set_option")
            self._bn_eval_set = want

    def _release(self, extra=True):
        if getattr(self, "_ctx", None) is not None:
            _lib.load().shapemol_destroy(self._ctx)
            self._ctx = None
        if extra:
            for ent in self.__dict__.get("_ctx_extra", {}).values():
                _lib.load().shapemol_destroy(ent["ctx"])
            self.__dict__["_ctx_extra"] = {}

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def set_option(self, name, value):
        """Library tuning / diagnostics knob (see shapemol_set_option)."""
        dev = next(self.parameters()).device
        _lib.check(_lib.load().shapemol_set_option(self._context(dev), name.encode(), int(value)), "shapemol_set_option")
        if name == "bn_eval":           # keep the cache of _sync_bn_mode truthful
            self._bn_eval_set = bool(value)
        else:                           # (the batch-norm mode follows module.training; everything else is remembered for contexts
            self.__dict__.setdefault("_options", {})[name] = int(value)      #  created later: weight reloads, extra slots)
        for slot, ent in self.__dict__.get("_ctx_extra", {}).items():
            _lib.check(_lib.load().shapemol_set_option(ent["ctx"], name.encode(), int(value)), "shapemol_set_option")
            if name == "bn_eval":
                ent["bn_eval"] = bool(value)

    def set_knn_pins(self, step=None, atom=None, nbr=None, num_steps=None):
        """Diagnostic of the parity tests (shapemol_set_knn_pins): pin the kNN graph of the following sample_diffusion calls at
        the given (reverse step, atom) pairs to the given neighbour lists; no arguments remove the pins."""
        dev = next(self.parameters()).device
        ctx = self._context(dev)
        lib = _lib.load()
        if step is None or len(step) == 0:
            _lib.check(lib.shapemol_set_knn_pins(ctx, None, 0, None, None, 0, 0), "shapemol_set_knn_pins")
            return
        step = np.asarray(step, np.int64)
        order = np.argsort(step, kind="stable")
        step, atom, nbr = step[order], np.ascontiguousarray(np.asarray(atom, np.int32)[order]), np.ascontiguousarray(np.asarray(nbr, np.int32)[order])
        n_steps = int(num_steps if num_steps is not None else self.num_timesteps)
        off = np.zeros(n_steps + 1, np.int32)
        np.add.at(off, step + 1, 1)
        off = np.ascontiguousarray(np.cumsum(off).astype(np.int32))
        _lib.check(lib.shapemol_set_knn_pins(ctx, off.ctypes.data_as(C.c_void_p), n_steps, atom.ctypes.data_as(C.c_void_p),
                                             nbr.ctypes.data_as(C.c_void_p), len(atom), nbr.shape[1]), "shapemol_set_knn_pins")

    def debug_read(self, name, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        n = _lib.load().shapemol_debug_read(self._ctx, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes)
        if n != out.nbytes:
            raise _lib.ShapeMolLibraryError(f"debug_read({name}): got {n} bytes, want {out.nbytes}")
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, ligand_pos_perturbed, ligand_v_perturbed, batch_ligand, ligand_shape, time_step=None,
                return_all=False):
        """f(x0, v0 | xt, vt): one score evaluation.  Returns the reference's dict
        {'pred_ligand_pos' (N,3), 'pred_ligand_h' (N,H), 'pred_ligand_v' (N,C)}."""
        if time_step is None:
            raise ValueError("time_step is required (time_emb_dim > 0)")
        pos = _check_device_tensor("ligand_pos_perturbed", ligand_pos_perturbed, torch.float32)
        v = _check_device_tensor("ligand_v_perturbed", ligand_v_perturbed, torch.int64)
        batch = _check_device_tensor("batch_ligand", batch_ligand, torch.int64)
        shape = _check_device_tensor("ligand_shape", ligand_shape, torch.float32).view(-1, self.dims.S, 3)
        t = _check_device_tensor("time_step", time_step, torch.int64)
        n, b = pos.shape[0], shape.shape[0]
        if v.shape[0] != n or batch.shape[0] != n or t.shape[0] != b or pos.shape[1] != 3:
            raise ValueError("inconsistent input shapes")
        dev = pos.device
        ctx = self._context(dev)
        self._sync_bn_mode(ctx)
        out_pos = torch.empty((n, 3), dtype=torch.float32, device=dev)
        out_h = torch.empty((n, self.dims.H), dtype=torch.float32, device=dev)
        out_v = torch.empty((n, self.dims.C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().shapemol_score(ctx, _ptr(pos), _ptr(v), _ptr(batch), n, b, _ptr(shape), _ptr(t),
                                            _ptr(out_pos), _ptr(out_h), _ptr(out_v),
                                            _stream_ptr(torch.cuda.current_stream(dev)))
            _lib.check(rc, "shapemol_score")
            preds = {"pred_ligand_pos": out_pos, "pred_ligand_h": out_h, "pred_ligand_v": out_v}
            if return_all:
                # the refine net's all_x / all_h hold its input and the output of each BLOCK (models/uni_transformer.py:489-533;
                # the shipped schema has num_blocks = 1): [input, final].  The atom-type head on the embedding h0 is one more
                # evaluation truncated to zero layers (embedding + v-head only).
                out_v0 = torch.empty_like(out_v)
                lib = _lib.load()
                _lib.check(lib.shapemol_set_option(ctx, b"stop_layer", 0), "shapemol_set_option")
                try:
                    rc = lib.shapemol_score(ctx, _ptr(pos), _ptr(v), _ptr(batch), n, b, _ptr(shape), _ptr(t), _ptr(torch.empty_like(out_pos)),
                                            None, _ptr(out_v0), _stream_ptr(torch.cuda.current_stream(dev)))
                    _lib.check(rc, "shapemol_score")
                finally:
                    lib.shapemol_set_option(ctx, b"stop_layer", -1)
                preds.update(layer_pred_ligand_pos=[pos, out_pos], layer_pred_ligand_v=[out_v0, out_v])
        return preds

    # ------------------------------------------------------------------ validation loss
    def sample_time(self, num_graphs, device):
        """Symmetric time sampling (models/molopt_score_model.py:415-422)."""
        time_step = torch.randint(0, self.num_timesteps, size=(num_graphs // 2 + 1,), device=device)
        time_step = torch.cat([time_step, self.num_timesteps - time_step - 1], dim=0)[:num_graphs]
        return time_step, torch.ones_like(time_step).float() / self.num_timesteps

    def _table(self, name):
        return getattr(self, name).detach()        # schedule vectors are registered on the module itself (top-level state-dict keys)

    def _v_mix(self, log_x, log_keep, log_drop):
        a, b = log_x + log_keep, log_drop - float(np.log(self.num_classes))
        m = torch.max(a, b)
        return m + torch.log(torch.exp(a - m) + torch.exp(b - m))

    def _q_v_posterior(self, log_v0, log_vt, t, batch):
        tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)[batch]
        tb = t[batch]
        un = self._v_mix(log_v0, self._table("log_alphas_cumprod_v")[tm1].unsqueeze(-1), self._table("log_one_minus_alphas_cumprod_v")[tm1].unsqueeze(-1)) \
            + self._v_mix(log_vt, self._table("log_alphas_v")[tb].unsqueeze(-1), self._table("log_one_minus_alphas_v")[tb].unsqueeze(-1))
        return un - torch.logsumexp(un, dim=-1, keepdim=True)

    @staticmethod
    def _scatter_mean(val, batch, n_mols):
        out = torch.zeros((n_mols,) + tuple(val.shape[1:]), dtype=val.dtype, device=val.device).index_add_(0, batch, val)
        cnt = torch.bincount(batch, minlength=n_mols).clamp(min=1).to(val.dtype)
        return out / cnt.view(-1, *([1] * (val.dim() - 1)))

    def get_diffusion_loss(self, ligand_pos, ligand_v, batch_ligand, ligand_shape=None, time_step=None, eval_mode=False, *,
                           noise=None):
        """The reference's loss evaluation (models/molopt_score_model.py:447-531): perturb positions and atom types at
        ``time_step`` (sampled symmetrically if None), one score evaluation on the device, position MSE and atom-type KL
        per molecule.  Same arguments and result dict.  Under torch.no_grad() -- the form validate() runs
        (scripts/train_diffusion.py:168-192, module in eval mode, which switches the batch-norm to its running statistics) --
        the score evaluation is the HIP sampling path.  With autograd enabled -- the training step,
        scripts/train_diffusion.py:135-147 -- it is the differentiable evaluation of shapemol_amd.training (every operator of a
        layer forward and backward in HIP; graph construction, distance features and embeddings in torch device ops), so that
        ``result['loss'].backward()`` fills ``.grad`` of every parameter (gate: tests/golden/grad_b12.npz, the reference's own
        gradients).

        Extension (keyword-only): ``noise=(pos_noise (N,3), u (N,C))`` feeds the normal draw of :461 and the uniforms of
        log_sample_categorical (:98-104, inside q_v_sample :366-374) instead of torch's generator (parity tests)."""
        if self.v_mode != "uniform":
            raise NotImplementedError("v_mode = 'uniform' only (the shipped training configuration)")
        pos = _check_device_tensor("ligand_pos", ligand_pos, torch.float32)
        v = _check_device_tensor("ligand_v", ligand_v, torch.int64)
        batch = _check_device_tensor("batch_ligand", batch_ligand, torch.int64)
        shape = _check_device_tensor("ligand_shape", ligand_shape, torch.float32)
        if self.center_pos_mode == "center":
            n_all = int(shape.view(-1, self.dims.S, 3).shape[0])
            pos = pos - self._scatter_mean(pos, batch, n_all)[batch]
        elif self.center_pos_mode not in (None, "none"):
            raise NotImplementedError(self.center_pos_mode)
        shape = shape.view(-1, self.dims.S, 3)
        num_graphs = shape.shape[0]
        if time_step is None:
            time_step, _ = self.sample_time(num_graphs, pos.device)
        t = _check_device_tensor("time_step", time_step, torch.int64)
        # perturb positions and atom types
        a_pos = self._table("alphas_cumprod")[t][batch].unsqueeze(-1)
        pos_noise = _check_device_tensor("noise[0]", noise[0], torch.float32) if noise is not None else torch.zeros_like(pos).normal_()
        pos_pert = a_pos.sqrt() * pos + (1.0 - a_pos).sqrt() * pos_noise
        log_v0 = torch.log(torch.nn.functional.one_hot(v, self.num_classes).float().clamp(min=1e-30))
        tb = t[batch]
        log_qvt = self._v_mix(log_v0, self._table("log_alphas_cumprod_v")[tb].unsqueeze(-1), self._table("log_one_minus_alphas_cumprod_v")[tb].unsqueeze(-1))
        v_pert = log_sample_categorical(log_qvt, u=(noise[1] if noise is not None else None))
        log_vt = torch.log(torch.nn.functional.one_hot(v_pert, self.num_classes).float().clamp(min=1e-30))
        if not eval_mode:       # classifier-free condition masking (:480-484; cond_mask_prob = 0 in the shipped configuration)
            keep = torch.bernoulli(torch.ones(num_graphs) * (1 - (self.cond_mask_prob or 0.0))).to(shape.device)
            shape = keep.view(-1, 1, 1) * shape
        if torch.is_grad_enabled():
            from .training import score_with_grad
            preds = score_with_grad(self, pos_pert, v_pert, batch, shape, t)
        else:
            preds = self(pos_pert, v_pert, batch, shape, time_step=t)
        pred_pos, pred_v = preds["pred_ligand_pos"], preds["pred_ligand_v"]
        # atom types: KL between the true and the model posterior, decoder NLL at t = 0
        log_recon = torch.nn.functional.log_softmax(pred_v, dim=-1)
        log_model = self._q_v_posterior(log_recon, log_vt, t, batch)
        log_true = self._q_v_posterior(log_v0, log_vt, t, batch)
        kl = (log_true.exp() * (log_true - log_model)).sum(dim=1)
        nll = -(log_v0.exp() * log_model).sum(dim=1)
        mask = (t == 0).float()[batch]
        kl_v = self._scatter_mean(mask * nll + (1.0 - mask) * kl, batch, num_graphs)
        loss_pos = self._scatter_mean(((pred_pos - pos) ** 2).sum(-1), batch, num_graphs)
        if self.loss_weight_type == "noise_level":
            loss_pos = torch.mean(self._table("loss_pos_step_weight")[t] * loss_pos)
        else:
            loss_pos = torch.mean(loss_pos)
        loss_v = torch.mean(kl_v)
        self.check_status()      # the reference raises from its indexing ops on a bad time step / atom type / batch vector
        return {"loss_pos": loss_pos, "loss_v": loss_v, "loss": loss_pos + loss_v * self.loss_v_weight, "x0": pos,
                "ligand_pos_perturbed": pos_pert, "ligand_v_perturbed": v_pert, "pred_ligand_pos": pred_pos,
                "pred_ligand_v": pred_v, "ligand_v_recon": torch.nn.functional.softmax(pred_v, dim=-1)}

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample_diffusion(self, init_ligand_pos, init_ligand_v, batch_ligand, ligand_shape, threshold_type=None,
                         threshold_args=None, num_steps=None, center_pos_mode=None, use_grad=False, grad_lr=1,
                         shape_AE=None, use_mesh_data=None, use_pointcloud_data=None, grad_step=500,
                         guide_stren=0, bounds=None, *, noise=None, seed=None, return_traj=True, use_graph=True,
                         first_step=0, guide_draws=None, _reuse_host_buffers=False, _slot=0, _async=False):
        """Reverse diffusion chain; same arguments and result dict as the reference.

        Extensions (keyword-only): ``noise=(eps, u)`` feeds host-chosen draws, eps (S,N,3) and u (S,N,C)
        device tensors in the reference's per-step order; otherwise device Philox noise keyed by
        ``seed`` (default: drawn from torch's CPU generator, so ``torch.manual_seed`` governs it).
        ``return_traj=False`` skips the per-step trajectories (the lists come back empty).
        ``first_step=s`` resumes a chain at reverse step s (t = T-1-s) from the given state and runs ``num_steps``
        steps from there (the windowed full-length parity test; the reference always starts at T-1).
        ``use_pointcloud_data=(point_clouds, kdtree, radius)`` with ``grad_step`` is the reference's point-cloud shape
        guidance (``:583-586,699-740``) as a device kernel inside the step (the KD-tree is not used: brute-force float64
        3-nearest search); ``guide_draws`` (S,5,N) float64 feeds the recorded ``np.random.random`` draws (parity mode).
        Private to shapemol_amd.sampling: ``_slot`` picks one of the model's library contexts (own workspace and captured
        graphs), ``_async=True`` returns a handle right after the chain has been enqueued; its ``.result()`` waits and
        builds the dict (chains on different slots then run side by side, and a finished chain's trajectories are
        unbatched and copied while the next one runs).
        """
        if use_mesh_data is not None or use_grad:
            raise NotImplementedError("mesh / gradient shape guidance is outside the accelerated path")
        if self.cond_mask_prob == 0:
            assert guide_stren == 0
        if guide_stren:
            raise NotImplementedError("classifier-free guidance is unreachable in the reference (SURVEY.md F10)")
        if center_pos_mode not in (None, "none", "center"):
            raise NotImplementedError(center_pos_mode)       # as center_pos() of the reference (:52-60)
        if num_steps is None:
            num_steps = self.num_timesteps
        print('sample center pos mode: ', center_pos_mode)

        pos = _check_device_tensor("init_ligand_pos", init_ligand_pos, torch.float32)
        v = _check_device_tensor("init_ligand_v", init_ligand_v, torch.int64)
        batch = _check_device_tensor("batch_ligand", batch_ligand, torch.int64)
        shape = _check_device_tensor("ligand_shape", ligand_shape, torch.float32).view(-1, self.dims.S, 3)
        n, b, cc, dev = pos.shape[0], shape.shape[0], self.dims.C, pos.device
        offset = None
        if center_pos_mode == "center":
            # the chain runs on coordinates centred per molecule; the offset goes back onto `pos` and `pos_traj` (reference :547,
            # :675-684; `pos_cond_traj`, the raw network outputs, stays in the centred frame there too)
            cnt = torch.bincount(batch, minlength=b).clamp(min=1).to(torch.float32)
            offset = (torch.zeros((b, 3), dtype=torch.float32, device=dev).index_add_(0, batch, pos) / cnt[:, None])[batch]
            pos = pos - offset
        ctx = self._context(dev, _slot)
        self._sync_bn_mode(ctx, _slot)
        lib = _lib.load()
        eps = u = None
        if noise is not None:
            eps = _check_device_tensor("noise[0]", noise[0], torch.float32)
            u = _check_device_tensor("noise[1]", noise[1], torch.float32)
            if tuple(eps.shape) != (num_steps, n, 3) or tuple(u.shape) != (num_steps, n, cc):
                raise ValueError("noise must be (eps (S,N,3), u (S,N,C))")
        guided = use_pointcloud_data is not None
        if seed is None:
            if noise is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            elif guided and guide_draws is None:
                # host-fed chain noise, device-drawn guidance uniforms: a fresh key per call from numpy's global generator (the
                # one the reference's guidance draws from, molopt_score_model.py:719), leaving torch's generator -- which the
                # host_rng mode of the sampling driver replays draw by draw -- untouched
                seed = int(np.random.randint(0, 2 ** 62, dtype=np.int64))
            else:
                seed = 0
        # the largest molecule of the batch lets the library fold the per-layer coordinate update into the next attention
        # kernel (one tiny synchronising reduction per chain; the reference synchronises at every step)
        _lib.check(lib.shapemol_set_option(ctx, b"max_mol_atoms", int(torch.bincount(batch).max().item()) if n else 0), "shapemol_set_option")
        gd = None
        if guided:
            cloud = np.ascontiguousarray(np.asarray(use_pointcloud_data[0], dtype=np.float64).reshape(-1, 3))
            if guide_draws is not None:
                gd = _check_device_tensor("guide_draws", guide_draws, torch.float64)
                if tuple(gd.shape) != (num_steps, 5, n):
                    raise ValueError("guide_draws must be (S, 5, N) float64")
            _lib.check(lib.shapemol_set_guidance(ctx, cloud.ctypes.data_as(C.c_void_p), cloud.shape[0], float(use_pointcloud_data[2]),
                                                 int(grad_step), _ptr(gd)), "shapemol_set_guidance")
        tr = _lib.Traj()
        bufs = {}
        if return_traj:
            for name, shp, dt in (("pos_traj", (num_steps, n, 3), torch.float32), ("v_traj", (num_steps, n), torch.int64),
                                  ("v0_traj", (num_steps, n, cc), torch.float32), ("vt_traj", (num_steps, n, cc), torch.float32),
                                  ("pos_cond_traj", (num_steps, n, 3), torch.float32),
                                  ("v_cond_traj", (num_steps, n, cc), torch.float32)):
                bufs[name] = torch.empty(shp, dtype=dt, device=dev)
                setattr(tr, name, bufs[name].data_ptr())
        out_pos = torch.empty((n, 3), dtype=torch.float32, device=dev)
        out_v = torch.empty((n,), dtype=torch.int64, device=dev)
        pending = _PendingChain(self, ctx, dev, guided, bufs, out_pos, out_v, return_traj, _reuse_host_buffers,
                                keep=(pos, v, batch, shape, eps, u, gd), offset=offset)
        try:
            with torch.cuda.device(dev):
                cur = torch.cuda.current_stream(dev)
                # hipGraph capture is not allowed on the legacy default stream: run the chain on a side stream
                side = self._side_stream(dev, _slot)
                side.wait_stream(cur)
                if first_step:
                    _lib.check(lib.shapemol_set_option(ctx, b"first_step", int(first_step)), "shapemol_set_option")
                try:
                    rc = lib.shapemol_sample(ctx, _ptr(pos), _ptr(v), _ptr(batch), n, b, _ptr(shape), int(num_steps),
                                             _ptr(eps), _ptr(u), C.c_uint64(seed), C.byref(tr), _ptr(out_pos), _ptr(out_v),
                                             1 if use_graph else 0, _stream_ptr(side))
                finally:
                    if first_step:
                        lib.shapemol_set_option(ctx, b"first_step", 0)
                pending.side, pending.cur = side, cur
            _lib.check(rc, "shapemol_sample")
        except BaseException:
            pending.abandon()
            raise
        return pending if _async else pending.result()

    def pointcloud_shape_guidance(self, use_pointcloud_data, pred_ligand_pos, k=3, ratio=0.2, *, draws=None, seed=None):
        """Method form of the module-level :func:`pointcloud_shape_guidance` (kept for callers that hold a model)."""
        return pointcloud_shape_guidance(use_pointcloud_data, pred_ligand_pos, k=k, ratio=ratio, draws=draws, seed=seed)

    def check_status(self):
        """Synchronise and raise if the last forward / sample_diffusion saw an invalid input (batch vector not sorted
        or >= the number of shapes, atom type or time step out of range).  forward() stays asynchronous, so its flags
        are read here or at the next sample_diffusion; the reference raises from torch's indexing ops instead."""
        if self._ctx is not None:
            _lib.check(_lib.load().shapemol_status(self._ctx, None), "input check")

    def _to_host(self, t, cache_key=None):
        """Device tensor -> host tensor with one pinned-memory DMA (falls back to a pageable copy).  With a cache key the
        pinned buffer is kept and handed out again by the next call of the same shape (pinning 1 GB costs as much as the copy)."""
        cache = self.__dict__.setdefault("_pinned", {})
        h = cache.get(cache_key) if cache_key is not None else None
        if h is None or h.shape != t.shape or h.dtype != t.dtype:
            try:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            except RuntimeError:
                return t.cpu()
            if cache_key is not None:
                cache[cache_key] = h
        h.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return h

    def _side_stream(self, dev, slot=0):
        streams = self.__dict__.setdefault("_streams", {})
        s = streams.get(slot)
        if s is None or s.device != dev:
            s = torch.cuda.Stream(device=dev)
            streams[slot] = s
        return s


class _PendingChain:
    """A reverse chain that has been enqueued on its context's side stream (ScorePosNet3D.sample_diffusion(_async=True));
    result() waits for it, reads the status flags and builds the reference's result dict."""

    def __init__(self, model, ctx, dev, guided, bufs, out_pos, out_v, return_traj, reuse, keep, offset=None):
        self.offset = offset                 # (N, 3) per-atom centre of its molecule (center_pos_mode='center') or None
        self.model, self.ctx, self.dev, self.guided, self.bufs = model, ctx, dev, guided, bufs
        self.out_pos, self.out_v, self.return_traj, self.reuse, self.keep = out_pos, out_v, return_traj, reuse, keep
        self.side = self.cur = None
        self.done = False

    def _drop_guidance(self):
        if self.guided:      # whatever happened, the context must not keep the cloud (and the caller-owned draws pointer) installed
            self.guided = False
            torch.cuda.synchronize(self.dev)
            _lib.check(_lib.load().shapemol_set_guidance(self.ctx, None, 0, 0.0, 0, None), "shapemol_set_guidance")

    def abandon(self):
        self.done = True
        self._drop_guidance()

    def result(self):
        assert not self.done, "result() of a chain can be taken once"
        self.done = True
        try:
            # the reference returns finished results; also the point where the input flags are read (only this chain's stream
            # is waited for: another slot's chain may be running beside it)
            _lib.check(_lib.load().shapemol_status_stream(self.ctx, None, _stream_ptr(self.side)), "input check")
        finally:
            self._drop_guidance()
        self.cur.wait_stream(self.side)
        m, bufs = self.model, self.bufs
        for t_ in (*self.keep, self.out_pos, self.out_v, *bufs.values()):
            if t_ is not None:
                t_.record_stream(self.side)
        if self.offset is not None:
            self.out_pos += self.offset
            if "pos_traj" in bufs:
                bufs["pos_traj"] += self.offset.unsqueeze(0)
        res = {"pos": self.out_pos, "v": self.out_v, "pos_uncond_traj": [], "v_uncond_traj": []}
        if self.return_traj and self.reuse == "device":
            # private to shapemol_amd.sampling: hand out the (S, N, ...) device buffers; the driver reorders them on the
            # device and copies each to the host once
            res.update(pos_traj=[], v_traj=[], v0_traj=[], vt_traj=[], pos_cond_traj=[], v_cond_traj=[])
            res["_stacked"] = dict(bufs)
        elif self.return_traj:
            # one D2H copy per trajectory, through pinned staging buffers (the reference copies step by step, :671-681)
            # (_reuse_host_buffers: private to shapemol_amd.sampling, which consumes the host tensors before the next call)
            host = {k: m._to_host(bufs[k], k if self.reuse else None) for k in ("pos_traj", "v_traj", "v0_traj", "vt_traj")}
            res.update(pos_traj=list(host["pos_traj"].unbind(0)), v_traj=list(host["v_traj"].unbind(0)),
                       v0_traj=list(host["v0_traj"].unbind(0)), vt_traj=list(host["vt_traj"].unbind(0)),
                       pos_cond_traj=list(bufs["pos_cond_traj"].unbind(0)), v_cond_traj=list(bufs["v_cond_traj"].unbind(0)))
            # not a key of the reference: the same trajectories as whole (S, N, ...) tensors, for callers that unbatch
            # them at once (shapemol_amd.sampling) instead of re-stacking the per-step lists
            res["_stacked"] = dict(host, pos_cond_traj=bufs["pos_cond_traj"], v_cond_traj=bufs["v_cond_traj"])
        else:
            res.update(pos_traj=[], v_traj=[], v0_traj=[], vt_traj=[], pos_cond_traj=[], v_cond_traj=[])
        return res


def pointcloud_shape_guidance(use_pointcloud_data, pred_ligand_pos, k=3, ratio=0.2, *, draws=None, seed=None):
    """``pointcloud_shape_guidance`` of the reference (module level there too, ``models/molopt_score_model.py:699-740``) as
    one device kernel: atoms of ``pred_ligand_pos`` (N,3) whose ``k`` = 3 nearest cloud points are on average farther than
    ``radius`` are pulled towards their mean by ``u * (0.8 - ratio) + ratio``, up to five times; the tensor is updated in
    place and returned.  ``use_pointcloud_data = (point_clouds, kdtree, radius)`` as there (the KD-tree is not used:
    brute-force float64 search on the device).
    Extensions (keyword-only): ``draws`` (5,N) float64 device tensor = the uniform of every (iteration, atom) (parity
    mode); otherwise device Philox keyed by ``seed`` (default: drawn from numpy's global generator, which the reference's
    function draws its uniforms from)."""
    if k != 3:
        raise NotImplementedError("the device kernel searches the reference's default k = 3 nearest cloud points")
    pos = _check_device_tensor("pred_ligand_pos", pred_ligand_pos, torch.float32)
    if pos.data_ptr() != pred_ligand_pos.data_ptr() or pos.dim() != 2 or pos.shape[1] != 3:
        raise ValueError("pred_ligand_pos must be a contiguous (N, 3) tensor (it is updated in place)")
    cloud = np.ascontiguousarray(np.asarray(use_pointcloud_data[0], dtype=np.float64).reshape(-1, 3))
    gd = None if draws is None else _check_device_tensor("draws", draws, torch.float64)
    if gd is not None and tuple(gd.shape) != (5, pos.shape[0]):
        raise ValueError("draws must be (5, N) float64")
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 62, dtype=np.int64)) if gd is None else 0
    if pos.shape[0] == 0:
        return pred_ligand_pos
    with torch.cuda.device(pos.device):
        rc = _lib.load().shapemol_pointcloud_guidance(cloud.ctypes.data_as(C.c_void_p), cloud.shape[0], float(use_pointcloud_data[2]),
                                                      float(ratio), _ptr(pos), pos.shape[0], _ptr(gd), C.c_uint64(seed),
                                                      _stream_ptr(torch.cuda.current_stream(pos.device)))
    _lib.check(rc, "shapemol_pointcloud_guidance")
    return pred_ligand_pos


def log_sample_categorical(logits, *, u=None, seed=None):
    """Gumbel-argmax categorical sample of each row of ``logits`` (device tensor) -> LongTensor."""
    lg = _check_device_tensor("logits", logits, torch.float32)
    n, c = lg.shape
    if u is not None:
        u = _check_device_tensor("u", u, torch.float32)
    if seed is None:     # host-fed uniforms must not disturb the host generator (the driver's seed-only parity mode)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if u is None else 0
    out = torch.empty((n,), dtype=torch.int64, device=lg.device)
    with torch.cuda.device(lg.device):
        rc = _lib.load().shapemol_log_sample_categorical(None, _ptr(lg), _ptr(u), n, c, C.c_uint64(seed), _ptr(out),
                                                         _stream_ptr(torch.cuda.current_stream(lg.device)))
    _lib.check(rc, "shapemol_log_sample_categorical")
    return out
