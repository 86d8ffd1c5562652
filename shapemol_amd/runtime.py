"""Persistent-buffer chain runner over the C ABI (used by bench.py and the sampling driver).

``ScorePosNet3D.sample_diffusion`` allocates fresh result tensors per call, as the reference does.
A serving loop that samples batch after batch wants the opposite: buffers allocated once, the
captured hipGraph of one chain step reused, no host work in the loop.  ``ChainRunner`` is that.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib


class ChainRunner:
    def __init__(self, model, n_atoms, n_mols, max_steps, keep_traj=True, device=None):
        self.model = model
        self.dev = torch.device(device) if device is not None else next(model.parameters()).device
        if self.dev.type != "cuda":
            raise RuntimeError("ChainRunner needs the model on a HIP device")
        d = model.dims
        self.n, self.b, self.max_steps, self.C = int(n_atoms), int(n_mols), int(max_steps), d.C
        f32, i64 = torch.float32, torch.int64
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.dev)  # noqa: E731
        self.pos0, self.v0 = z((self.n, 3), f32), z((self.n,), i64)
        self.batch, self.shape = z((self.n,), i64), z((self.b, d.S, 3), f32)
        self.out_pos, self.out_v = z((self.n, 3), f32), z((self.n,), i64)
        self.traj = _lib.Traj()
        self.bufs = {}
        if keep_traj:
            S, n, c = self.max_steps, self.n, d.C
            for name, shp, dt in (("pos_traj", (S, n, 3), f32), ("v_traj", (S, n), i64), ("v0_traj", (S, n, c), f32),
                                  ("vt_traj", (S, n, c), f32), ("pos_cond_traj", (S, n, 3), f32), ("v_cond_traj", (S, n, c), f32)):
                self.bufs[name] = z(shp, dt)
                setattr(self.traj, name, self.bufs[name].data_ptr())
        self.eps = self.u = None
        self.stream = torch.cuda.Stream(device=self.dev)
        self.ctx = model._context(self.dev)
        _lib.check(_lib.load().shapemol_reserve(self.ctx, self.n, self.b), "shapemol_reserve")
        # a library context owns ONE workspace: chains of the same model must not be in flight on two streams at once.  The
        # registry holds weak references: a runner's buffers (GBs at B = 1024 with trajectories) go when the runner goes
        model.__dict__.setdefault("_runners", weakref.WeakSet()).add(self)

    def close(self):
        """Wait for the runner's chain and leave the model's registry (its buffers are freed with the last reference)."""
        self.stream.synchronize()
        self.model.__dict__.get("_runners", set()).discard(self)

    def load_batch(self, init_pos, init_v, batch, shape):
        self.pos0.copy_(torch.as_tensor(init_pos)); self.v0.copy_(torch.as_tensor(init_v))
        self.batch.copy_(torch.as_tensor(batch)); self.shape.copy_(torch.as_tensor(shape).reshape(self.b, -1, 3))
        counts = np.bincount(np.asarray(torch.as_tensor(batch).cpu()))
        _lib.check(_lib.load().shapemol_set_option(self.ctx, b"max_mol_atoms", int(counts.max()) if counts.size else 0), "shapemol_set_option")

    def set_noise(self, eps, u):
        """Host-fed noise for the whole chain (parity mode); None, None -> device Philox."""
        self.eps = None if eps is None else torch.as_tensor(eps, dtype=torch.float32).to(self.dev).contiguous()
        self.u = None if u is None else torch.as_tensor(u, dtype=torch.float32).to(self.dev).contiguous()

    def run(self, num_steps, seed=0, use_graph=True):
        """Enqueue a chain of `num_steps` reverse steps on the runner's stream (no host sync)."""
        assert 1 <= num_steps <= self.max_steps
        for other in list(self.model.__dict__.get("_runners", ())):
            if other is not self and not other.stream.query():
                raise RuntimeError("another ChainRunner of the same model still has a chain in flight: a context has one workspace; "
                                   "synchronize() it first, or give the second runner its own model (own context)")
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
        self.model._sync_bn_mode(self.ctx)      # module.eval() / .train() since the last call (shared context)
        with torch.cuda.device(self.dev):
            rc = _lib.load().shapemol_sample(self.ctx, p(self.pos0), p(self.v0), p(self.batch), self.n, self.b, p(self.shape),
                                             int(num_steps), p(self.eps), p(self.u), C.c_uint64(seed), C.byref(self.traj),
                                             p(self.out_pos), p(self.out_v), 1 if use_graph else 0,
                                             C.c_void_p(self.stream.cuda_stream))
        _lib.check(rc, "shapemol_sample")

    def synchronize(self):
        """Wait for the chain and raise if the library flagged an invalid input (unsorted batch vector, atom type or
        time step out of range) or a timed-out grid barrier (option vn_fuse = 1)."""
        self.stream.synchronize()
        _lib.check(_lib.load().shapemol_status(self.ctx, None), "chain")

    def profile(self, num_steps, seed=0):
        """Per-kernel-class launch time (HIP events on the launch stream, eager launches).
        Returns {class: (total_ms, launches)}."""
        lib = _lib.load()
        _lib.check(lib.shapemol_profile_begin(self.ctx), "shapemol_profile_begin")
        self.run(num_steps, seed=seed, use_graph=False)
        cap = 32
        names = ((C.c_char * 32) * cap)()
        ms = (C.c_double * cap)()
        cnt = (C.c_int64 * cap)()
        k = lib.shapemol_profile_end(self.ctx, names, ms, cnt, cap)
        if k < 0:
            _lib.check(1, "shapemol_profile_end")
        return {names[i].value.decode(): (ms[i], cnt[i]) for i in range(k)}
