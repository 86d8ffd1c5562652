"""Drop-in for the encoder half of the reference's shape auto-encoder: ``VN_DGCNN_Encoder``
(``models/shape_pointcloud_modelAE.py:207-255``), the module that turns a molecule's surface point cloud into the
``shape_emb`` (32, 3) the diffusion model is conditioned on (``utils/shape.py:240-283``).

Same constructor arguments and forward contract (``input (B, 1, N, 3) -> (B, latent_dim, 3)``).  State-dict keys follow
the reference for ``conv_pos`` and ``conv_c``; the DGCNN blocks are registered here as ``blocks.{i}.*`` -- in the
reference they live in a plain Python list, so they are neither saved in ``se_model.pt`` nor moved by ``.to()``
(SURVEY.md F5): loading that checkpoint with ``strict=False`` leaves them at their initial values, exactly as the
reference leaves them at random initial values.  Batch-norm uses batch statistics, as the reference's auto-encoder does
(it is never switched to eval mode, ``utils/shape.py:226-238``); running statistics are carried but not updated.
All arithmetic runs in libshapemol_hip.so (``shapemol_se_*``, hand-written HIP); there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib

__all__ = ["VN_DGCNN_Encoder"]


class _VNLinearLeakyReLU(nn.Module):
    """Parameter container with the reference's names (models/shape_vn_layers.py:95-124)."""

    def __init__(self, cin, cout, share_nonlinearity=False):
        super().__init__()
        self.map_to_feat = nn.Linear(cin, cout, bias=False)
        self.batchnorm = nn.Module()
        self.batchnorm.bn = nn.BatchNorm1d(cout)          # parameter / buffer names as BatchNorm1d / BatchNorm2d alike
        self.map_to_dir = nn.Linear(cin, 1 if share_nonlinearity else cout, bias=False)


class VN_DGCNN_Encoder(nn.Module):
    def __init__(self, hidden_dim, latent_dim, layer_num, num_k):
        super().__init__()
        self.hidden_dim, self.latent_dim, self.layer_num, self.num_k = hidden_dim, latent_dim, layer_num, num_k
        self.conv_pos = _VNLinearLeakyReLU(2, hidden_dim)
        self.blocks = nn.ModuleList([_VNLinearLeakyReLU(2 * hidden_dim, hidden_dim) for _ in range(layer_num)])
        self.conv_c = _VNLinearLeakyReLU(layer_num * hidden_dim, latent_dim, share_nonlinearity=True)
        self._ctx, self._key = None, None

    def _pack(self):
        parts = []
        for m in [self.conv_pos, *self.blocks, self.conv_c]:
            parts += [m.map_to_feat.weight, m.batchnorm.bn.weight, m.batchnorm.bn.bias, m.map_to_dir.weight]
        return np.concatenate([p.detach().cpu().numpy().astype(np.float32).reshape(-1) for p in parts])

    def _context(self, device):
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._ctx is not None and key == self._key:
            return self._ctx
        lib = _lib.load()
        self._release()
        w = self._pack()
        ctx = C.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        _lib.check(lib.shapemol_se_create(self.hidden_dim, self.latent_dim, self.layer_num, self.num_k, w.ctypes.data_as(C.c_void_p),
                                          w.size, idx, C.byref(ctx)), "shapemol_se_create")
        self._ctx, self._key = ctx, key
        return ctx

    def _release(self):
        if getattr(self, "_ctx", None) is not None:
            _lib.load().shapemol_se_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, input):
        """input (B, 1, N, 3) (or (B, N, 3)) float32 device tensor -> latent (B, latent_dim, 3)."""
        if not isinstance(input, torch.Tensor) or not input.is_cuda:
            raise RuntimeError("input must be a tensor on a HIP device (shapemol_amd has no CPU path)")
        x = input.reshape(input.shape[0], -1, 3).to(torch.float32).contiguous()
        b, n = x.shape[0], x.shape[1]
        dev = x.device
        out = torch.empty((b, self.latent_dim, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().shapemol_se_encode(self._context(dev), C.c_void_p(x.data_ptr()), b, n, C.c_void_p(out.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "shapemol_se_encode")
        return out
