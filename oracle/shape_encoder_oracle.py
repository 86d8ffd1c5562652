"""CPU oracle of the frozen shape encoder -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

A functional torch-CPU float32 restatement of ``VN_DGCNN_Encoder.forward``
(/root/reference/models/shape_pointcloud_modelAE.py:231-255) over a flat state dict, with
``get_graph_feature_cross`` / ``knn`` (/root/reference/models/shape_vn_layers.py:257-292) and
``VNLinearLeakyReLU`` with ``VNBatchNorm`` in train mode (:41-61,95-124).  Pinned by tests/golden/shape_encoder.npz,
which the reference class itself produced (tests/golden/make_golden_r2.py se).
"""
import torch

EPS = 1e-6


def _knn(x, k):
    """x (B, D, N) -> idx (B, N, k): topk of -|x_i - x_j|^2 in the reference's arithmetic (:288-292)."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    return (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]


def _graph_feature(x, k):
    """x (B, C, 3, N) -> (B, 2C, 3, N, k): [x_j - x_i | x_i] (:257-286, if_cross False)."""
    B, C, _, N = x.shape
    flat = x.reshape(B, C * 3, N)
    idx = _knn(flat, k) + torch.arange(B).view(-1, 1, 1) * N
    pts = flat.transpose(2, 1).reshape(B * N, C * 3)
    nb = pts[idx.reshape(-1)].view(B, N, k, C, 3)
    ctr = pts.view(B, N, 1, C, 3).expand(-1, -1, k, -1, -1)
    return torch.cat((nb - ctr, ctr), dim=3).permute(0, 3, 4, 1, 2).contiguous()


def _vn_linear_lrelu(sd, p, x, slope=0.2):
    """VNLinearLeakyReLU on x (B, Cin, 3, ...) with batch-statistics VNBatchNorm (:41-61,95-124)."""
    lin = lambda w, t: torch.matmul(t.transpose(1, -1), w.t()).transpose(1, -1)   # noqa: E731
    q = lin(sd[p + ".map_to_feat.weight"], x)
    norm = torch.norm(q, dim=2) + EPS
    dims = [0] + list(range(2, norm.dim()))
    mean = norm.mean(dim=dims, keepdim=True)
    var = norm.var(dim=dims, unbiased=False, keepdim=True)
    shape = [1, -1] + [1] * (norm.dim() - 2)
    nbn = (norm - mean) / torch.sqrt(var + 1e-5) * sd[p + ".batchnorm.bn.weight"].view(shape) + sd[p + ".batchnorm.bn.bias"].view(shape)
    q = q / norm.unsqueeze(2) * nbn.unsqueeze(2)
    d = lin(sd[p + ".map_to_dir.weight"], x)
    dot = (q * d).sum(2, keepdim=True)
    mask = (dot >= 0).float()
    dsq = (d * d).sum(2, keepdim=True)
    return slope * q + (1 - slope) * (mask * q + (1 - mask) * (q - (dot / (dsq + EPS)) * d))


@torch.no_grad()
def encode(sd, points, layer_num=4, k=20):
    """points (B, N, 3) -> latent (B, latent, 3)."""
    x = points.unsqueeze(1).transpose(2, 3)                       # (B, 1, 3, N)
    h = _vn_linear_lrelu(sd, "conv_pos", _graph_feature(x, k)).mean(dim=-1)
    hs = []
    for i in range(layer_num):
        h = _vn_linear_lrelu(sd, f"blocks.{i}", _graph_feature(h, k)).mean(dim=-1)
        hs.append(h)
    return _vn_linear_lrelu(sd, "conv_c", torch.cat(hs, dim=1)).mean(dim=-1)
