"""Multi-GPU sharding of a sampling job: one process per GPU, whole batches per rank.

A batch is the atomic unit because the VN batch-norm runs on batch statistics, so
ranks never exchange anything inside a chain; the reference itself processes ``num_samples`` in
independent chunks of ``--batch_size`` (scripts/sample_diffusion.py:57-65).  The only collective
is the final gather of the generated molecules (RCCL over xGMI on the GPU box: backend "nccl";
"gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_batches(num_batches, rank, world_size):
    """Batch indices owned by `rank` (contiguous blocks, sizes differing by at most one)."""
    base, rem = divmod(num_batches, world_size)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def gather_molecules(pos, v, counts, group=None):
    """All-gather the final molecules of every rank.

    pos (N_r,3) f32, v (N_r,) i64, counts (B_r,) i64 atoms per molecule on this rank.
    Returns (pos_all, v_all, counts_all) concatenated in rank order on every rank.
    Ragged sizes are handled by padding to the largest rank (one all_gather of the sizes first)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return pos, v, counts
    ws = dist.get_world_size(group)
    dev = pos.device
    sizes = torch.tensor([pos.shape[0], counts.shape[0]], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(ws)]
    dist.all_gather(all_sizes, sizes, group=group)
    max_n = int(max(s[0] for s in all_sizes))
    max_b = int(max(s[1] for s in all_sizes))
    # one payload per rank: positions, types and counts packed into a single int64/float32 pair
    pad_pos = torch.zeros((max_n, 3), dtype=pos.dtype, device=dev); pad_pos[:pos.shape[0]] = pos
    pad_iv = torch.zeros((max_n + max_b,), dtype=torch.int64, device=dev)
    pad_iv[:v.shape[0]] = v
    pad_iv[max_n:max_n + counts.shape[0]] = counts
    g_pos = [torch.empty_like(pad_pos) for _ in range(ws)]
    g_iv = [torch.empty_like(pad_iv) for _ in range(ws)]
    dist.all_gather(g_pos, pad_pos, group=group)
    dist.all_gather(g_iv, pad_iv, group=group)
    out_p, out_v, out_c = [], [], []
    for r in range(ws):
        n_r, b_r = int(all_sizes[r][0]), int(all_sizes[r][1])
        out_p.append(g_pos[r][:n_r]); out_v.append(g_iv[r][:n_r]); out_c.append(g_iv[r][max_n:max_n + b_r])
    return torch.cat(out_p), torch.cat(out_v), torch.cat(out_c)
