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


def gather_molecules(pos, v, counts, group=None, _single_rank_too=False):
    """All-gather the final molecules of every rank.

    pos (N_r,3) f32, v (N_r,) i64, counts (B_r,) i64 atoms per molecule on this rank (a rank may own nothing).
    Returns (pos_all, v_all, counts_all) concatenated in rank order on every rank.
    Two collectives: one all-gather of the (N_r, B_r) sizes, then ONE padded ``all_gather_into_tensor`` of a packed
    int32 buffer per rank: [pos bits (3 max_n) | atom types (max_n) | counts (max_b)] -- atom types (< num_classes)
    and atom counts fit 32 bits, coordinates travel as their bit patterns."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _single_rank_too):
        return pos, v, counts          # (_single_rank_too: tests and `bench.py --force-collective` push a 1-rank group through the
                                       #  packing / all_gather_into_tensor / unpacking below, e.g. RCCL on a one-GPU box)
    ws = dist.get_world_size(group)
    out_dev = pos.device
    if dist.get_backend(group) == "gloo" and pos.is_cuda:      # rehearsals on one GPU (bench.py --backend gloo): collectives on the host
        pos, v, counts = pos.cpu(), v.cpu(), counts.cpu()
    dev = pos.device
    n_r, b_r = int(pos.shape[0]), int(counts.shape[0])
    sizes = torch.tensor([n_r, b_r], dtype=torch.int64, device=dev)
    all_sizes = torch.empty((ws * 2,), dtype=torch.int64, device=dev)      # flat outputs: gloo and RCCL both accept them
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    all_sizes = all_sizes.view(ws, 2).cpu()
    max_n, max_b = int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max())
    payload = torch.zeros((4 * max_n + max_b,), dtype=torch.int32, device=dev)
    payload[:3 * n_r] = pos.to(torch.float32).contiguous().view(torch.int32).reshape(-1)
    payload[3 * max_n:3 * max_n + n_r] = v.to(torch.int32)
    payload[4 * max_n:4 * max_n + b_r] = counts.to(torch.int32)
    gathered = torch.empty((ws * payload.numel(),), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, payload, group=group)
    gathered = gathered.view(ws, payload.numel())
    out_p, out_v, out_c = [], [], []
    for r in range(ws):
        nr, br = int(all_sizes[r, 0]), int(all_sizes[r, 1])
        out_p.append(gathered[r, :3 * nr].view(torch.float32).reshape(nr, 3))
        out_v.append(gathered[r, 3 * max_n:3 * max_n + nr].to(torch.int64))
        out_c.append(gathered[r, 4 * max_n:4 * max_n + br].to(torch.int64))
    return torch.cat(out_p).to(out_dev), torch.cat(out_v).to(out_dev), torch.cat(out_c).to(out_dev)
