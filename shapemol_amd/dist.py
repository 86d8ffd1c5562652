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


class GatherPlan:
    """What every rank must know before the molecules of a job can be gathered with ONE collective: the (atoms, molecules)
    sizes of all ranks, exchanged here (one small all-gather -- the atom counts of a sampling job are drawn before its chains
    start, scripts/sample_diffusion.py:66-72, so this happens at job set-up, not behind the chains), and from them the padded
    payload layout and the index maps that unpack the gathered payloads with three ``index_select``s."""

    def __init__(self, n_r, b_r, device, group=None):
        ws = dist.get_world_size(group)
        self.group, self.ws, self.n_r, self.b_r = group, ws, int(n_r), int(b_r)
        self.host = dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"    # rehearsal: collectives on the host
        dev = torch.device("cpu") if self.host else torch.device(device)
        sizes = torch.tensor([self.n_r, self.b_r], dtype=torch.int64, device=dev)
        all_sizes = torch.empty((ws * 2,), dtype=torch.int64, device=dev)      # flat outputs: gloo and RCCL both accept them
        dist.all_gather_into_tensor(all_sizes, sizes, group=group)
        self.all_sizes = all_sizes.view(ws, 2).cpu()
        self.max_n, self.max_b = int(self.all_sizes[:, 0].max()), int(self.all_sizes[:, 1].max())
        self.width = 4 * self.max_n + self.max_b
        atom_idx, mol_idx = [], []
        for r in range(ws):
            nr, br = int(self.all_sizes[r, 0]), int(self.all_sizes[r, 1])
            atom_idx.append(r * self.width + torch.arange(nr, dtype=torch.int64))
            mol_idx.append(r * self.width + 4 * self.max_n + torch.arange(br, dtype=torch.int64))
        atom_idx = torch.cat(atom_idx) if atom_idx else torch.zeros(0, dtype=torch.int64)
        # element indices into the flat gathered int32 buffer: coordinates (3 per atom), atom types, atom counts
        rank_of_atom = torch.div(atom_idx, self.width, rounding_mode="floor")
        local = atom_idx - rank_of_atom * self.width
        self.pos_idx = ((rank_of_atom * self.width + 3 * local)[:, None] + torch.arange(3)[None, :]).reshape(-1).to(dev)
        self.v_idx = (rank_of_atom * self.width + 3 * self.max_n + local).to(dev)
        self.c_idx = (torch.cat(mol_idx) if mol_idx else torch.zeros(0, dtype=torch.int64)).to(dev)
        self.dev = dev
        self.payload = torch.zeros((self.width,), dtype=torch.int32, device=dev)
        self.gathered = torch.empty((ws * self.width,), dtype=torch.int32, device=dev)


def gather_molecules(pos, v, counts, group=None, _single_rank_too=False, plan=None):
    """All-gather the final molecules of every rank.

    pos (N_r,3) f32, v (N_r,) i64, counts (B_r,) i64 atoms per molecule on this rank (a rank may own nothing).
    Returns (pos_all, v_all, counts_all) concatenated in rank order on every rank.
    ONE padded ``all_gather_into_tensor`` of a packed int32 buffer per rank: [pos bits (3 max_n) | atom types (max_n) |
    counts (max_b)] -- atom types (< num_classes) and atom counts fit 32 bits, coordinates travel as their bit patterns.
    The sizes every rank needs for the layout come from ``plan`` (a :class:`GatherPlan` made at job set-up); without one it
    is built here, which costs a second, small collective and a host synchronisation."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _single_rank_too):
        return pos, v, counts          # (_single_rank_too: tests and `bench.py --force-collective` push a 1-rank group through the
                                       #  packing / all_gather_into_tensor / unpacking below, e.g. RCCL on a one-GPU box)
    out_dev = pos.device
    n_r, b_r = int(pos.shape[0]), int(counts.shape[0])
    if plan is None:
        plan = GatherPlan(n_r, b_r, pos.device, group)
    elif (plan.n_r, plan.b_r) != (n_r, b_r) or plan.group is not group:
        raise ValueError("gather_molecules: the plan was made for other sizes or another group")
    if plan.host:
        pos, v, counts = pos.cpu(), v.cpu(), counts.cpu()
    payload, max_n = plan.payload, plan.max_n
    payload[:3 * n_r] = pos.to(torch.float32).contiguous().view(torch.int32).reshape(-1)
    payload[3 * max_n:3 * max_n + n_r] = v.to(torch.int32)
    payload[4 * max_n:4 * max_n + b_r] = counts.to(torch.int32)
    dist.all_gather_into_tensor(plan.gathered, payload, group=group)
    g = plan.gathered
    out_p = g.index_select(0, plan.pos_idx).view(torch.float32).reshape(-1, 3)
    out_v = g.index_select(0, plan.v_idx).to(torch.int64)
    out_c = g.index_select(0, plan.c_idx).to(torch.int64)
    return out_p.to(out_dev), out_v.to(out_dev), out_c.to(out_dev)


def sample_diffusion_ligand_sharded(model, shape_emb, num_samples, batch_size=16, *, job_seed, group=None, **kw):
    """One sampling job (``scripts/sample_diffusion.py:47-162`` for one shape condition) spread over the ranks of a process group.

    The job's ``ceil(num_samples / batch_size)`` batches -- the chunks the reference loops over, each coupled internally by the
    batch-norm and independent of the others -- are dealt to the ranks in contiguous blocks (:func:`shard_batches`); every rank
    runs its batches through :func:`shapemol_amd.sampling.sample_diffusion_ligand` (two chains in flight, trajectories delivered
    to this rank's host), and ONE collective at the end gathers the generated molecules.  Every batch's host random numbers
    (atom counts, initial positions, the chain's noise key) are keyed by ``job_seed + batch index``: the molecules of the job do
    not depend on the number of ranks.

    Returns ``(outputs, pred_pos, pred_v)``: ``outputs`` = the reference's 9-tuple for THIS rank's batches (trajectories stay
    rank-local, as the reference writes one result file per process), ``pred_pos`` / ``pred_v`` = per-molecule arrays of the
    WHOLE job in job order, identical on every rank.  ``kw``: the other arguments of ``sample_diffusion_ligand``."""
    import numpy as np
    from .sampling import sample_diffusion_ligand
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if on else (0, 1)
    num_batch = -(-int(num_samples) // int(batch_size))
    mine = shard_batches(num_batch, rank, world)
    out = sample_diffusion_ligand(model, shape_emb, num_samples, batch_size=batch_size, _batches=mine, _batch_seed=int(job_seed), **kw)
    pred_pos, pred_v = out[0], out[1]
    if not on or world == 1:
        return out, pred_pos, pred_v
    dev = torch.device(kw.get("device", "cuda:0"))
    counts = torch.tensor([len(x) for x in pred_v], dtype=torch.int64, device=dev)
    pos = torch.from_numpy(np.concatenate(pred_pos) if pred_pos else np.zeros((0, 3))).to(torch.float32).to(dev)   # float32 on the device: lossless
    v = torch.from_numpy(np.concatenate(pred_v) if pred_v else np.zeros((0,), np.int64)).to(torch.int64).to(dev)
    g_pos, g_v, g_counts = gather_molecules(pos, v, counts, group=group)
    cum = np.concatenate([[0], np.cumsum(g_counts.cpu().numpy())])
    g_pos, g_v = g_pos.cpu().numpy().astype(np.float64), g_v.cpu().numpy()
    return out, [g_pos[cum[k]:cum[k + 1]] for k in range(len(cum) - 1)], [g_v[cum[k]:cum[k + 1]] for k in range(len(cum) - 1)]
