"""N > 1 path on CPU: world_size-2 gloo run of the batch sharding and the final molecule gather
(the only collective of the job; RCCL on the GPU box)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import synth
from shapemol_amd.dist import GatherPlan, gather_molecules, shard_batches


def test_shard_batches_partition():
    for nb in (1, 2, 7, 8, 9, 64):
        for ws in (1, 2, 3, 8):
            parts = [shard_batches(nb, r, ws) for r in range(ws)]
            assert sorted(sum(parts, [])) == list(range(nb))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_batch(rank, empty_rank):
    """What rank `rank` "sampled": differently sized batches; `empty_rank` owns nothing (more ranks than batches)."""
    if rank == empty_rank:
        return np.zeros((0, 3), np.float32), np.zeros((0,), np.int64), np.zeros((0,), np.int64)
    bb = synth.synthetic_batch(3 + rank, seed=100 + rank)
    return (bb["init_pos"] + rank).astype(np.float32), bb["init_v"], bb["counts"]


def _worker(rank, world, port, q, empty_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pos, v, counts = (torch.from_numpy(a) for a in _rank_batch(rank, empty_rank))
        p, vv, c = gather_molecules(pos, v, counts)
        # the same through a plan made at "job set-up" (one collective per gather, reusable for jobs of the same sizes)
        plan = GatherPlan(len(pos), len(counts), pos.device)
        for _ in range(2):
            p2, v2, c2 = gather_molecules(pos, v, counts, plan=plan)
            assert torch.equal(p2, p) and torch.equal(v2, vv) and torch.equal(c2, c)
        if len(pos) > 0:
            try:
                gather_molecules(pos[:-1], v[:-1], counts, plan=plan)
                raise AssertionError("a plan for other sizes must be refused")
            except ValueError:
                pass
        q.put((rank, p.numpy(), vv.numpy(), c.numpy()))
    finally:
        dist.destroy_process_group()


def _run_gather(world, empty_rank=-1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, empty_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = [_rank_batch(r, empty_rank) for r in range(world)]
    exp_p, exp_v, exp_c = (np.concatenate([pt[i] for pt in parts]) for i in range(3))
    assert sorted(r for r, *_ in got) == list(range(world))
    for rank, p, v, c in got:
        assert p.dtype == np.float32 and v.dtype == np.int64 and c.dtype == np.int64
        assert np.array_equal(p, exp_p) and np.array_equal(v, exp_v) and np.array_equal(c, exp_c)
    assert exp_c.sum() == len(exp_v)


def test_gather_molecules_gloo_world2():
    _run_gather(2)


def test_gather_molecules_gloo_world3_with_empty_rank():
    """Three ranks, the middle one owns no batch at all (more ranks than batches): ragged padding down to zero rows."""
    _run_gather(3, empty_rank=1)


def test_gather_is_identity_without_process_group():
    pos, v, c = torch.zeros(5, 3), torch.zeros(5, dtype=torch.long), torch.tensor([2, 3])
    p2, v2, c2 = gather_molecules(pos, v, c)
    assert p2 is pos and v2 is v and c2 is c


def _single_rank_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        pos, v, counts = (torch.from_numpy(a) for a in _rank_batch(0, -1))
        p0, v0, c0 = gather_molecules(pos, v, counts)                               # shortcut: the same objects
        p1, v1, c1 = gather_molecules(pos, v, counts, _single_rank_too=True)        # the packed collective path
        q.put((p0 is pos, torch.equal(p1, pos) and torch.equal(v1, v) and torch.equal(c1, counts) and p1 is not pos,
               str(p1.dtype), str(v1.dtype), str(c1.dtype)))
    finally:
        dist.destroy_process_group()


def test_gather_single_rank_forced_through_the_collective():
    """A 1-rank group pushed through packing / all_gather_into_tensor / unpacking (what tests/test_gpu_parity.py does with the
    nccl backend on one MI355X, and `bench.py --force-collective`)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(_free_port(), q))
    p.start()
    shortcut, same, dp, dv, dc = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert shortcut and same and (dp, dv, dc) == ("torch.float32", "torch.int64", "torch.int64")
