"""N > 1 path on CPU: world_size-2 gloo run of the batch sharding and the final molecule gather
(the only collective of the job; RCCL on the GPU box)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import synth
from shapemol_amd.dist import gather_molecules, shard_batches


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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank "samples" a different, differently sized batch (rank 1 owns one molecule more)
        bb = synth.synthetic_batch(3 + rank, seed=100 + rank)
        pos = torch.from_numpy(bb["init_pos"]) + rank
        v = torch.from_numpy(bb["init_v"])
        counts = torch.from_numpy(bb["counts"])
        p, vv, c = gather_molecules(pos, v, counts)
        q.put((rank, p.numpy(), vv.numpy(), c.numpy()))
    finally:
        dist.destroy_process_group()


def test_gather_molecules_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_p, exp_v, exp_c = [], [], []
    for r in range(2):
        bb = synth.synthetic_batch(3 + r, seed=100 + r)
        exp_p.append(bb["init_pos"] + r); exp_v.append(bb["init_v"]); exp_c.append(bb["counts"])
    exp_p, exp_v, exp_c = np.concatenate(exp_p), np.concatenate(exp_v), np.concatenate(exp_c)
    for rank, p, v, c in got:
        assert np.array_equal(p, exp_p.astype(np.float32)) and np.array_equal(v, exp_v) and np.array_equal(c, exp_c)
    assert exp_c.sum() == len(exp_v)


def test_gather_is_identity_without_process_group():
    pos, v, c = torch.zeros(5, 3), torch.zeros(5, dtype=torch.long), torch.tensor([2, 3])
    p2, v2, c2 = gather_molecules(pos, v, c)
    assert p2 is pos and v2 is v and c2 is c
