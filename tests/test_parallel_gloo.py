"""world_size-2 tests of the member-per-rank sharding on CPU (gloo): the ONE collective on the data path (the
all-gather of the members' comparison images, trainer_council.py:853-856), the replicated host-RNG draws, the
logging gather, and the trainer's ownership bookkeeping.  The RCCL path on the GPU box runs the same code with
backend "nccl"."""
import copy
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import council_gan_amd as cga
from golden_util import Golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _member_image(m, b=2, h=6, w=5):
    g = torch.Generator().manual_seed(100 + m)
    return torch.randn(b, 3, h, w, generator=g).contiguous(memory_format=torch.channels_last)


def _worker(rank, world, port, council, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = cga.init_distributed("gloo")
    assert (r, w) == (rank, world)
    try:
        shard = cga.CouncilShard.from_env(council)
        per = council // world
        assert shard.local == list(range(rank * per, (rank + 1) * per))
        assert [shard.owner(m) for m in range(council)] == [m // per for m in range(council)]
        # the exchange: every rank ends up with every member's image, bit-exact, as logical NCHW
        got = shard.exchange([_member_image(m) for m in shard.local])
        assert sorted(got) == list(range(council))
        for m in range(council):
            assert got[m].shape == (2, 3, 6, 5)
            assert torch.equal(got[m].contiguous(), _member_image(m).contiguous()), (rank, m)
        # replicated host RNG: same seed -> every rank replays every member's colleague picks identically
        random.seed(1)
        picks = [cga.Council_Trainer.draw_colleagues(i, council, 4) for i in range(council)]
        # logging gather: only local entries are meaningful on a rank
        vals = [float(10 * m + 1) if m in shard.local else -1.0 for m in range(council)]
        full = shard.gather_scalars(vals)
        assert full == [float(10 * m + 1) for m in range(council)]
        # trainer bookkeeping under a shard: all members are constructed (same RNG stream on every rank), only the
        # local ones would be moved to the device
        cfg = copy.deepcopy(Golden("m2f_c3").cfg)
        cfg['council']['council_size'] = council
        torch.manual_seed(1)
        tr = cga.Council_Trainer(cfg, 'cuda:0')
        assert tr.shard.local == shard.local and len(tr.gen_a2b_s) == council
        w0 = tr.gen_a2b_s[council - 1].state_dict()['enc_content.model.0.conv.weight']
        q.put((rank, picks, float(w0.double().sum())))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("council", [2, 4])
def test_council_shard_world2_gloo(council):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, council, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == res[1][1], "colleague picks differ between ranks"
    assert res[0][2] == res[1][2], "replicated construction differs between ranks"


def test_council_must_divide_world():
    with pytest.raises(ValueError):
        cga.CouncilShard(3, rank=0, world_size=2)


def _dp_worker(rank, world, port, council, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    cga.init_distributed("gloo")
    try:
        shard = cga.CouncilShard.from_env(council)
        dp = world // council
        assert (shard.dp, shard.slice_idx, shard.local, shard.per_rank) == (dp, rank % dp, [rank // dp], 1)
        assert [shard.owner(m) for m in range(council)] == [m * dp for m in range(council)]
        # this rank's samples of a replicated batch
        batch = torch.arange(4 * 3, dtype=torch.float32).view(4, 3)
        n = 4 // dp
        assert torch.equal(shard.batch_slice(batch), batch[(rank % dp) * n:(rank % dp + 1) * n])
        # image exchange inside the slice group: every member's image for THIS rank's samples
        mine = _member_image(rank // dp, b=4)
        got = shard.exchange([shard.batch_slice(mine)])
        assert sorted(got) == list(range(council))
        for m in range(council):
            want = _member_image(m, b=4)[(rank % dp) * n:(rank % dp + 1) * n]
            assert torch.equal(got[m].contiguous(), want.contiguous()), (rank, m)
        # gradient averaging inside the member group
        g = torch.full((5,), float(rank))
        shard.replica_mean_(g)
        base = (rank // dp) * dp
        assert torch.equal(g, torch.full((5,), sum(range(base, base + dp)) / dp))
        # the bucket form trainer._gen_body uses for the decoder's share of the flat gradient (replica_mean_begin under the
        # encoder's backward, replica_mean_end before the optimizer step): a VIEW of the flat buffer is averaged in place; on
        # every transport but RCCL the mean is complete at begin and there is nothing to wait for
        flat = torch.arange(12, dtype=torch.float32) + 100.0 * rank
        h = shard.replica_mean_begin(flat[4:9])
        assert h is None
        shard.replica_mean_end(h)
        mean_rank = sum(range(base, base + dp)) / dp
        want = torch.arange(12, dtype=torch.float32) + 100.0 * rank
        want[4:9] = torch.arange(4, 9, dtype=torch.float32) + 100.0 * mean_rank
        assert torch.equal(flat, want), (rank, flat)
        # logging: a member's value is the mean of its replicas' batch-slice values
        vals = [-1.0] * council
        vals[rank // dp] = 10.0 * (rank // dp) + (rank % dp)
        full = shard.gather_scalars(vals)
        assert full == [10.0 * m + (dp - 1) / 2.0 for m in range(council)]
        q.put(rank)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_data_parallel_inside_member_world4_gloo():
    """4 ranks, 2 members: every member is replicated on 2 ranks, each takes half of the batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 4, port, 2, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [0, 1, 2, 3]


def test_shard_layouts():
    s = cga.CouncilShard(4, rank=1, world_size=2)
    assert (s.local, s.dp, s.per_rank) == ([2, 3], 1, 2)
    with pytest.raises(ValueError):
        cga.CouncilShard(4, rank=0, world_size=6)
    with pytest.raises(ValueError):
        cga.CouncilShard(4, rank=0, world_size=8)          # replicas need their process groups (from_env)
