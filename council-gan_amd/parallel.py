"""Member-per-GPU sharding of the council -- new relative to the reference, which
is single-process.  Council members are independent models; the only per-iteration cross-member
data dependency is that each member's council discriminator compares its own image with the
OTHER members' generated images (trainer_council.py:853-856, 872-874).  So:

  * rank r owns members [r*L, (r+1)*L), L = council_size / world_size (their generator, both
    discriminators, three optimizers, loss-history rings, checkpoint files);
  * the batch, the style noise and the Python/NumPy RNG streams are replicated (same seeds on
    every rank; every rank replays every member's `random.choice` draws);
  * ONE collective on the data path: an all-gather of the local members' comparison images
    ([L, B, H, W, C] fp32 per rank; 3.1 MB at B=4, 256x256) -- RCCL over xGMI on the GPU box
    (`backend="nccl"`), gloo in the CPU tests.  Latency-bound, never bandwidth-bound.

This module has no dependency on the HIP library: the exchange works on any torch tensor, which
is what lets the world_size-2 gloo tests exercise it on CPU."""
import os

import torch
import torch.distributed as dist


class CouncilShard:
    def __init__(self, council_size, rank=0, world_size=1, group=None):
        if council_size % world_size != 0:
            raise ValueError("council_size %d must be a multiple of the number of ranks %d "
                             "(intra-member data parallelism is a later row, SURVEY.md 8f.4)" % (council_size, world_size))
        self.council_size = council_size
        self.rank = rank
        self.world_size = world_size
        self.group = group
        self.per_rank = council_size // world_size
        self.local = list(range(rank * self.per_rank, (rank + 1) * self.per_rank))

    @classmethod
    def from_env(cls, council_size):
        """Single process unless torch.distributed has been initialised by the launcher."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return cls(council_size, dist.get_rank(), dist.get_world_size())
        return cls(council_size)

    def owner(self, member):
        return member // self.per_rank

    def exchange(self, local_images):
        """local_images: list (len = per_rank) of logical-NCHW tensors (channels_last or contiguous).
        Returns {member id: image} for EVERY member.  world_size == 1: no copy, no collective."""
        if len(local_images) != self.per_rank:
            raise ValueError("expected %d local images" % self.per_rank)
        if self.world_size == 1:
            return {m: t for m, t in zip(self.local, local_images)}
        # stack the PHYSICAL (NHWC) layout so nothing is re-ordered before / after the collective
        send = torch.stack([t.permute(0, 2, 3, 1).contiguous() for t in local_images], 0)
        recv = torch.empty((self.world_size,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        out = {}
        for r in range(self.world_size):
            for k in range(self.per_rank):
                m = r * self.per_rank + k
                out[m] = local_images[k] if r == self.rank else recv[r, k].permute(0, 3, 1, 2)
        return out

    def gather_scalars(self, values):
        """values: list of council_size floats with only the local entries meaningful -> full list
        (logging only; off the hot path)."""
        if self.world_size == 1:
            return list(values)
        t = torch.zeros(self.council_size, dtype=torch.float64)
        for m in self.local:
            t[m] = float(values[m])
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=self.group)
        return t.cpu().tolist()


def init_distributed(backend=None):
    """Launcher contract of bench.py / train: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR/PORT in the env."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank
