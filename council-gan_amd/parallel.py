"""Sharding of the council over the GPUs of a node -- new relative to the reference, which
is single-process.  Council members are independent models; the only per-iteration cross-member
data dependency is that each member's council discriminator compares its own image with the
OTHER members' generated images (trainer_council.py:853-856, 872-874).  So:

  world_size <= council_size  (member sharding)
  * rank r owns members [r*L, (r+1)*L), L = council_size / world_size (their generator, both
    discriminators, three optimizers, loss-history rings, checkpoint files);
  * the batch, the style noise and the Python/NumPy RNG streams are replicated (same seeds on
    every rank; every rank replays every member's `random.choice` draws);
  * ONE collective on the data path: an all-gather of the local members' comparison images
    ([L, B, H, W, C] fp32 per rank; 3.1 MB at B=4, 256x256) -- RCCL over xGMI on the GPU box
    (`backend="nccl"`), gloo in the CPU tests.  Latency-bound, never bandwidth-bound.

  world_size = D * council_size  (one member per D ranks: data parallelism INSIDE a member)
  * rank r holds a full replica of member r // D and works on samples [s*B/D, (s+1)*B/D) of every batch,
    s = r % D; the networks have no cross-sample operator (instance / adaptive-instance norm are per sample), so
    a batch slice is exact;
  * the image exchange runs inside the SLICE group (the ranks with the same s: one per member);
  * gradients are averaged inside the MEMBER group (the D replicas) with one all-reduce of the optimizer's flat
    gradient buffer per optimizer step -- every loss is a batch mean, so the average of the replicas' gradients
    is the full-batch gradient; the two statistics that are not linear in the batch (the squared mask mean of
    the focus loss, the loss-matching history) are averaged before they are used (Council_Trainer).

By default the collectives go through torch.distributed (backend "nccl" = RCCL on the GPU box), which works on any torch
tensor and is what lets the gloo tests exercise them on CPU; with the gloo backend device tensors are staged through the
host (gloo is the test transport; RCCL takes device pointers).  With CG_NATIVE_COLLECTIVES=1 the two data-path
collectives use the C-ABI instead (include/council_gan_hip.h: cg_comm_*, cg_allgather_images, cg_allreduce_sum --
communicators of this library's own, enqueued on the CURRENT stream, so they are ordered with the kernels around them
without an event hand-off); torch.distributed is then only the side channel that distributes the communicator ids."""
import os

import torch
import torch.distributed as dist


def _is_nccl(group):
    return dist.get_backend(group) == "nccl"


def _all_gather(recv, send, group, comm=None):
    """recv[g] = rank g's send (flat views); device tensors go through the host unless the backend is RCCL."""
    if comm is not None:
        comm.all_gather(recv.view(-1), send.view(-1))
    elif send.is_cuda and not _is_nccl(group):
        r, s = torch.empty(recv.shape, dtype=recv.dtype), send.cpu()
        dist.all_gather_into_tensor(r.view(-1), s.view(-1), group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)


def _all_reduce_sum(t, group, comm=None):
    if comm is not None:
        comm.all_reduce_sum_(t)
    elif t.is_cuda and not _is_nccl(group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


def native_comm(group, ranks):
    """A C-ABI communicator (hip.Comm) over `ranks` (global ranks, in group order) of the torch process group `group`:
    the group's first rank draws the id, torch.distributed carries it to the others.  Collective over the group."""
    from . import hip
    me = dist.get_rank()
    buf = torch.zeros(hip.COMM_ID_BYTES, dtype=torch.uint8)
    if me == ranks[0]:
        buf = torch.tensor(list(hip.comm_unique_id()), dtype=torch.uint8)
    if _is_nccl(group):
        buf = buf.cuda()
    dist.broadcast(buf, src=ranks[0], group=group)
    return hip.Comm(bytes(buf.cpu().tolist()), ranks.index(me), len(ranks))


class CouncilShard:
    def __init__(self, council_size, rank=0, world_size=1, group=None, member_group=None, slice_group=None):
        self.member_comm = self.slice_comm = None     # C-ABI communicators (use_native_collectives)
        self.timing = None          # a list: exchange_flat appends a (start, end) event pair per exchange (bench.py N > 1)
        self.council_size = council_size
        self.rank = rank
        self.world_size = world_size
        self.group = group
        if world_size <= council_size:
            if council_size % world_size != 0:
                raise ValueError("council_size %d must be a multiple of the number of ranks %d (or the number of "
                                 "ranks a multiple of council_size)" % (council_size, world_size))
            self.dp, self.slice_idx = 1, 0
            self.per_rank = council_size // world_size
            self.local = list(range(rank * self.per_rank, (rank + 1) * self.per_rank))
            self.member_group, self.slice_group = None, group
            self.slice_ranks = world_size
        else:
            if world_size % council_size != 0:
                raise ValueError("the number of ranks %d must be a multiple of council_size %d (or divide it)"
                                 % (world_size, council_size))
            self.dp = world_size // council_size
            self.slice_idx = rank % self.dp
            self.per_rank = 1
            self.local = [rank // self.dp]
            if member_group is None or slice_group is None:
                raise ValueError("data parallelism inside a member needs the member and slice process groups "
                                 "(CouncilShard.from_env creates them)")
            self.member_group, self.slice_group = member_group, slice_group
            self.slice_ranks = council_size

    @classmethod
    def from_env(cls, council_size):
        """Single process unless torch.distributed has been initialised by the launcher."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return cls(council_size)
        rank, world = dist.get_rank(), dist.get_world_size()
        if world <= council_size or world % council_size != 0:
            return cls(council_size, rank, world)
        dp = world // council_size
        member_group = slice_group = None
        for m in range(council_size):             # every rank creates every group, in the same order
            g = dist.new_group([m * dp + s for s in range(dp)])
            if rank // dp == m:
                member_group = g
        for s in range(dp):
            g = dist.new_group([m * dp + s for m in range(council_size)])
            if rank % dp == s:
                slice_group = g
        return cls(council_size, rank, world, None, member_group, slice_group)

    def use_native_collectives(self):
        """Route the image exchange and the replica gradient average through the C-ABI communicators (collective: every
        rank calls it, after its HIP device has been selected).  No-op for a single process."""
        if self.world_size == 1:
            return self
        if self.dp == 1:
            self.slice_comm = native_comm(self.slice_group, list(range(self.world_size)))
        else:
            m, s = self.rank // self.dp, self.slice_idx
            self.member_comm = native_comm(self.member_group, [m * self.dp + k for k in range(self.dp)])
            self.slice_comm = native_comm(self.slice_group, [k * self.dp + s for k in range(self.council_size)])
        return self

    def owner(self, member):
        """First rank that holds `member`."""
        return member * self.dp if self.dp > 1 else member // self.per_rank

    # ---- batch slicing / replica averaging (no-ops unless a member spans several ranks) ------------------------
    def batch_slice(self, t):
        if self.dp == 1:
            return t
        b = t.shape[0]
        if b % self.dp != 0:
            raise ValueError("batch size %d is not a multiple of the %d ranks per council member" % (b, self.dp))
        n = b // self.dp
        return t[self.slice_idx * n:(self.slice_idx + 1) * n]

    def replica_mean_(self, t):
        """In-place mean over the replicas of this rank's member (gradients, the focus-loss sums, logged losses)."""
        if self.dp > 1:
            _all_reduce_sum(t, self.member_group, self.member_comm)
            t.mul_(1.0 / self.dp)
        return t

    def replica_mean_begin(self, t):
        """Start the in-place replica mean of `t` WITHOUT making the current stream wait for it (RCCL through torch.distributed:
        an asynchronous all-reduce ordered behind what the current stream has queued so far); every other transport does the
        whole mean now.  Returns a handle for replica_mean_end, or None when there is nothing left to wait for."""
        if self.dp == 1:
            return None
        if self.member_comm is None and t.is_cuda and _is_nccl(self.member_group):
            return (dist.all_reduce(t, group=self.member_group, async_op=True), t)
        self.replica_mean_(t)
        return None

    def replica_mean_end(self, handle):
        if handle is not None:
            handle[0].wait()                       # the current stream continues behind the collective
            handle[1].mul_(1.0 / self.dp)

    # ---- the image exchange -------------------------------------------------------------------------------------
    def exchange(self, local_images):
        """local_images: list (len = per_rank) of logical-NCHW tensors (channels_last or contiguous).
        Returns {member id: image} for EVERY member (of this rank's batch slice).  One rank: no copy, no collective."""
        if len(local_images) != self.per_rank:
            raise ValueError("expected %d local images" % self.per_rank)
        if self.world_size == 1:
            return {m: t for m, t in zip(self.local, local_images)}
        # stack the PHYSICAL (NHWC) layout so nothing is re-ordered before / after the collective
        send = torch.stack([t.permute(0, 2, 3, 1).contiguous() for t in local_images], 0)
        recv = torch.empty((self.slice_ranks,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        _all_gather(recv, send, self.slice_group, self.slice_comm)
        me = self.local[0] // self.per_rank
        out = {}
        for r in range(self.slice_ranks):
            for k in range(self.per_rank):
                m = r * self.per_rank + k
                out[m] = local_images[k] if r == me else recv[r, k].permute(0, 3, 1, 2)
        return out

    def exchange_flat(self, local):
        """`local`: this rank's members' comparison images as ONE member-major tensor [per_rank * B, C, H, W] (logical
        NCHW, channels_last).  Returns every member's images as one member-major tensor [council_size * B, C, H, W] (of this
        rank's batch slice): block m = member m.  One rank: the input itself, no copy, no collective."""
        if self.world_size == 1:
            return local
        send = local.permute(0, 2, 3, 1).contiguous()                 # the PHYSICAL (NHWC) layout, untouched
        recv = torch.empty((self.slice_ranks,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        ev = None
        if self.timing is not None and send.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        _all_gather(recv, send, self.slice_group, self.slice_comm)
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)
        return recv.view((-1,) + tuple(send.shape[1:])).permute(0, 3, 1, 2)

    def gather_scalars(self, values):
        """values: list of council_size floats with only the local entries meaningful -> full list
        (logging only; off the hot path).  Replicas of a member contribute their batch-slice values' mean."""
        if self.world_size == 1:
            return list(values)
        t = torch.zeros(self.council_size, dtype=torch.float64)
        for m in self.local:
            t[m] = float(values[m]) / self.dp
        if _is_nccl(self.group):
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
