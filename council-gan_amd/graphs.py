"""Host inputs and hipGraph segments of the training step (new relative to the reference, which enqueues every kernel from
Python every iteration: trainer_council.py:328,558,747,826,858).

An update of the Council-GAN step launches 150-300 kernels whose SHAPES never change from one iteration to the next; what
changes is a handful of host-side values: the style noise (CPU RNG, trainer_council.py:284-285,741,744,807-809), the
colleague picks (Python RNG, :861-868), Adam's bias corrections, the loss-matching ring positions.  `HostInputs` gives every
such value a STATIC device buffer that the host refreshes each call (pinned staging, asynchronous copy on the stream the
update runs on); everything downstream reads the buffer.  With that, the device work of an update is a fixed kernel
sequence, and `Segment` captures it once into a hipGraph (torch.cuda.graph = hipStreamBeginCapture on ROCm) and replays it:
host cost per update ~0.1 ms instead of ~10 ms of Python -- the difference between an enqueue-bound and a GPU-bound rank
once the council is sharded one member per GPU."""
import torch


class HostInputs:
    """name -> static device tensor, refreshed from a host tensor on every stage() (outside any graph capture)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.generation = 0            # bumped when an existing static buffer is re-allocated: captured graphs hold its old address
        self._dev = {}
        self._ring = [None] * 48       # pinned staging buffers, each guarded by the event of its last copy
        self._i = 0

    def stage(self, name, t):
        t = t.detach()
        if t.is_cuda:
            raise ValueError("HostInputs.stage takes host tensors")
        t = t.contiguous()
        d = self._dev.get(name)
        if d is None or d.shape != t.shape or d.dtype != t.dtype:
            if d is not None:
                self.generation += 1       # a buffer some captured graph may read has moved (a NEW name is in no graph yet)
            d = self._dev[name] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return d
        k = self._i % len(self._ring)
        self._i += 1
        slot = self._ring[k]
        if slot is None or slot[0].numel() < nbytes:
            slot = [torch.empty(max(nbytes, 4096), dtype=torch.uint8, pin_memory=True), None]
            self._ring[k] = slot
        if slot[1] is not None:
            slot[1].synchronize()      # the copy that last used this staging buffer has completed (48 uploads ago)
        stage = slot[0][:nbytes].view(t.dtype).view(t.shape)
        stage.copy_(t)
        d.copy_(stage, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        return d

    def get(self, name):
        return self._dev[name]


class Segment:
    """One captured piece of an update: the graph, what its body returned (tensors in the graph's private pool, valid after
    every replay) and the host-side effects (optimizer step counts, ring positions, version counters) to repeat after each
    replay."""

    def __init__(self):
        self.graph = None
        self.warm = 0
        self.out = None
        self.effects = []
        self.generation = -1
        self.ws = {}              # scratch buffers of the captured launches (hip.capture_workspaces): live as long as the graph
        self.serial = 0           # capture number (trainer-wide) of the current graph
        self.parents = {}         # kind -> serial of the earlier segments of the iteration this graph was captured behind
        self.used = 0             # trainer-wide use clock (least recently used segment of a kind is evicted first)
