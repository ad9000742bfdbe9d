"""Flat-buffer Adam: torch.optim.Adam semantics (trainer_council.py:170-179 -- lr, betas,
L2 weight_decay, eps 1e-8, no amsgrad) executed by ONE HIP kernel per contiguous run of
parameters that received a gradient.

All parameters of one optimizer live in one flat fp32 buffer (data / grad / exp_avg / exp_avg_sq);
`p.data` and `p.grad` become views into it (conv weights keep their logical OIHW shape with
channels_last strides, so state_dict shapes are unchanged).  Backward kernels accumulate straight
into the grad views (ops.py), `zero_grad()` is a single fill, `step()` a single kernel.

torch.optim.Adam skips parameters whose `.grad is None` (no decay, no step count): that is how
the reference leaves `enc_style` untouched in gen_update.  Here a parameter counts
as "without gradient" until a backward kernel has written to its grad view since the last
zero_grad (the `_cg_touched` flag set in ops.py)."""
import torch

from . import hip
from .hip import check, ptr, stream


def _phys_view(flat, off, shape):
    """View of flat[off: off+n] with logical `shape`; 4-D tensors get channels_last strides."""
    n = 1
    for s in shape:
        n *= s
    seg = flat[off:off + n]
    if len(shape) == 4:
        o, i, kh, kw = shape
        return seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
    return seg.view(shape)


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = [p for p in params if p.requires_grad]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False)
        super().__init__(params, defaults)
        self._params = params
        self._flat = None
        self._steps = [0] * len(params)
        self.version = 0          # bumped whenever the parameter values may have changed (step / load / move)

    # -- flat storage ---------------------------------------------------------------------
    def layout(self):
        """(sizes, start offsets, padded total) of the parameters inside this optimizer's flat buffers."""
        sizes = [p.numel() for p in self._params]
        # every parameter starts on a 32-ELEMENT boundary of the flat buffers: the fp16 {hi, lo} mirror of the buffer
        # (ops.SplitWeights) interleaves the halves per 32 elements of the flat index, so a tensor's groups must not
        # straddle its start (and its 128-byte lines are aligned); padding stays zero for ever
        starts, total = [], 0
        for n in sizes:
            starts.append(total)
            total = (total + n + 31) & ~31
        return sizes, starts, total

    def materialize(self, device, storage=None):
        """Move every parameter into the flat buffer on `device` (idempotent per device).  `storage`: (data, grad, m, v)
        slices of a ParamPool that become this optimizer's buffers instead of private allocations."""
        device = torch.device(device)
        if self._flat is not None and self._flat["data"].device == device and storage is None:
            return
        sizes, starts, total = self.layout()
        old = self._flat
        if storage is not None:
            data, grad, m, v = storage
            if min(t.numel() for t in storage) < total:
                raise ValueError("pool slice smaller than the optimizer's parameters")
            for t in storage:
                t.zero_()
        else:
            data = torch.zeros(total, dtype=torch.float32, device=device)
            grad = torch.zeros(total, dtype=torch.float32, device=device)
            m = torch.zeros(total, dtype=torch.float32, device=device)
            v = torch.zeros(total, dtype=torch.float32, device=device)
        if old is not None:          # moments survive a move (host -> device, private -> pool); the values come from p.data
            n_old = old["m"].numel()
            m[:n_old].copy_(old["m"])
            v[:n_old].copy_(old["v"])
        offs = []
        for p, n, off in zip(self._params, sizes, starts):
            view = _phys_view(data, off, tuple(p.shape))
            view.copy_(p.data)
            p.data = view
            g = _phys_view(grad, off, tuple(p.shape))
            g._cg_touched = False
            p._cg_grad = g
            p.grad = g
            offs.append(off)
        self._flat = dict(data=data, grad=grad, m=m, v=v, offs=offs, sizes=sizes)
        self.version += 1

    @property
    def flat(self):
        if self._flat is None:
            raise RuntimeError("FlatAdam.materialize(device) has not been called")
        return self._flat

    # -- torch.optim API -------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        f = self.flat
        if f["grad"].is_cuda:
            check(hip.load().cg_fill(ptr(f["grad"]), f["grad"].numel(), 0.0, stream()), "cg_fill")
        else:
            f["grad"].zero_()
        for p in self._params:
            p._cg_grad._cg_touched = False
            p.grad = p._cg_grad

    def touched_runs(self):
        """Maximal runs [i0, i1) of consecutive parameters that received a gradient and share a step count."""
        runs, i, n = [], 0, len(self._params)
        while i < n:
            if not self._params[i]._cg_grad._cg_touched:
                i += 1
                continue
            j = i
            while j + 1 < n and self._params[j + 1]._cg_grad._cg_touched and self._steps[j + 1] == self._steps[i]:
                j += 1
            runs.append((i, j + 1))
            i = j + 1
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        f = self.flat
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        lib = hip.load()
        self.version += 1
        for i0, i1 in self.touched_runs():
            off = f["offs"][i0]
            n = f["offs"][i1 - 1] + f["sizes"][i1 - 1] - off
            step = self._steps[i0] + 1
            sl = slice(off, off + n)
            check(lib.cg_adam_step(ptr(f["data"][sl]), ptr(f["grad"][sl]), ptr(f["m"][sl]), ptr(f["v"][sl]), n,
                                   float(grp["lr"]), float(b1), float(b2), float(grp["eps"]),
                                   float(grp["weight_decay"]), step, stream()), "cg_adam_step")
            for k in range(i0, i1):
                self._steps[k] = step

    # -- checkpoint compatibility with torch.optim.Adam (trainer_council.py:989-992) --------------
    def state_dict(self):
        f = self.flat
        state = {}
        for i, p in enumerate(self._params):
            if self._steps[i] == 0:
                continue
            off = f["offs"][i]
            state[i] = {"step": torch.tensor(float(self._steps[i])),
                        "exp_avg": _phys_view(f["m"], off, tuple(p.shape)).clone(memory_format=torch.contiguous_format),
                        "exp_avg_sq": _phys_view(f["v"], off, tuple(p.shape)).clone(memory_format=torch.contiguous_format)}
        groups = []
        for g in self.param_groups:
            gg = {k: v for k, v in g.items() if k != "params"}
            gg["params"] = list(range(len(self._params)))
            groups.append(gg)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        f = self.flat
        self.version += 1
        for i, p in enumerate(self._params):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            off = f["offs"][i]
            if st is None:
                self._steps[i] = 0
                continue
            self._steps[i] = int(float(st["step"]))
            _phys_view(f["m"], off, tuple(p.shape)).copy_(st["exp_avg"])
            _phys_view(f["v"], off, tuple(p.shape)).copy_(st["exp_avg_sq"])
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v


class ParamPool:
    """The same optimizer kind (generator / discriminator / council discriminator) of SEVERAL council members in ONE
    flat storage: member k's FlatAdam buffers are slice k of the pool's data / grad / exp_avg / exp_avg_sq tensors, all
    slices `stride` elements long (a multiple of 32).  The members have identical architectures, so a parameter sits at the
    same offset in every slice -- which is what lets ONE kernel launch serve the same layer of all members (cg_group in
    include/council_gan_hip.h: member z's weights = member 0's pointer + z * stride) and one Adam / zero-fill / fp16-split
    launch serve all members.  The per-member FlatAdam objects stay the public optimizers (state_dict, param_groups,
    schedulers, checkpoints: trainer_council.py:139-183, 969-992)."""

    def __init__(self, opts):
        self.opts = list(opts)
        lay = [o.layout() for o in self.opts]
        if any(l[0] != lay[0][0] for l in lay):
            raise ValueError("council members must have identical parameter layouts to share a pool")
        self.stride = lay[0][2]
        self.data = self.grad = self.m = self.v = None
        self.split = None            # ops.SplitWeights of the pool (set by the trainer when the split datapath is on)
        for k, o in enumerate(self.opts):
            o._pool, o._pool_index = self, k

    def materialize(self, device):
        device = torch.device(device)
        if self.data is not None and self.data.device == device:
            return
        n = len(self.opts) * self.stride
        self.data, self.grad, self.m, self.v = (torch.zeros(n, dtype=torch.float32, device=device) for _ in range(4))
        for k, o in enumerate(self.opts):
            sl = slice(k * self.stride, (k + 1) * self.stride)
            o.materialize(device, storage=(self.data[sl], self.grad[sl], self.m[sl], self.v[sl]))
            for p in o._params:
                p._cg_pool = self

    @property
    def version(self):
        return sum(o.version for o in self.opts)

    def index_of(self, param):
        """(member index in the pool, parameter index) of a parameter tensor, or None."""
        key = id(param)
        tab = self.__dict__.get('_index')
        if tab is None:
            tab = {id(p): (k, i) for k, o in enumerate(self.opts) for i, p in enumerate(o._params)}
            self.__dict__['_index'] = tab
        return tab.get(key)

    def zero_grad(self, k0=0, n=None):
        """Zero the gradient slices of members [k0, k0+n) with one fill."""
        n = len(self.opts) - k0 if n is None else n
        g = self.grad[k0 * self.stride:(k0 + n) * self.stride]
        if g.is_cuda:
            check(hip.load().cg_fill(ptr(g), g.numel(), 0.0, stream()), "cg_fill")
        else:
            g.zero_()
        for o in self.opts[k0:k0 + n]:
            for p in o._params:
                p._cg_grad._cg_touched = False
                p.grad = p._cg_grad

    def _same(self, opts, lockstep):
        """Can members `opts` take ONE launch per run of touched parameters?  (identical step counts and hyper-parameters; a
        member-batched launch writes every member's slice but flags only the lead's gradient views -- a member whose own
        flags show MORE than the lead's was touched outside a batched launch and steps on its own)"""
        lead = opts[0]
        if len(opts) == 1:
            return True
        hyper = lambda o: tuple((k, tuple(v) if isinstance(v, (list, tuple)) else v)      # noqa: E731
                                for k, v in sorted(o.param_groups[0].items()) if k not in ("params", "initial_lr"))
        same = lockstep and all(o._steps == lead._steps and hyper(o) == hyper(lead) for o in opts[1:])
        if lockstep:
            for o in opts[1:]:
                for p, q in zip(o._params, lead._params):
                    if p._cg_grad._cg_touched and not q._cg_grad._cg_touched:
                        same = False
                    p._cg_grad._cg_touched = p._cg_grad._cg_touched or q._cg_grad._cg_touched
        return same

    def plan_hyper(self, k0, n):
        """The per-run {step_size, sqrt(bias_correction2)} pairs the NEXT step of members [k0, k0+n) will use, as a host
        tensor [runs, 2] (cg_adam_hyper: the very floats cg_adam_step_g computes) -- or None while the runs of that step
        are not known yet (before its first execution).  Staged into device memory by the trainer (graphs.HostInputs) so
        that the step's launches can be captured in a hipGraph."""
        runs = self.__dict__.setdefault('_runs', {}).get((k0, n))
        if runs is None:
            return None
        import ctypes
        lead = self.opts[k0]
        grp = lead.param_groups[0]
        b1, b2 = grp["betas"]
        out = torch.empty(len(runs), 2, dtype=torch.float32)
        buf = (ctypes.c_float * 2)()
        lib = hip.load()
        for r, (i0, _i1) in enumerate(runs):
            check(lib.cg_adam_hyper(float(grp["lr"]), float(b1), float(b2), lead._steps[i0] + 1, buf), "cg_adam_hyper")
            out[r, 0], out[r, 1] = buf[0], buf[1]
        return out

    def advance(self, k0, n, runs):
        """Host bookkeeping of one executed step of members [k0, k0+n): step counts of the touched runs, weight versions."""
        for o in self.opts[k0:k0 + n]:
            for i0, i1 in runs:
                for k in range(i0, i1):
                    o._steps[k] += 1
            o.version += 1

    @torch.no_grad()
    def step(self, k0=0, n=None, lockstep=False, hyper=None, advance=True):
        """Adam step of members [k0, k0+n).  `lockstep`: the members were updated by member-batched launches that only
        flagged the FIRST member's gradient views as written -- its pattern holds for all of them; one launch per run of
        touched parameters serves every member (their slices are `stride` apart).  `hyper`: device tensor [runs, 2] from
        plan_hyper() -- the per-step scalars then come from device memory (same floats, bit-identical update) and the
        launches are valid under hipGraph replay.  `advance=False`: launch only; the caller applies advance() itself (once
        per execution of a captured step).  Returns the runs it launched (None when the members stepped one by one)."""
        n = len(self.opts) - k0 if n is None else n
        opts = self.opts[k0:k0 + n]
        lead = opts[0]
        if not self._same(opts, lockstep):
            if hyper is not None:
                raise hip.HipError("members with different step counts / hyper-parameters cannot take a graph-captured step")
            for o in opts:
                o.step()
            self.__dict__.setdefault('_runs', {})[(k0, n)] = None
            return None
        f = lead.flat
        grp = lead.param_groups[0]
        b1, b2 = grp["betas"]
        lib = hip.load()
        runs = lead.touched_runs()
        if hyper is not None and tuple(hyper.shape) != (len(runs), 2):
            raise hip.HipError("staged Adam scalars do not match the step's parameter runs")
        for r, (i0, i1) in enumerate(runs):
            off = f["offs"][i0]
            cnt = f["offs"][i1 - 1] + f["sizes"][i1 - 1] - off
            base = k0 * self.stride + off
            sl = slice(base, base + cnt)
            if hyper is not None:
                check(lib.cg_adam_step_dev(ptr(self.data[sl]), ptr(self.grad[sl]), ptr(self.m[sl]), ptr(self.v[sl]), cnt, n,
                                           self.stride, float(b1), float(b2), float(grp["eps"]), float(grp["weight_decay"]),
                                           ptr(hyper[r]), stream()), "cg_adam_step_dev")
            else:
                check(lib.cg_adam_step_g(ptr(self.data[sl]), ptr(self.grad[sl]), ptr(self.m[sl]), ptr(self.v[sl]), cnt, n,
                                         self.stride, float(grp["lr"]), float(b1), float(b2), float(grp["eps"]),
                                         float(grp["weight_decay"]), lead._steps[i0] + 1, stream()), "cg_adam_step_g")
        self.__dict__.setdefault('_runs', {})[(k0, n)] = runs
        if advance:
            self.advance(k0, n, runs)
        return runs
