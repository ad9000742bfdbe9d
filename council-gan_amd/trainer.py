"""Council_Trainer on the gfx950 kernels -- drop-in for `/root/reference/trainer_council.py`.

Same constructor, same `dis_update / dis_council_update / gen_update / sample / save / resume /
update_learning_rate` signatures, same member lists (`gen_a2b_s[i]` ...), same reflected logging
attributes, same host-RNG draws in the same order, so the reference `train.py`
runs unchanged on top of it.

What is restructured (results unchanged, SURVEY.md 8d "redundancy the build may remove"):
  * the generator runs without an autograd tape in the two discriminator updates (the reference
    builds the graph and detaches, trainer_council.py:760-769, 837-872);
  * the style encoder is skipped (its output is discarded on every shipped config);
  * fake + real go through D as one batch; in the council-D update the own image and the DISTINCT
    colleagues' images go through once with per-sample loss weights (the reference repeats the
    identical fake pass per pick, :862-874);
  * D / council-D weight gradients are not computed in gen_update (the reference accumulates and
    then zeroes them);
  * loss matching (:518-524, 576-586) runs on device rings: no host sync inside the step;
  * the members of a rank run MEMBER-BATCHED: their samples are stacked along the batch, their parameters are slices of
    one pool per optimizer kind (optim.ParamPool), and every layer / loss / Adam step of the group is one launch
    (`ops.members`; the reference's sequential member loops, :328,558,747,826,858).  CG_GROUP caps the members per
    launch (default 4; 1 = member by member);
  * the discriminator and the council-discriminator update of an iteration touch disjoint networks: they run on two
    side streams behind a shared prologue (CG_OVERLAP_UPDATES=0 turns that off).

Multi-GPU: one process per GPU, council members sharded across ranks (member m lives on rank
m // members_per_rank); everything host-side (batch, style noise, Python RNG picks) is replicated;
the ONLY collective on the data path is one all-gather of the generated images in
`dis_council_update` (the cross-member dependency at trainer_council.py:853-856, 872-874).  With more ranks than
members every member is replicated, each replica takes a slice of the batch and the replicas' flat gradient buffers are
averaged with one all-reduce per optimizer step (parallel.py).
"""
import contextlib
import os
import random
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import hip, ops
from .hip import check, ptr, stream
from .networks import AdaINGen, MsImageDis, MsImageDisCouncil
from .optim import FlatAdam, ParamPool
from .parallel import CouncilShard
from .utils import get_model_list, get_scheduler, weights_init


def _check_supported(hp):
    for k in ('recon_x_w', 'recon_s_w', 'recon_c_w', 'recon_x_cyc_w', 'vgg_w', 'abs_beginning_end', 'council_abs_w'):
        if hp.get(k, 0) != 0:
            raise NotImplementedError("%s != 0 is outside the shipped-config hot path (SURVEY.md 8a R2)" % k)
    if hp['gen']['useRandomDis'] or hp['dis']['useRandomGen']:
        raise NotImplementedError("useRandomDis / useRandomGen break member sharding and are False in every "
                                  "shipped config (SURVEY.md 8e)")
    if hp['dis']['do_Dis_only_gray']:
        raise NotImplementedError("dis.do_Dis_only_gray is False in every shipped config")
    if hp['focus_loss']['do_w_loss_matching_focus']:
        raise NotImplementedError("focus_loss.do_w_loss_matching_focus is False in every shipped config")


class Council_Trainer(nn.Module):
    def __init__(self, hyperparameters, cuda_device='cuda:0', shard=None):
        super().__init__()
        hp = hyperparameters
        _check_supported(hp)
        lr = hp['lr']
        self.council_size = hp['council']['council_size']
        self.council_size_conf = self.council_size
        self.do_dis_council = hp['council_w'] != 0
        self.do_ads_council_loss = hp['council_abs_w'] != 0
        self.numberOfCouncil_dis_relative_iteration_conf = hp['council']['numberOfCouncil_dis_relative_iteration']
        self.discriminetro_less_style_by_conf = hp['council']['discriminetro_less_style_by']
        self.cuda_device = cuda_device
        self.shard = shard if shard is not None else CouncilShard.from_env(self.council_size)

        # every variable ending in '_conf' is displayed in the tensorboard logs (trainer_council.py:37-68)
        self.recon_x_w_conf = hp['recon_x_w']
        self.recon_c_w_conf = hp['recon_c_w']
        self.recon_s_w_conf = hp['recon_s_w']
        self.recon_x_cyc_w_conf = hp['recon_x_cyc_w']
        self.gan_w_conf = hp['gan_w']
        self.vgg_w_conf = hp['vgg_w']
        self.abs_beginning_end_w_conf = hp['abs_beginning_end']
        self.flipOnOff_On_iteration_conf = hp['council']['flipOnOff_On_iteration']
        self.flipOnOff_Off_iteration_conf = hp['council']['flipOnOff_start_with']   # sic, :46-47
        self.council_abs_w_conf = hp['council_abs_w']
        self.council_w_conf = hp['council_w']
        self.council_start_at_iter_conf = hp['council']['council_start_at_iter']
        self.focus_loss_start_at_iter_conf = hp['focus_loss']['focus_loss_start_at_iter']
        self.mask_zero_or_one_w_conf = hp['mask_zero_or_one_w']
        self.mask_zero_or_one_center_conf = hp['focus_loss']['mask_zero_or_one_center']
        self.mask_zero_or_one_epsilon_conf = hp['focus_loss']['mask_zero_or_one_epsilon']
        self.mask_total_w_conf = hp['mask_total_w']
        self.mask_tv_w_conf = hp['mask_tv_w']
        self.batch_size_conf = hp['batch_size']
        self.do_w_loss_matching = hp['do_w_loss_matching']
        self.do_w_loss_matching_focus = hp['focus_loss']['do_w_loss_matching_focus']
        self.los_matching_hist_size_conf = hp['loss_matching_hist_size']
        self.do_a2b_conf = hp['do_a2b']
        self.do_b2a_conf = hp['do_b2a']
        self.w_match_b2a_conf = 1
        self.w_match_a2b_conf = 1
        self.do_council_loss = None
        self._dirs = [d for d in ('a2b', 'b2a') if hp['do_' + d]]

        # networks, built on the host in the reference's order (same RNG stream), trainer_council.py:100-133
        gen_a2b, dis_a2b, disc_a2b, gen_b2a, dis_b2a, disc_b2a = [], [], [], [], [], []
        for _ in range(self.council_size):
            if self.do_a2b_conf:
                gen_a2b.append(AdaINGen(hp['input_dim_a'], hp['gen'], cuda_device=cuda_device))
                dis_a2b.append(MsImageDis(hp['input_dim_a'], hp['dis'], cuda_device=cuda_device))
                if self.do_dis_council:
                    disc_a2b.append(MsImageDisCouncil(hp['input_dim_a'], hp['dis'], cuda_device=cuda_device))
            if self.do_b2a_conf:
                gen_b2a.append(AdaINGen(hp['input_dim_b'], hp['gen'], cuda_device=cuda_device))
                dis_b2a.append(MsImageDis(hp['input_dim_b'], hp['dis'], cuda_device=cuda_device))
                if self.do_dis_council:
                    disc_b2a.append(MsImageDisCouncil(hp['input_dim_b'], hp['dis'], cuda_device=cuda_device))
        self.instancenorm = nn.InstanceNorm2d(512, affine=False)     # :121 (VGG loss helper, unused; kept for order)
        self.style_dim = hp['gen']['style_dim']
        self.gen_a2b_s, self.gen_b2a_s, self.dis_a2b_s, self.dis_b2a_s = [], [], [], []
        if self.do_dis_council:
            self.dis_council_a2b_s, self.dis_council_b2a_s = [], []
        if self.do_a2b_conf:
            self.gen_a2b_s = nn.ModuleList(gen_a2b)
            self.dis_a2b_s = nn.ModuleList(dis_a2b)
            if self.do_dis_council:
                self.dis_council_a2b_s = nn.ModuleList(disc_a2b)
        if self.do_b2a_conf:
            self.gen_b2a_s = nn.ModuleList(gen_b2a)
            self.dis_b2a_s = nn.ModuleList(dis_b2a)
            if self.do_dis_council:
                self.dis_council_b2a_s = nn.ModuleList(disc_b2a)

        # fixed sampling noise (:135-138) -- drawn here to keep the RNG stream aligned
        display_size = int(hp['display_size'])
        self.s_a = torch.randn(display_size, self.style_dim, 1, 1)
        self.s_b = torch.randn(display_size, self.style_dim, 1, 1)

        # optimizers (:139-183): one flat-buffer Adam per member and network kind
        beta1, beta2, wd = hp['beta1'], hp['beta2'], hp['weight_decay']
        self.dis_opt_s, self.gen_opt_s, self.dis_scheduler_s, self.gen_scheduler_s = [], [], [], []
        if self.do_dis_council:
            self.dis_council_opt_s, self.dis_council_scheduler_s = [], []
        for i in range(self.council_size):
            dis_p, gen_p, disc_p = [], [], []
            if self.do_a2b_conf:
                dis_p += list(self.dis_a2b_s[i].parameters())
                gen_p += list(self.gen_a2b_s[i].parameters())
                if self.do_dis_council:
                    disc_p += list(self.dis_council_a2b_s[i].parameters())
            if self.do_b2a_conf:
                dis_p += list(self.dis_b2a_s[i].parameters())
                gen_p += list(self.gen_b2a_s[i].parameters())
                if self.do_dis_council:
                    disc_p += list(self.dis_council_b2a_s[i].parameters())
            self.dis_opt_s.append(FlatAdam(dis_p, lr=lr, betas=(beta1, beta2), weight_decay=wd))
            self.gen_opt_s.append(FlatAdam(gen_p, lr=lr, betas=(beta1, beta2), weight_decay=wd))
            if self.do_dis_council:
                self.dis_council_opt_s.append(FlatAdam(disc_p, lr=lr, betas=(beta1, beta2), weight_decay=wd))
            self.dis_scheduler_s.append(get_scheduler(self.dis_opt_s[i], hp))
            self.gen_scheduler_s.append(get_scheduler(self.gen_opt_s[i], hp))
            if self.do_dis_council:
                self.dis_council_scheduler_s.append(get_scheduler(self.dis_council_opt_s[i], hp))

        # weight initialisation, same call sequence as :185-197
        self.apply(weights_init(hp['init']))
        for i in range(self.council_size):
            if self.do_a2b_conf:
                self.gen_a2b_s[i].apply(weights_init(hp['init']))
                self.dis_a2b_s[i].apply(weights_init('gaussian'))
                if self.do_dis_council:
                    self.dis_council_a2b_s[i].apply(weights_init('gaussian'))
            if self.do_b2a_conf:
                self.gen_b2a_s[i].apply(weights_init(hp['init']))
                self.dis_b2a_s[i].apply(weights_init('gaussian'))
                if self.do_dis_council:
                    self.dis_council_b2a_s[i].apply(weights_init('gaussian'))
        self.vgg = None

        # loss-matching history (:71-92): device rings, filled with ones, one push per gen_update
        self._ring_n = int(self.los_matching_hist_size_conf)
        self._ring_pos = {d: [0] * self.council_size for d in self._dirs}
        self._ring_pos_c = {d: [0] * self.council_size for d in self._dirs}
        self._rings = None
        self._device = None
        # content-code cache ("re-encoding identical c_a"): the three updates of one iteration
        # encode the SAME batch with the SAME generator weights, so the encoder runs once per iteration
        self._img_cache = {}
        self._enc_cache = {}
        self._rep_cache = {}
        self._const_cache = {}
        self._streams = []
        # datapath of the convolutions with >= 32 channels -- forward, data-gradient and weight-gradient, generators and
        # both discriminators: "split" = fp16 x 3 MFMA on {hi, lo} fp16 operand planes (22 significand bits -- error
        # below the fp32 kernel's accumulation round-off -- at 2.5-2.8x its rate; fp32 storage and accumulation),
        # "fp32" = exact fp32 MFMA everywhere.  Layers whose INPUT has 3 / 6 / 12 channels are always exact fp32 (forward and
        # weight gradient; so is the data gradient of a 3- / 12-channel OUTPUT layer, whose dz is that thin).
        self._split_fwd = str(hp.get('cg_forward_precision', os.environ.get('CG_FORWARD_PRECISION', 'split'))) == 'split'

    # ------------------------------------------------------------------------------------
    # device placement
    # ------------------------------------------------------------------------------------
    def _nets(self, kind, d):
        return getattr(self, {'gen': 'gen_%s_s', 'dis': 'dis_%s_s', 'disc': 'dis_council_%s_s'}[kind] % d)

    def cuda(self, device=None):
        dev = torch.device(device if device is not None else self.cuda_device)
        if dev.type != 'cuda':
            raise hip.HipError("Council_Trainer runs on an MI355X only (got device %s): there is no CPU fallback" % dev)
        torch.cuda.set_device(dev)
        hip.load()
        if os.environ.get("CG_NATIVE_COLLECTIVES", "0") == "1" and self.shard.slice_comm is None:
            self.shard.use_native_collectives()       # collective: every rank reaches cuda()
        self.s_a, self.s_b = self.s_a.to(dev), self.s_b.to(dev)
        local = self.shard.local
        for i in local:                  # non-local members stay on the host, untouched
            for d in self._dirs:
                for kind in ('gen', 'dis') + (('disc',) if self.do_dis_council else ()):
                    net = self._nets(kind, d)[i]
                    for b in net.buffers():
                        b.data = b.data.to(dev)
        # One pool per optimizer kind over this rank's members: member k's parameters / gradients / Adam moments are slice
        # k of the pool's flat tensors, at a uniform stride -- the layout member-batched launches address (optim.ParamPool)
        self._pools = {'gen': ParamPool([self.gen_opt_s[i] for i in local]),
                       'dis': ParamPool([self.dis_opt_s[i] for i in local])}
        if self.do_dis_council:
            self._pools['disc'] = ParamPool([self.dis_council_opt_s[i] for i in local])
        for pool in self._pools.values():
            pool.materialize(dev)
        L = len(local)
        self._rings = {d: (torch.ones(L * self._ring_n, device=dev), torch.ones(L * self._ring_n, device=dev),
                           torch.ones(L, device=dev)) for d in self._dirs}
        self._device = dev
        # split-precision datapath: one lazily refreshed {hi, lo} fp16 copy of every pool's flat weight storage
        # (generators: instance-normalised activations travel unscaled; discriminators and all gradients: per-tensor
        # power-of-two scales chosen on the device)
        ops.X3_FORWARD = ops.X3_BACKWARD = ops.X3_DYNAMIC_INPUT = self._split_fwd
        if self._split_fwd:
            for kind, pool in self._pools.items():
                mgr = ops.SplitWeights(pool)
                pool.split = mgr
                for k, i in enumerate(local):
                    opt = pool.opts[k]
                    for d in self._dirs:
                        net = self._nets(kind, d)[i]
                        for m in net.modules():
                            m._cg_wmgr = mgr
                        # a checkpoint load rewrites the weights behind the optimizer's back: invalidate the copies
                        net.register_load_state_dict_post_hook(
                            lambda module, keys, _opt=opt: setattr(_opt, 'version', _opt.version + 1))
        # Member-batched execution (default): this rank's members run every layer as ONE launch, `CG_GROUP` members at a
        # time (default 4; the 31-bit buffer offsets of the kernels bound the batched tensors, _plan_groups).  CG_GROUP=1
        # walks the members one by one like the reference's loops (trainer_council.py:328,558,747,826,858); then each
        # member's kernels go to one of CG_MEMBER_STREAMS HIP streams so that low-occupancy launches overlap.
        self._group_max = max(1, int(os.environ.get('CG_GROUP', '4')))
        self._groups = None
        # The discriminator update and the council-discriminator update of one iteration touch disjoint networks and
        # both only READ the generators: their kernels go to two side streams that start from the same point (after the
        # shared prologue: batch upload, content codes, the generators' fp16 weight mirror) and so overlap on the GPU --
        # the HBM-bound passes of one hide under the MFMA-bound convolutions of the other.  The caller's stream waits
        # for a side stream at the end of the call that fed it, so anything enqueued afterwards (gen_update, a .item() on
        # a loss) is ordered behind it.  CG_OVERLAP_UPDATES=0 keeps everything on the caller's stream.
        # (members replicated over several ranks: both updates issue collectives -- the replicas' gradient all-reduce from
        # either stream, the image all-gather -- on different communicators; until that has run on RCCL the two updates
        # stay on one stream there unless CG_OVERLAP_UPDATES=1 asks for the overlap)
        self._overlap = os.environ.get('CG_OVERLAP_UPDATES', '0' if self.shard.dp > 1 else '1') != '0'
        self._side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)] if self._overlap else []
        self._e0 = None
        n = min(int(os.environ.get('CG_MEMBER_STREAMS', '2')), L)
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(n)] if n > 1 else []
        return self

    def _plan_groups(self, x):
        """Split this rank's members into runs of consecutive members that execute as one launch.  The largest batched
        tensor (the council discriminator's first full-resolution feature map: (1 + colleagues) * B samples per member,
        dis.dim channels) must stay below 2 GiB -- the kernels address operands with 31-bit byte offsets."""
        hp = self.__dict__.get('_hp_last') or {}
        n_rel = hp.get('council', {}).get('numberOfCouncil_dis_relative_iteration', 0) if hp else 0
        key = tuple(x.shape) + (n_rel,)
        if self._groups is not None and self._groups[0] == key:
            return self._groups[1]
        local = self.shard.local
        B, _, H, W = x.shape
        u = min(n_rel, self.council_size - 1) if self.do_dis_council else 0
        width = max(self._nets('dis', self._dirs[0])[local[0]].dim, 64)
        per_member = max((1 + u) * B, 2 * B) * width * H * W * 4
        if per_member >= (1 << 31):
            raise hip.HipError("one council member's largest activation (%d samples x %d channels x %dx%d fp32 = %.2f GiB) "
                               "exceeds the kernels' 31-bit byte offsets: lower batch_size or the image size"
                               % (max((1 + u) * B, 2 * B), width, H, W, per_member / 2.0 ** 30))
        g = 1
        for cand in range(1, len(local) + 1):
            if len(local) % cand == 0 and cand <= self._group_max and cand * per_member < (1 << 31):
                g = cand
        if self.shard.dp > 1:
            g = 1
        groups = [local[k:k + g] for k in range(0, len(local), g)]
        self._groups = (key, groups)
        return groups

    def _fork(self):
        """Member streams start after everything already queued on the caller's stream."""
        cur = torch.cuda.current_stream()
        for st in self._streams:
            st.wait_stream(cur)

    def _join(self):
        """The caller's stream continues after every member stream."""
        cur = torch.cuda.current_stream()
        for st in self._streams:
            cur.wait_stream(st)

    def _on(self, i, groups=None):
        """Stream context of the group led by member i (nothing to overlap when one group holds every member)."""
        if not self._streams or (groups is not None and len(groups) == 1):
            return contextlib.nullcontext()
        return torch.cuda.stream(self._streams[self.shard.local.index(i) % len(self._streams)])

    def to(self, *args, **kwargs):
        dev = args[0] if args else kwargs.get('device')
        return self.cuda(dev)

    def _ready(self):
        if self._device is None:
            self.cuda(self.cuda_device)
        # the datapath choice belongs to THIS trainer: the ops-level switches are re-applied on entry of every compute
        # method, so trainers with different `cg_forward_precision` can alternate inside one process
        ops.X3_FORWARD = ops.X3_BACKWARD = ops.X3_DYNAMIC_INPUT = self._split_fwd

    def _img(self, x, slot=None):
        """Device / NHWC copy of a caller tensor.  The copy is re-used while the caller passes the very same
        (unmodified) tensor object again -- train.py:244-250 hands one batch to all three updates."""
        ent = self._img_cache.get(slot) if slot is not None else None
        if ent is not None and ent[0] is x and ent[1] == x._version:
            return ent[2]
        y = x.to(self._device, dtype=torch.float32)
        if slot is not None:
            y = self.shard.batch_slice(y)                  # training batches: this rank's samples (parallel.py)
        y = y.contiguous(memory_format=torch.channels_last)
        if slot is not None:
            self._img_cache[slot] = (x, x._version, y)     # holding `x` keeps its address from being recycled
        return y

    def _refresh_split_weights(self, d, i):
        """Point the decoder's split-precision trunk at the current {hi, lo} fp16 weights (ops.SplitWeights re-splits
        the generator's flat parameter buffer lazily, once per generator step)."""
        if not self._split_fwd:
            return
        for blk in self._nets('gen', d)[i].dec._split_blocks():
            blk._cg_wsplit = blk._cg_wmgr.get(blk.conv.weight)

    @contextlib.contextmanager
    def _split_decode(self, d, i):
        """Scope in which member i's decoder may take the split-precision trunk (weights refreshed on entry)."""
        self._refresh_split_weights(d, i)
        dec = self._nets('gen', d)[i].dec
        dec.split_active = self._split_fwd
        try:
            yield
        finally:
            dec.split_active = False

    @contextlib.contextmanager
    def _fresh_mirrors(self, *kinds):
        """Scope of one update: the {hi, lo} fp16 mirrors of the pools it READS are brought up to date once, on the stream
        the update starts on (before its member groups fork), and stay as they are until it ends -- a group's optimizer
        step changes only its own members' weights, which no other group reads (ops.SplitWeights.frozen)."""
        mgrs = [self._pools[k].split for k in kinds if k in self._pools and self._pools[k].split is not None]
        for m in mgrs:
            m.refresh()
            m.frozen = True
        try:
            yield
        finally:
            for m in mgrs:
                m.frozen = False

    def _weights_version(self, d, grp):
        opts = [self.gen_opt_s[i] for i in grp]
        gen = self._nets('gen', d)[grp[0]]
        return (tuple(o.version for o in opts), sum(p._version for p in gen.enc_content.parameters()))

    def _rep(self, x, g):
        """The batch repeated once per member of a launch (member-major): every member sees the same images."""
        if g == 1:
            return x
        key = (id(x), g)
        ent = self._rep_cache.get(key)
        if ent is None or ent[0] is not x:
            if len(self._rep_cache) > 8:
                self._rep_cache.clear()
            ent = (x, ops.take_rows(x, None, list(range(x.shape[0])) * g))
            self._rep_cache[key] = ent
        return ent[1]

    def _content(self, d, grp, x, need_grad):
        """Content codes of the members of `grp` for batch x (the member-major repetition of the NHWC device copy from
        _img).  Encoded once per (batch, generator weights) with the autograd tape attached; the discriminator updates
        use it detached, gen_update back-propagates through it (and drops it, the tape being consumed)."""
        key = (id(x), self._weights_version(d, grp))
        slot = (d, grp[0], len(grp))
        ent = self._enc_cache.get(slot)
        if ent is None or ent[0] != key or ent[1] is not x:
            with torch.enable_grad():
                content = self._nets('gen', d)[grp[0]].encode_content(x)
            ent = (key, x, content)
            self._enc_cache[slot] = ent
        if need_grad:
            del self._enc_cache[slot]
            return ent[2]
        return ent[2].detach()

    def _side_stream(self, k, x, groups, prologue):
        """Context of side stream k for one discriminator-side update (see cuda()).  `prologue`: this call is the first
        of the iteration for batch x -- run what both updates read on the caller's stream and mark the fork point;
        otherwise continue from the mark left for this very batch, or from the caller's stream if there is none."""
        if not self._overlap:
            return contextlib.nullcontext(), None
        main = torch.cuda.current_stream()
        # the fork mark is valid for THIS batch and THESE generator weights only: a generator step (or a checkpoint load)
        # after it means the side stream must be ordered behind the caller's stream again (gen_update also clears it)
        key = tuple(id(x[d]) for d in self._dirs) + (self._pools['gen'].version,)
        side = self._side[k]
        if prologue:
            less = self._hp_last['council']['discriminetro_less_style_by'] if self.do_dis_council else 0
            for grp in groups:
                for d in self._dirs:
                    xr = self._rep(x[d], len(grp))
                    if less != 0 and self.council_size > 1:
                        self._rep(x[d], 2 * len(grp))
                    with ops.members(len(grp)):
                        self._content(d, grp, xr, need_grad=False)
            if self._split_fwd:
                self._pools['gen'].split.refresh()
            ev = torch.cuda.Event()
            ev.record(main)
            self._e0 = (ev, key)
            side.wait_event(ev)
        elif self._e0 is not None and self._e0[1] == key:
            side.wait_event(self._e0[0])
        else:
            side.wait_stream(main)
        return torch.cuda.stream(side), (main, side)

    @staticmethod
    def _side_done(tok):
        if tok is not None:
            tok[0].wait_stream(tok[1])        # the caller's stream continues behind the side stream (no host wait)

    def _const(self, value, n):
        """Device vector of n copies of `value` (upstream gradients of the per-member loss vectors), cached."""
        key = (float(value), n)
        t = self._const_cache.get(key)
        if t is None:
            t = self._const_cache[key] = torch.full((n,), float(value), dtype=torch.float32, device=self._device)
            torch.cuda.current_stream().synchronize()      # created once, then read from several streams
        return t

    def _noise(self, n):
        # CPU RNG then upload, exactly as the reference (trainer_council.py:284-285,741,744,807-809)
        return torch.randn(n, self.style_dim, 1, 1)

    def _style(self, n):
        """Style codes of a training batch: drawn for the WHOLE batch on every rank (replicated RNG stream), then cut
        to this rank's samples when a member spans several ranks."""
        return self.shard.batch_slice(self._noise(n))

    def _full_batch(self, loss):
        """Detached full-batch value of a batch-mean loss (the loss-matching history must be the same on every
        replica of a member)."""
        v = loss.detach()
        return self.shard.replica_mean_(v.clone()) if self.shard.dp > 1 else v

    def _sync_grads(self, pool, k0, g):
        """Full-batch gradient = mean of the member replicas' gradients: one all-reduce of the members' gradient slices."""
        if self.shard.dp > 1:
            self.shard.replica_mean_(pool.grad[k0 * pool.stride:(k0 + g) * pool.stride])

    def _upload(self, t):
        """Host tensor -> device without stalling the host: a pageable-memory copy blocks until the stream has drained,
        which would serialise the host with the GPU at every update; a small ring of pinned staging buffers (each guarded
        by the event of its last copy) keeps the enqueue running ahead."""
        ring = self.__dict__.setdefault('_pin_ring', {'i': 0, 'slots': [None] * 16})
        k = ring['i'] % 16
        ring['i'] += 1
        slot = ring['slots'][k]
        if slot is None or slot[0].numel() < t.numel():
            slot = [torch.empty(max(t.numel(), 4096), dtype=torch.float32, pin_memory=True), None]
            ring['slots'][k] = slot
        if slot[1] is not None:
            slot[1].synchronize()
        stage = slot[0][:t.numel()].view(t.shape)
        stage.copy_(t)
        out = stage.to(self._device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        return out

    # ------------------------------------------------------------------------------------
    # schedules (host integers), trainer_council.py:541-555, 784-801
    # ------------------------------------------------------------------------------------
    @staticmethod
    def _flip_state(hp):
        c = hp['council']
        cycle = c['flipOnOff_On_iteration'] + c['flipOnOff_Off_iteration']
        cur = hp['iteration'] % cycle
        start = c['flipOnOff_On_iteration'] if c['flipOnOff_start_with'] else c['flipOnOff_Off_iteration']
        return c['flipOnOff_start_with'] if cur < start else (not c['flipOnOff_start_with'])

    # ------------------------------------------------------------------------------------
    # dis_update, trainer_council.py:735-780
    # ------------------------------------------------------------------------------------
    def dis_update(self, x_a=None, x_b=None, hyperparameters=None):
        hp = self._hp_last = hyperparameters
        self._ready()
        x = {'a2b': self._img(x_a, 'a'), 'b2a': self._img(x_b, 'b')}      # source image per direction
        tgt = {'a2b': x['b2a'], 'b2a': x['a2b']}                 # real image of the target domain
        groups = self._plan_groups(x[self._dirs[0]])
        ctx, tok = self._side_stream(0, x, groups, prologue=True)
        with ctx, self._fresh_mirrors('gen', 'dis'):
            pool = self._pools['dis']
            pool.zero_grad()
            s = {}
            if self.do_a2b_conf:
                s['a2b'] = self._style(x_b.size(0))
                self.loss_dis_a2b_s = [0] * self.council_size
            if self.do_b2a_conf:
                s['b2a'] = self._style(x_a.size(0))
                self.loss_dis_b2a_s = [0] * self.council_size
            self.loss_dis_total_s = [0] * self.council_size
            s_dev = {}
            self._fork()
            for grp in groups:
                g, lead, k0 = len(grp), grp[0], self.shard.local.index(grp[0])
                with self._on(lead, groups), ops.members(g):
                    losses, ups = [], []
                    for d in self._dirs:
                        gen = self._nets('gen', d)[lead]
                        if (d, g) not in s_dev:
                            s_dev[(d, g)] = self._upload(s[d].repeat(g, 1, 1, 1) if g > 1 else s[d])
                        xr = self._rep(x[d], g)
                        content = self._content(d, grp, xr, need_grad=False)
                        with torch.no_grad(), self._split_decode(d, lead):
                            x_fake = gen.decode(content, s_dev[(d, g)], xr)
                        # :775-777 -- only the a2b term is scaled by gan_w (reference quirk, kept): the unscaled loss is what
                        # train.py logs, the scale rides on the upstream gradient
                        w = float(hp['gan_w']) if d == 'a2b' else 1.0
                        l = self._nets('dis', d)[lead].calc_dis_loss(x_fake, tgt[d]).view(-1)       # one loss per member
                        losses.append(l)
                        ups.append(self._const(w, g))
                        for m, i in enumerate(grp):
                            getattr(self, 'loss_dis_%s_s' % d)[i] = l.detach()[m]
                    torch.autograd.backward(losses, ups)          # members and directions own disjoint parameters
                    for m, i in enumerate(grp):
                        tot = None
                        for d, l in zip(self._dirs, losses):
                            w = float(hp['gan_w']) if d == 'a2b' else 1.0
                            t = l.detach()[m] if w == 1.0 else l.detach()[m] * w
                            tot = t if tot is None else tot + t
                        self.loss_dis_total_s[i] = tot
                    ops.wgrad_join()
                    self._sync_grads(pool, k0, g)
                    pool.step(k0, g, lockstep=g > 1)
            self._join()
        self._side_done(tok)

    # ------------------------------------------------------------------------------------
    # dis_council_update, trainer_council.py:782-883
    # ------------------------------------------------------------------------------------
    def dis_council_update(self, x_a=None, x_b=None, hyperparameters=None):
        hp = self._hp_last = hyperparameters
        c = hp['council']
        if self.council_size <= 1 or c['numberOfCouncil_dis_relative_iteration'] == 0:
            print('no council discriminetor is needed (council size <= 1 or numberOfCouncil_dis_relative_iteration == 0)')
            return
        self.do_council_loss = self._flip_state(hp)
        if not c['flipOnOff']:
            self.do_council_loss = c['flipOnOff_start_with']
        if not self.do_council_loss or hp['council_w'] == 0 or hp['iteration'] < c['council_start_at_iter']:
            return
        self._ready()
        x = {'a2b': self._img(x_a, 'a'), 'b2a': self._img(x_b, 'b')}
        groups = self._plan_groups(x[self._dirs[0]])
        ctx, tok = self._side_stream(1, x, groups, prologue=False)
        with ctx, self._fresh_mirrors('gen', 'disc'):
            pool = self._pools['disc']
            pool.zero_grad()
            s, s_less = {}, {}
            if self.do_b2a_conf:                       # s_a is drawn first here (:806-809)
                s['b2a'] = self._style(x_a.size(0))
            if self.do_a2b_conf:
                s['a2b'] = self._style(x_b.size(0))
            less = c['discriminetro_less_style_by']
            n_rel = c['numberOfCouncil_dis_relative_iteration']
            L = len(self.shard.local)

            # ---- every member's translation (full style) and comparison image (reduced style) ------------------------
            x_full = {d: {} for d in self._dirs}                 # group lead -> the group's own translations [g*B]
            x_cmp_local = {}
            for d in self._dirs:
                b = x[d].shape[0]
                x_cmp_local[d] = torch.empty((L * b,) + tuple(x[d].shape[1:]), dtype=torch.float32, device=self._device,
                                             memory_format=torch.channels_last)
            s_dev = {}
            self._fork()
            for grp in groups:
                g, lead, k0 = len(grp), grp[0], self.shard.local.index(grp[0])
                with self._on(lead, groups), ops.members(g):
                    for d in self._dirs:
                        gen = self._nets('gen', d)[lead]
                        b = x[d].shape[0]
                        xr = self._rep(x[d], g)
                        content = self._content(d, grp, xr, need_grad=False)
                        with torch.no_grad(), self._split_decode(d, lead):
                            if less != 0:
                                # the two translations differ only in the style code: one decode over 2B samples per member
                                # (every operator of the decoder is per sample) -- twice the rows per launch, half the launches
                                if (d, g) not in s_dev:
                                    s_dev[(d, g)] = self._upload(torch.cat((s[d], s[d] * less), 0).repeat(g, 1, 1, 1))
                                twice = [m * b + r for m in range(g) for _ in range(2) for r in range(b)]
                                both = gen.decode(ops.take_rows(content, None, twice), s_dev[(d, g)], self._rep(x[d], 2 * g))
                                own = [m * 2 * b + r for m in range(g) for r in range(b)]
                                x_full[d][lead] = ops.take_rows(both, None, own)
                                ops.take_rows(both, None, [i + b for i in own], out=x_cmp_local[d][k0 * b:(k0 + g) * b])
                            else:
                                if (d, g) not in s_dev:
                                    s_dev[(d, g)] = self._upload(s[d].repeat(g, 1, 1, 1) if g > 1 else s[d])
                                x_full[d][lead] = gen.decode(content, s_dev[(d, g)], xr)
                                ops.take_rows(x_full[d][lead], None, list(range(g * b)), out=x_cmp_local[d][k0 * b:(k0 + g) * b])
            self._join()      # every member's council discriminator reads the OTHER members' images
            # the ONE cross-member exchange (trainer_council.py:853-856): every member's comparison image, member-major
            x_cmp = {d: self.shard.exchange_flat(x_cmp_local[d]) for d in self._dirs}

            self.loss_dis_council_a2b_s = [0] * self.council_size
            self.loss_dis_council_b2a_s = [0] * self.council_size
            self.loss_dis_council_total_s = [0] * self.council_size
            scale = float(hp['council_w']) / float(n_rel)                      # :878-880
            picks = [self.draw_colleagues(i, self.council_size, n_rel) for i in range(self.council_size)]   # every rank replays
            self._fork()                                                                                  # every member's draws
            for grp in groups:
                g, lead, k0 = len(grp), grp[0], self.shard.local.index(grp[0])
                pk = [[(j, float(picks[i].count(j))) for j in sorted(set(picks[i]))] for i in grp]
                with self._on(lead, groups), ops.members(g):
                    losses = []
                    for d in self._dirs:
                        l = self._nets('disc', d)[lead].calc_dis_loss_members(
                            x_full[d][lead], x_cmp[d], pk, x[d], fake_weight=float(len(picks[lead])), weight=scale).view(-1)
                        losses.append(l)
                        for m, i in enumerate(grp):
                            getattr(self, 'loss_dis_council_%s_s' % d)[i] = l.detach()[m] / scale
                    torch.autograd.backward(losses, [self._const(1.0, g)] * len(losses))
                    for m, i in enumerate(grp):
                        tot = None
                        for l in losses:
                            tot = l.detach()[m] if tot is None else tot + l.detach()[m]
                        self.loss_dis_council_total_s[i] = tot
                    ops.wgrad_join()
                    self._sync_grads(pool, k0, g)
                    pool.step(k0, g, lockstep=g > 1)
            self._join()
        self._side_done(tok)

    @staticmethod
    def draw_colleagues(i, council_size, n_rel):
        """trainer_council.py:861-868 (Python global RNG, identical call sequence)."""
        picks = []
        pool = list(range(0, i)) + list(range(i + 1, council_size))
        for k in range(n_rel):
            if k == council_size:
                break
            if len(pool) == 0:
                pool = list(range(0, i)) + list(range(i + 1, council_size))
            j = random.choice(pool)
            pool.remove(j)
            picks.append(j)
        return picks

    # ------------------------------------------------------------------------------------
    # gen_update, trainer_council.py:280-634
    # ------------------------------------------------------------------------------------
    def gen_update(self, x_a, x_b, hyperparameters, iterations=0):
        hp = self._hp_last = hyperparameters
        self.hyperparameters = hp
        self._ready()
        lib = hip.load()
        x = {'a2b': self._img(x_a, 'a'), 'b2a': self._img(x_b, 'b')}
        groups = self._plan_groups(x[self._dirs[0]])
        pool = self._pools['gen']
        self._e0 = None                    # the generators are about to change: no side stream may start from the old mark
        pool.zero_grad()
        s_a = self._style(x_a.size(0))     # both drawn, s_a first (:284-285)
        s_b = self._style(x_b.size(0))
        s = {'a2b': s_b, 'b2a': s_a}
        fl = hp['focus_loss']
        focus_live = hp['iteration'] > fl['focus_loss_start_at_iter']
        self.council_w_conf = hp['council_w'] if hp['iteration'] > hp['council']['council_start_at_iter'] else 0
        self.mask_zero_or_one_w_conf = hp['mask_zero_or_one_w'] if focus_live else 0
        self.mask_total_w_conf = hp['mask_total_w'] if focus_live else 0
        self.mask_tv_w_conf = hp['mask_tv_w'] if focus_live else 0
        focus_on = focus_live and (hp['mask_zero_or_one_w'] != 0 or hp['mask_total_w'] != 0)      # :390

        self.do_council_loss = self._flip_state(hp)                                                # :541-555
        if not hp['council']['flipOnOff']:
            self.do_council_loss = True
        if hp['iteration'] < hp['council']['council_start_at_iter']:
            self.do_council_loss = False
        council_on = (hp['council_w'] != 0) and self.do_council_loss and self.council_size > 1 and self.do_dis_council

        C = self.council_size
        ab = {'a2b': 'ab', 'b2a': 'ba'}
        self.loss_gen_total_s = [0] * C
        for d in self._dirs:
            setattr(self, 'loss_gen_adv_%s_s' % d, [0] * C)
            setattr(self, 'loss_gen_mask_zero_one_%s_s' % ab[d], [0] * C if focus_on and hp['mask_zero_or_one_w'] != 0 else [])
            setattr(self, 'loss_gen_mask_total_%s_s' % ab[d], [0] * C)
            setattr(self, 'loss_gen_mask_TV_%s_s' % ab[d], [0] * C)
            setattr(self, 'council_loss_%s_s' % ab[d], [0] * C)
        for d in ('a2b', 'b2a'):
            if d not in self._dirs:
                setattr(self, 'loss_gen_adv_%s_s' % d, [0] * C)

        frozen = []
        for i in self.shard.local:       # no weight gradients for D / council-D in this update
            for d in self._dirs:
                for kind in ('dis',) + (('disc',) if self.do_dis_council else ()):
                    for p in self._nets(kind, d)[i].parameters():
                        if p.requires_grad:
                            p.requires_grad_(False)
                            frozen.append(p)
        s_dev = {}
        n_ring = self._ring_n
        mirrors = self._fresh_mirrors('gen', 'dis', 'disc')
        mirrors.__enter__()
        self._fork()
        try:
            for grp in groups:
                g, lead, k0 = len(grp), grp[0], self.shard.local.index(grp[0])
                with self._on(lead, groups), ops.members(g):
                    roots, ups, totals = [], [], []
                    for d in self._dirs:
                        gen = self._nets('gen', d)[lead]
                        if (d, g) not in s_dev:
                            s_dev[(d, g)] = self._upload(s[d].repeat(g, 1, 1, 1) if g > 1 else s[d])
                        xr = self._rep(x[d], g)
                        x_fake = gen.decode(self._content(d, grp, xr, need_grad=True), s_dev[(d, g)], xr)
                        mask = gen.dec.mask_s
                        ftot = adv = lc = w_dev = None
                        if focus_on:                                                   # :390-451
                            ftot, parts = ops.focus_loss(mask, fl['mask_zero_or_one_center'], fl['mask_zero_or_one_epsilon'],
                                                         hp['mask_zero_or_one_w'], hp['mask_total_w'], hp['mask_tv_w'],
                                                         fl['mask_small_use_abs'], fl['mask_small_use_square'],
                                                         reduce=self.shard.replica_mean_ if self.shard.dp > 1 else None)
                            roots.append(ftot.view(-1))
                            ups.append(self._const(1.0, g))
                            parts = parts.view(g, 3)
                            for m, i in enumerate(grp):
                                if hp['mask_zero_or_one_w'] != 0:
                                    getattr(self, 'loss_gen_mask_zero_one_%s_s' % ab[d])[i] = parts[m, 0]
                                if hp['mask_total_w'] != 0:
                                    getattr(self, 'loss_gen_mask_total_%s_s' % ab[d])[i] = parts[m, 1]
                                if hp['mask_tv_w'] != 0:
                                    getattr(self, 'loss_gen_mask_TV_%s_s' % ab[d])[i] = parts[m, 2]
                        ring_g, ring_c, w_all = self._rings[d]
                        rg = ring_g[k0 * n_ring:(k0 + g) * n_ring]
                        rc = ring_c[k0 * n_ring:(k0 + g) * n_ring]
                        if hp['gan_w'] != 0:                                           # :498-529
                            adv = self._nets('dis', d)[lead].calc_gen_loss(x_fake).view(-1)
                            adv_full = self._full_batch(adv)
                            for m, i in enumerate(grp):
                                getattr(self, 'loss_gen_adv_%s_s' % d)[i] = adv_full[m]
                            if self.do_w_loss_matching:
                                check(lib.cg_ring_push_g(ptr(rg), n_ring, self._ring_pos[d][lead], ptr(adv_full), g, stream()),
                                      "cg_ring_push")
                                for i in grp:
                                    self._ring_pos[d][i] += 1
                            roots.append(adv)
                            ups.append(self._const(float(hp['gan_w']), g))
                        if council_on:                                                 # :558-624
                            lc = self._nets('disc', d)[lead].calc_gen_loss(x_fake, xr).view(-1)
                            if self.do_w_loss_matching:
                                w_dev = w_all[k0:k0 + g]
                                check(lib.cg_loss_match_g(ptr(rg), ptr(rc), n_ring, self._ring_pos_c[d][lead],
                                                          ptr(self._full_batch(lc)), ptr(w_dev), g, stream()), "cg_loss_match")
                                for i in grp:
                                    self._ring_pos_c[d][i] += 1
                                setattr(self, 'w_match_%s_conf' % d, w_dev[0])
                        # per-member objective + the upstream gradient of the council term (council_w x matching weight)
                        total, council, gcouncil = ops.gen_total(ftot, adv.detach() if adv is not None else None,
                                                                 lc.detach() if lc is not None else None, w_dev,
                                                                 hp['gan_w'], hp['council_w'], g)
                        if council_on:
                            roots.append(lc)
                            ups.append(gcouncil)
                            for m, i in enumerate(grp):
                                getattr(self, 'council_loss_%s_s' % ab[d])[i] = council[m]
                        totals.append(total)
                    for m, i in enumerate(grp):
                        tot = None
                        for t in totals:
                            tot = t[m] if tot is None else tot + t[m]
                        self.loss_gen_total_s[i] = tot
                    torch.autograd.backward(roots, ups)
                    ops.wgrad_join()
                    self._sync_grads(pool, k0, g)
                    pool.step(k0, g, lockstep=g > 1)
        finally:
            self._join()
            mirrors.__exit__(None, None, None)
            for p in frozen:
                p.requires_grad_(True)

    # ------------------------------------------------------------------------------------
    # forward-only paths, trainer_council.py:252-278, 643-733
    # ------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, x_a=None, x_b=None, s_a=None, s_b=None, council_member_to_sample_vec=None, return_mask=True):
        """trainer_council.py:643-733: the 8-tuple (x_a_s, masks | reconstructions, x_ab1, x_ab2, x_b_s, ..., x_ba1, x_ba2),
        rows ordered (image n, member j).  With the council sharded over several ranks every rank renders its own
        members and one all-gather per output completes the strips, so every rank returns the reference's layout."""
        self._ready()
        self.eval()
        out = {}
        members = list(range(self.council_size) if council_member_to_sample_vec is None else council_member_to_sample_vec)
        sharded = self.shard.world_size > 1
        for d, xin, s_fixed in (('a2b', x_a, s_b if s_b is not None else self.s_b),
                                ('b2a', x_b, s_a if s_a is not None else self.s_a)):
            if d not in self._dirs:
                continue
            xin = self._img(xin)
            s1 = s_fixed.to(self._device)
            s2 = self._noise(xin.size(0)).to(self._device)
            per = {}                                   # member -> (second, x1, x2), each [n images, C, H, W]
            for j in (self.shard.local if sharded else members):
                gen = self._nets('gen', d)[j]
                second, x1, x2 = [], [], []
                for n in range(xin.size(0)):
                    xi = xin[n:n + 1]
                    if not return_mask:
                        content, s_fake = gen.encode(xi)
                        second.append(gen.decode(content, s_fake, xi))
                        x1.append(gen.decode(content, s1[n:n + 1], xi))
                    else:
                        content = gen.encode_content(xi)       # the style code (:671) is not used on this branch
                        im, m = gen.decode(content, s1[n:n + 1], xi, return_mask=True)
                        x1.append(im)
                        second.append(m)
                    x2.append(gen.decode(content, s2[n:n + 1], xi))
                per[j] = (torch.cat(second), torch.cat(x1), torch.cat(x2))
            if sharded:
                full = [self.shard.exchange([per[j][k] for j in self.shard.local]) for k in range(3)]
                per = {j: tuple(full[k][j] for k in range(3)) for j in members}
            rows = [(n, j) for n in range(xin.size(0)) for j in members]
            out[d] = (torch.cat([xin[n:n + 1] for n, _ in rows]),) + tuple(
                torch.cat([per[j][k][n:n + 1] for n, j in rows]) for k in range(3))
        self.train()
        none4 = (None, None, None, None)
        return out.get('a2b', none4) + out.get('b2a', none4)

    def forward(self, x_a, s_t=None, x_b=None, s_a=None, s_b=None):
        """trainer_council.py:252-278 (its a2b branch references a non-existent self.gen_a2b; the
        evident intent -- every member's translation -- is what runs here)."""
        self._ready()
        self.eval()
        if s_t is not None:
            s_a = s_b = s_t
        res = {}
        with torch.no_grad():
            for d, xin, sd in (('a2b', x_a, s_b if s_b is not None else self.s_b),
                               ('b2a', x_b if x_b is not None else x_a, s_a if s_a is not None else self.s_a)):
                if d not in self._dirs:
                    continue
                xin = self._img(xin)
                sd = sd.to(self._device)
                res[d] = []
                for i in self.shard.local:
                    gen = self._nets('gen', d)[i]
                    res[d].append(gen.decode(gen.encode_content(xin), sd, xin))
                if self.shard.world_size > 1:          # every rank returns every member's translation
                    full = self.shard.exchange(res[d])
                    res[d] = [full[i] for i in range(self.council_size)]
        if self.do_a2b_conf and self.do_b2a_conf:
            return res['a2b'], res['b2a']
        return res['b2a'] if self.do_b2a_conf else res['a2b']

    def update_learning_rate(self):
        """trainer_council.py:885-896."""
        for sch in self.dis_scheduler_s + self.gen_scheduler_s + (self.dis_council_scheduler_s if self.do_dis_council else []):
            if sch is not None:
                sch.step()

    # ------------------------------------------------------------------------------------
    # checkpoints, trainer_council.py:898-992 (file names / dict keys / tensor shapes unchanged;
    # each rank writes and reads the files of its own members)
    # ------------------------------------------------------------------------------------
    def _io_device(self):
        """Device the checkpoint tensors live on.  save() / resume() are file I/O, not compute: they also work on a
        trainer that was never moved to a GPU (checkpoint conversion / inspection on a host; the optimizers' flat buffers
        are then laid out in host memory and move with a later .cuda())."""
        if self._device is not None:
            return self._device
        if torch.cuda.is_available():
            self._ready()
            return self._device
        for i in self.shard.local:
            for opt in (self.gen_opt_s[i], self.dis_opt_s[i]) + ((self.dis_council_opt_s[i],) if self.do_dis_council else ()):
                opt.materialize('cpu')
        return torch.device('cpu')

    @staticmethod
    def _plain_state(net):
        """state_dict with independent, contiguous (OIHW) tensors: the parameters are channels_last VIEWS into the
        optimizer's flat buffer, and torch.save would serialise that whole buffer once per file."""
        sd = net.state_dict()
        for k in list(sd):
            sd[k] = sd[k].detach().clone(memory_format=torch.contiguous_format)
        return sd

    def save(self, snapshot_dir, iterations):
        self._io_device()
        if self.shard.slice_idx != 0:      # replicas of a member hold identical weights: the first one writes
            return
        for i in self.shard.local:
            tag = '_%d_%08d.pt' % (i, iterations + 1)
            for d in self._dirs:
                torch.save({d: self._plain_state(self._nets('gen', d)[i])}, os.path.join(snapshot_dir, d + '_gen' + tag))
                torch.save({d: self._plain_state(self._nets('dis', d)[i])}, os.path.join(snapshot_dir, d + '_dis' + tag))
                if self.do_dis_council:
                    torch.save({d: self._plain_state(self._nets('disc', d)[i])},
                               os.path.join(snapshot_dir, d + '_dis_council' + tag))
            opt = {'gen': self.gen_opt_s[i].state_dict(), 'dis': self.dis_opt_s[i].state_dict()}
            if self.do_dis_council:
                opt['dis_council'] = self.dis_council_opt_s[i].state_dict()
            torch.save(opt, os.path.join(snapshot_dir, 'optimizer_%d.pt' % i))

    def resume(self, checkpoint_dir, hyperparameters):
        dev = self._io_device()
        self._e0 = None
        self._enc_cache.clear()
        iterations = 0
        for i in self.shard.local:
            for kind, key in (('gen', 'gen_%d' % i), ('dis', 'dis_%d' % i)) + \
                    ((('disc', 'dis_council_%d' % i),) if self.do_dis_council else ()):
                for d in self._dirs:
                    name = get_model_list(checkpoint_dir, d + '_' + key)
                    if name is None:
                        warnings.warn('Failed to find %s checkpoint, did not load model' % key)
                        continue
                    print('loading: ' + name)
                    self._nets(kind, d)[i].load_state_dict(torch.load(name, map_location=dev)[d])
                    if kind == 'gen':
                        iterations = int(name[-11:-3])
            try:
                sd = torch.load(os.path.join(checkpoint_dir, 'optimizer_%d.pt' % i), map_location=dev)
                self.dis_opt_s[i].load_state_dict(sd['dis'])
                self.gen_opt_s[i].load_state_dict(sd['gen'])
                if self.do_dis_council:
                    self.dis_council_opt_s[i].load_state_dict(sd['dis_council'])
                self.dis_scheduler_s[i] = get_scheduler(self.dis_opt_s[i], hyperparameters, iterations)
                self.gen_scheduler_s[i] = get_scheduler(self.gen_opt_s[i], hyperparameters, iterations)
                if self.do_dis_council:
                    self.dis_council_scheduler_s[i] = get_scheduler(self.dis_council_opt_s[i], hyperparameters, iterations)
            except Exception:
                warnings.warn('some optimizer FAILED to load ')
        if iterations > 0:
            print('Resume from iteration %d' % iterations)
        else:
            warnings.warn('FAILED TO RESUME STARTED FROM 0')
        return iterations
