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
import gc
import os
import random
import time
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import hip, ops
from .hip import check, ptr, stream
from .networks import AdaINGen, MsImageDis, MsImageDisCouncil
from .optim import FlatAdam, ParamPool
from .graphs import HostInputs, Segment
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
        self._hp_cfg = hp
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
        # trainer_council.py:65-68: logged by utils.write_loss like every '*_conf' member; they only move under
        # focus_loss.do_w_loss_matching_focus, which _check_supported() refuses (False in every shipped config)
        self.w_match_focus_a2b_conf = 1
        self.w_match_focus_b2a_conf = 1
        self.w_match_focus_zero_one_a2b_conf = 1
        self.w_match_focus_zero_one_b2a_conf = 1
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
        # hipGraph mode (CG_GRAPH=1 / cfg['cg_graph']): the device work of every update is captured once per (shapes,
        # schedule flags) and replayed; the host only refreshes the static input buffers (graphs.HostInputs).  Host cost
        # per iteration drops from ~17 ms of Python to < 1 ms -- what a rank needs once it holds ONE member (20 ms of GPU
        # work per iteration).  The two discriminator-side updates then stay on the caller's stream (a graph is replayed
        # on one stream); replicated members (gradient all-reduces inside the update) keep the eager path.
        # Default: on when the council is sharded over several ranks (a rank with one member has ~20 ms of GPU work per
        # iteration against ~15 ms of eager enqueue), off on a single GPU (GPU-bound either way; eager keeps the overlap of
        # the two discriminator-side updates: 68.5 vs 69.1 ms per step, profiles/r03_g_*).
        # ... and on for a single gen / dis pair (council 1: a short, enqueue-bound step with no second update to overlap:
        # 8.89 eager vs 8.44 ms replayed at 128x128 batch 8, profiles/r03_s_graph_vs_eager.txt)
        # Round 6: a SHARDED rank decides by measurement ('auto').  Graph replay runs the two discriminator-side updates on one stream
        # where eager overlaps them on two: on a one-member rank eager is 19.05 ms against 19.76 ms replayed (profiles/
        # r06_one_member_rank.txt) -- provided the host keeps up (14.2 ms of enqueue work per iteration on the pool's boxes).  'auto'
        # starts eagerly, times iterations 3 and 4 (host's own enqueue work against the iteration's GPU time, queue empty at the
        # start) and switches to graph replay for the rest of the run when the host needs more than CG_GRAPH_AUTO_RATIO (0.80) of
        # the GPU time (same call, a one-member rank: host 14.7 of 18.75 ms -> eager 18.75 vs 19.16 ms replayed; host 18.0 ms -> eager 19.24: the
        # eager gain needs head-room).  Eager and replayed iterations are bit-identical and issue the same collectives in the same order, so
        # ranks may decide differently.
        dflt = 'auto' if (self.shard.world_size > 1 and self.shard.dp == 1) else ('1' if self.council_size == 1 else '0')
        mode = str(self._hp_cfg.get('cg_graph', os.environ.get('CG_GRAPH', dflt)))
        self._graph_mode = mode == '1' and self.shard.dp == 1
        self._auto = None
        self.graph_auto = None                  # the decision of an 'auto' run, for logs / the bench line
        if mode == 'auto' and self.shard.dp == 1:
            self._auto = {'it': 0, 'host': [], 'gpu': [], 'ratio': float(os.environ.get('CG_GRAPH_AUTO_RATIO', '0.80'))}
        self._graph_warmup = max(1, int(os.environ.get('CG_GRAPH_WARMUP', '1')))
        # captured graphs kept resident per segment kind (LRU).  Each keeps its private activation pool: 2 (the default) doubles
        # the pools of a kind whose schedule flags alternate (council.flipOnOff) -- README "hipGraph mode"; parsed ONCE, here
        try:
            self._graph_keep = max(1, int(os.environ.get('CG_GRAPH_KEEP', '2')))
        except ValueError:
            raise ValueError("CG_GRAPH_KEEP must be a positive integer, got %r" % os.environ.get('CG_GRAPH_KEEP'))
        # replicas of a member: all-reduce the decoder's gradient bucket under the encoder's backward (CG_DP_OVERLAP=0: one
        # all-reduce of the whole flat gradient after the backward)
        self._dp_overlap = os.environ.get('CG_DP_OVERLAP', '1') != '0'
        # gen_update: the discriminator and the council-discriminator branch over x_fake on the two side streams
        # (CG_GEN_OVERLAP=0: both on the caller's stream)
        self._gen_overlap = os.environ.get('CG_GEN_OVERLAP', '1') != '0'
        # gen_update: the per-layer data-gradient weight mirrors of the new weight versions are prepared on their own stream at the
        # start of the update instead of one by one inside the backward chain (CG_DGRAD_PREFETCH=0: lazily, as before; eager mode only)
        self._dgrad_prefetch = os.environ.get('CG_DGRAD_PREFETCH', '1') != '0'
        self._prep_stream = torch.cuda.Stream(device=dev)
        self._hin = HostInputs(dev)
        self._segs, self._recording, self._gx = {}, None, {}
        self._iter_serials = {}
        self._iter_eager, self._phase = True, 0
        self._overlap = os.environ.get('CG_OVERLAP_UPDATES', '0' if (self.shard.dp > 1 or self._graph_mode) else '1') != '0'
        if self._graph_mode and self._overlap:
            raise hip.HipError("CG_GRAPH=1 replays each update on one stream: it excludes CG_OVERLAP_UPDATES=1")
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
        if slot is not None and self._graph_mode:
            # graph mode: the batch lives in ONE static NHWC buffer per input slot (captured kernels read its address)
            st = self._gx.get(slot)
            if st is None or st.shape != x.shape:
                if st is not None:
                    self._hin.generation += 1
                st = self._gx[slot] = torch.empty(tuple(x.shape), dtype=torch.float32, device=self._device,
                                                  memory_format=torch.channels_last)
            st.copy_(x, non_blocking=True)
            self._img_cache[slot] = (x, x._version, st)
            return st
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
        key = (id(x), x._version, g)
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
        # the batch is identified by object AND version: in graph mode every batch lands in the same static buffer (copy_ bumps
        # its version), so two discriminator updates on different batches must not share content codes
        key = (id(x), x._version, self._weights_version(d, grp))
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
                # the generators' summed-tap upsample-convolution weights are prepared once per weight version and cached: do it
                # HERE, before the fork -- both side streams decode with them, and a cache entry filled on one side stream
                # would be read by the other without an ordering between the two
                for grp in groups:
                    for d in self._dirs:
                        with ops.members(len(grp)):
                            self._nets('gen', d)[grp[0]].dec.prepare_split_weights()
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

    # ------------------------------------------------------------------------------------
    # hipGraph segments (graphs.py)
    # ------------------------------------------------------------------------------------
    def _effect(self, fn):
        """A host-side effect of an update's device work (optimizer step counts, weight versions, ring positions): applied
        now, and -- when the surrounding body is being captured -- remembered so that every replay repeats it."""
        if self._recording is not None:
            self._recording.append(fn)       # capture runs no kernel: the effect follows the replay (see _run)
        else:
            fn()

    def _run(self, key, body):
        """Run one segment of an update.  Eager mode: body().  Graph mode: `_graph_warmup` eager executions per key (they
        create the cached constants and size the workspaces), then ONE capture, then replays.  `body` returns a dict of
        trainer attributes (the loss lists train.py logs): tensors of a captured body live in the graph's private pool
        and hold the replay's results."""
        if not self._graph_mode:
            out = body()
        else:
            # The segments of one iteration hand tensors to each other (the repeated batch, the content codes and their
            # autograd tape, the translations): a segment may REPLAY -- or be CAPTURED -- only while every earlier segment
            # of this iteration replayed or was captured in it, so that what it reads lives at static addresses of a graph
            # pool.  Once one segment of an iteration runs eagerly (warm-up of a new key, a moved input buffer, an unusual
            # call order), the rest of that iteration runs eagerly too (self._iter_eager, reset by dis_update).
            seg = self._segs.get(key)
            if seg is None:
                # a new (shapes, schedule flags, hyper-parameters) key of this kind: at most CG_GRAPH_KEEP (default 2) segments
                # per kind stay resident (schedule flags that alternate -- council.flipOnOff -- then flip between two captures
                # instead of re-capturing at every flip); beyond that the least recently used one and the activation pool its
                # graph owns are released -- after the device has drained, no replay of it may still be in flight
                same = sorted((k for k in self._segs if k[0] == key[0]), key=lambda k: self._segs[k].used)
                keep = self._graph_keep
                if len(same) >= keep:
                    torch.cuda.synchronize(self._device)
                    for k in same[:len(same) - keep + 1]:
                        old = self._segs.pop(k)
                        old.graph, old.out, old.effects, old.ws = None, None, [], {}
                seg = self._segs[key] = Segment()
            if seg.graph is not None and seg.generation != self._hin.generation:
                seg.graph, seg.warm = None, 0                      # a static input buffer moved: capture again
            if not self._iter_eager and seg.graph is not None and seg.parents != self._iter_serials:
                # the earlier segments of THIS iteration are not the captures this graph was recorded behind (one of them was
                # re-captured or evicted since): the tensors it reads from their pools -- content codes, the repeated batch, the
                # translations -- have moved.  Its warm-up has happened; capture again, behind the current ones.
                torch.cuda.synchronize(self._device)
                seg.graph, seg.out, seg.effects, seg.ws = None, None, [], {}
            self._seg_clock = self.__dict__.get('_seg_clock', 0) + 1
            seg.used = self._seg_clock
            if self._iter_eager or (seg.graph is None and seg.warm < self._graph_warmup):
                seg.warm += 1
                self._iter_eager = True
                out = body()
            else:
                if seg.graph is None:
                    # a captured body must not bake the address of a tensor an EAGER pass allocated and a later refresh
                    # frees (the prepared data-gradient weights cached per weight version): it prepares its own
                    for pool in self._pools.values():
                        if pool.split is not None:
                            pool.split._dgrad = {}
                    cur = torch.cuda.current_stream()
                    cap = self.__dict__.get('_cap_stream')
                    if cap is None:
                        cap = self.__dict__['_cap_stream'] = torch.cuda.Stream(device=self._device)
                    cap.wait_stream(cur)
                    g = torch.cuda.CUDAGraph()
                    self._recording = []
                    # no garbage collection inside the capture: collecting a dead trainer's CUDAGraph there would call
                    # hipGraphDestroy on a capturing thread ("operation not permitted when stream is capturing")
                    gc.collect()
                    gc_was_on = gc.isenabled()
                    gc.disable()
                    failed = None
                    try:
                        # capture_error_mode "thread_local": only the capturing thread is policed.  Under "global" (torch's
                        # default) a potentially-unsafe HIP call from ANY thread invalidates the capture -- and the
                        # collective library's watchdog thread polls hipEventQuery on the image exchange it has just run
                        # scratch buffers of the captured launches belong to THIS graph (hip.capture_workspaces: a buffer from
                        # the process-wide per-stream cache would be baked in and later freed under the graph)
                        seg.ws = {}
                        hip.pin_const_caches()
                        with hip.capture_workspaces(seg.ws), \
                                torch.cuda.graph(g, stream=cap, capture_error_mode=os.environ.get('CG_GRAPH_CAPTURE_MODE', 'thread_local')):
                            seg.out = body()
                        seg.effects = self._recording
                    except Exception as e:      # noqa: BLE001 -- whatever stopped the capture, the eager path below still stands
                        failed = e
                    finally:
                        self._recording = None
                        if gc_was_on:
                            gc.enable()
                    cur.wait_stream(cap)
                    if failed is not None:
                        # A capture launches nothing and applies no host-side effect (they were being recorded), so the update
                        # has not happened yet: leave graph mode for good and run it -- and everything after it -- eagerly.
                        # Same kernels, same numbers; a genuine error of the body shows up again, from the eager run.
                        torch.cuda.set_stream(cur)       # (a capture that died in its begin / end leaves the capture stream current)
                        self._leave_graph_mode(key, failed)
                        out = body()
                        for k, v in out.items():
                            setattr(self, k, v)
                        return
                    seg.graph, seg.generation = g, self._hin.generation
                    self._n_captures = self.__dict__.get('_n_captures', 0) + 1
                    seg.serial, seg.parents = self._n_captures, dict(self._iter_serials)
                seg.graph.replay()
                self._iter_serials[key[0]] = seg.serial
                for fn in seg.effects:
                    fn()
                out = seg.out
        for k, v in out.items():
            setattr(self, k, v)

    def _auto_begin(self):
        """cg_graph 'auto' (sharded ranks): iterations 3 and 4 start from an empty queue and are timed (see cuda())."""
        a = self._auto
        if a is None:
            return
        a['it'] += 1
        a.pop('t0', None)
        if a['it'] in (3, 4):
            torch.cuda.synchronize(self._device)
            a['t0'] = time.perf_counter()

    def _auto_end(self):
        a = self._auto
        if a is None or 't0' not in a:
            return
        t1 = time.perf_counter()
        torch.cuda.synchronize(self._device)
        t2 = time.perf_counter()
        a['host'].append(t1 - a['t0'])
        a['gpu'].append(t2 - a['t0'])
        del a['t0']
        if len(a['host']) >= 2:
            host, gpu = min(a['host']), min(a['gpu'])
            use_graph = host > a['ratio'] * gpu
            self.graph_auto = {'host_ms': round(1e3 * host, 2), 'gpu_ms': round(1e3 * gpu, 2), 'threshold': a['ratio'],
                               'graph': bool(use_graph)}
            self._auto = None
            if use_graph:                      # from the next iteration on: warm-up, capture, replay -- on the caller's stream
                self._graph_mode = True
                self._overlap = False
                self._side = []
                self._e0 = None

    def _leave_graph_mode(self, key, err):
        """A segment could not be captured (a driver / collective-library interaction this build has not met): drop every
        reference to tensors of the abandoned capture and continue eagerly for the rest of the run."""
        warnings.warn("council-gan_amd: hipGraph capture of segment %r failed (%s: %s); continuing WITHOUT graph replay "
                      "(CG_GRAPH=0 selects that from the start)" % (key[0], type(err).__name__, err))
        self._graph_mode = False
        self._iter_eager = True
        seg = self._segs.get(key)
        if seg is not None:
            seg.graph, seg.out, seg.effects = None, None, []
        self._rep_cache.clear()
        self._enc_cache.clear()
        for pool in self._pools.values():
            if pool.split is not None:
                pool.split._dgrad = {}

    def _static(self, name, t):
        """Graph mode, several ranks: the result of an eager collective is copied into a static buffer (captured kernels
        read one address); otherwise `t` itself."""
        if not self._graph_mode or self.shard.world_size == 1:
            return t
        st = self._gx.get(name)
        if st is None or st.shape != t.shape:
            if st is not None:
                self._hin.generation += 1
            st = self._gx[name] = torch.empty(tuple(t.shape), dtype=t.dtype, device=t.device, memory_format=torch.channels_last)
        st.copy_(t)
        return st

    def _stage_hyper(self, kind, groups):
        """{first local member of a group: device [runs, 2] tensor of its next Adam step's scalars} (optim.ParamPool.plan_hyper),
        None entries while a group's runs are unknown (before its first step) or outside graph mode."""
        out = {}
        pool = self._pools[kind]
        for grp in groups:
            k0, g = self.shard.local.index(grp[0]), len(grp)
            h = pool.plan_hyper(k0, g) if self._graph_mode else None
            out[k0] = None if h is None else self._hin.stage('adam/%s/%d' % (kind, k0), h)
        return out

    def _opt_key(self, kind):
        """The optimizer hyper-parameters a captured step bakes into its launches (everything but lr and the step count)."""
        g = self._pools[kind].opts[0].param_groups[0]
        return (tuple(g['betas']), g['eps'], g['weight_decay'])

    def _step(self, pool, k0, g, hyper):
        """Optimizer step of one member group inside a body: launches now, bookkeeping as a replayable effect."""
        if self._graph_mode:
            runs = pool.step(k0, g, lockstep=g > 1, hyper=hyper.get(k0), advance=False)
            if runs is None:
                if self._recording is not None:
                    raise hip.HipError("CG_GRAPH=1: the members of a launch diverged (step counts / hyper-parameters)")
                return
            self._effect(lambda: pool.advance(k0, g, runs))
        else:
            pool.step(k0, g, lockstep=g > 1)

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

    def _sync_grads(self, pool, k0, g, early=()):
        """Full-batch gradient = mean of the member replicas' gradients: one all-reduce of the members' gradient slices -- or,
        when `early` names ranges whose all-reduce was started during the backward (_dec_bucket), the finish of those and one
        all-reduce per remaining piece."""
        if self.shard.dp <= 1:
            return
        lo, hi = k0 * pool.stride, (k0 + g) * pool.stride
        done = sorted((a, b) for a, b, _ in early)
        for _, _, h in early:
            self.shard.replica_mean_end(h)
        pos = lo
        for a, b in done + [(hi, hi)]:
            if a > pos:
                self.shard.replica_mean_(pool.grad[pos:a])
            pos = max(pos, b)

    def _dec_bucket(self, d, i):
        """[start, end) of generator (d, i)'s DECODER parameters inside the generator pool's flat buffers: the bucket whose
        gradient is complete as soon as the content code's gradient exists (every decoder layer's backward has run by then;
        the style MLP, fed by all AdaIN layers, is not part of it)."""
        pool = self._pools['gen']
        gen = self._nets('gen', d)[i]
        dec, mlp = list(gen.dec.parameters()), list(gen.mlp.parameters())
        a, b = (pool.index_of(dec[0]) if dec else None), (pool.index_of(mlp[0]) if mlp else None)
        if a is None or b is None:
            return None                       # a parameter outside the pool (frozen / not optimised): no early bucket
        (k, pi), (k2, pj) = a, b
        offs = pool.opts[k].flat['offs']
        if k != k2 or pj != pi + len(dec):
            return None                       # not laid out [.. decoder | mlp ..]: no early bucket
        return k * pool.stride + offs[pi], k * pool.stride + offs[pj]

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
        self._auto_begin()
        if self._graph_mode:
            self._rep_cache.clear()            # the static input buffers hold a new batch: repeat it again (inside the body)
            self._iter_eager, self._phase = False, 1       # a new iteration starts here (train.py:244-250 call order)
            self._iter_serials = {}                        # kind -> capture serial of the segments replayed so far in it
        ctx, tok = self._side_stream(0, x, groups, prologue=True)
        with ctx:
            # ---- host part: the reference's RNG draws (trainer_council.py:741,744), staged into static device buffers
            s = {}
            if self.do_a2b_conf:
                s['a2b'] = self._style(x_b.size(0))
            if self.do_b2a_conf:
                s['b2a'] = self._style(x_a.size(0))
            s_dev = {(d, g): self._hin.stage('dis/s/%s/%d' % (d, g), s[d].repeat(g, 1, 1, 1) if g > 1 else s[d])
                     for d in self._dirs for g in sorted({len(grp) for grp in groups})}
            hyper = self._stage_hyper('dis', groups)
            key = ('dis', tuple(x[self._dirs[0]].shape), tuple(map(tuple, groups)), float(hp['gan_w']), self._opt_key('dis'))
            with self._fresh_mirrors('gen', 'dis'):
                self._run(key, lambda: self._dis_body(x, tgt, groups, float(hp['gan_w']), s_dev, hyper))
        self._side_done(tok)

    def _dis_body(self, x, tgt, groups, gan_w, s_dev, hyper):
        """Device work of dis_update (everything below the RNG draws): eager, or captured once and replayed."""
        pool = self._pools['dis']
        pool.zero_grad()
        out = {'loss_dis_total_s': [0] * self.council_size}
        for d in self._dirs:
            out['loss_dis_%s_s' % d] = [0] * self.council_size
        self._fork()
        for grp in groups:
            g, lead, k0 = len(grp), grp[0], self.shard.local.index(grp[0])
            with self._on(lead, groups), ops.members(g):
                losses, ups = [], []
                for d in self._dirs:
                    gen = self._nets('gen', d)[lead]
                    xr = self._rep(x[d], g)
                    content = self._content(d, grp, xr, need_grad=False)
                    with torch.no_grad(), self._split_decode(d, lead):
                        x_fake = gen.decode(content, s_dev[(d, g)], xr)
                    # :775-777 -- only the a2b term is scaled by gan_w (reference quirk, kept): the unscaled loss is what
                    # train.py logs, the scale rides on the upstream gradient
                    w = gan_w if d == 'a2b' else 1.0
                    l = self._nets('dis', d)[lead].calc_dis_loss(x_fake, tgt[d]).view(-1)       # one loss per member
                    losses.append(l)
                    ups.append(self._const(w, g))
                    for m, i in enumerate(grp):
                        out['loss_dis_%s_s' % d][i] = l.detach()[m]
                torch.autograd.backward(losses, ups)          # members and directions own disjoint parameters
                for m, i in enumerate(grp):
                    tot = None
                    for d, l in zip(self._dirs, losses):
                        w = gan_w if d == 'a2b' else 1.0
                        t = l.detach()[m] if w == 1.0 else l.detach()[m] * w
                        tot = t if tot is None else tot + t
                    out['loss_dis_total_s'][i] = tot
                ops.wgrad_join()
                self._sync_grads(pool, k0, g)
                self._step(pool, k0, g, hyper)
        self._join()
        return out

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
        if self._graph_mode and self._phase != 1:      # not right behind this iteration's dis_update: nothing static to build on
            self._iter_eager = True
        self._phase = 2
        x = {'a2b': self._img(x_a, 'a'), 'b2a': self._img(x_b, 'b')}
        groups = self._plan_groups(x[self._dirs[0]])
        ctx, tok = self._side_stream(1, x, groups, prologue=False)
        with ctx:
            # ---- host part: style noise (s_a first, :806-809), then every member's colleague picks (:861-868; every rank
            # replays every member's draws) -- the two RNG streams are independent, so drawing the picks up front leaves
            # both sequences as the reference consumes them
            s = {}
            if self.do_b2a_conf:
                s['b2a'] = self._style(x_a.size(0))
            if self.do_a2b_conf:
                s['a2b'] = self._style(x_b.size(0))
            less = c['discriminetro_less_style_by']
            n_rel = c['numberOfCouncil_dis_relative_iteration']
            scale = float(hp['council_w']) / float(n_rel)                      # :878-880
            picks = [self.draw_colleagues(i, self.council_size, n_rel) for i in range(self.council_size)]
            sizes = sorted({len(grp) for grp in groups})
            if less != 0:
                s_dev = {(d, g): self._hin.stage('disc/s/%s/%d' % (d, g), torch.cat((s[d], s[d] * less), 0).repeat(g, 1, 1, 1))
                         for d in self._dirs for g in sizes}
            else:
                s_dev = {(d, g): self._hin.stage('disc/s/%s/%d' % (d, g), s[d].repeat(g, 1, 1, 1) if g > 1 else s[d])
                         for d in self._dirs for g in sizes}
            plans = {}
            for grp in groups:
                g, lead = len(grp), grp[0]
                pk = [[(j, float(picks[i].count(j))) for j in sorted(set(picks[i]))] for i in grp]
                for d in self._dirs:
                    b = x[d].shape[0]
                    idx, idx_in, tgt, wt = MsImageDisCouncil.plan_members(pk, g, b, float(len(picks[lead])), scale)
                    tag = 'disc/%s/%d/' % (d, lead)
                    plans[(d, lead)] = (idx, idx_in, self._hin.stage(tag + 'tgt', torch.tensor(tgt, dtype=torch.float32)),
                                        self._hin.stage(tag + 'wt', torch.tensor(wt, dtype=torch.float32)),
                                        self._hin.stage(tag + 'idx', torch.tensor(idx, dtype=torch.int32)))
            hyper = self._stage_hyper('disc', groups)
            shape = tuple(x[self._dirs[0]].shape)
            gkey = tuple(map(tuple, groups))
            u = len(set(picks[0]))
            with self._fresh_mirrors('gen', 'disc'):
                # segment 1: every member's translation (full style) and comparison image (reduced style)
                self._run(('disc1', shape, gkey, float(less)), lambda: self._disc_body_translate(x, groups, less, s_dev))
                # the ONE cross-member exchange (trainer_council.py:853-856): every member's comparison image, member-major.
                # Never part of a captured segment: a collective (RCCL, or gloo through the host) runs between two graphs.
                x_cmp = {d: self._static('cmp/' + d, self.shard.exchange_flat(self._x_cmp_local[d])) for d in self._dirs}
                # segment 2: the council discriminators on [own | colleagues], backward, Adam
                self._run(('disc2', shape, gkey, u, scale, self._opt_key('disc')),
                          lambda: self._disc_body_update(x, x_cmp, groups, plans, scale, hyper))
        self._side_done(tok)

    def _disc_body_translate(self, x, groups, less, s_dev):
        L = len(self.shard.local)
        x_full = {d: {} for d in self._dirs}                 # group lead -> the group's own translations [g*B]
        x_cmp_local = {}
        for d in self._dirs:
            b = x[d].shape[0]
            x_cmp_local[d] = torch.empty((L * b,) + tuple(x[d].shape[1:]), dtype=torch.float32, device=self._device,
                                         memory_format=torch.channels_last)
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
                            twice = [m * b + r for m in range(g) for _ in range(2) for r in range(b)]
                            both = gen.decode(ops.take_rows(content, None, twice), s_dev[(d, g)], self._rep(x[d], 2 * g))
                            own = [m * 2 * b + r for m in range(g) for r in range(b)]
                            x_full[d][lead] = ops.take_rows(both, None, own)
                            ops.take_rows(both, None, [i + b for i in own], out=x_cmp_local[d][k0 * b:(k0 + g) * b])
                        else:
                            x_full[d][lead] = gen.decode(content, s_dev[(d, g)], xr)
                            ops.take_rows(x_full[d][lead], None, list(range(g * b)), out=x_cmp_local[d][k0 * b:(k0 + g) * b])
        self._join()      # every member's council discriminator reads the OTHER members' images
        return {'_x_full': x_full, '_x_cmp_local': x_cmp_local}

    def _disc_body_update(self, x, x_cmp, groups, plans, scale, hyper):
        pool = self._pools['disc']
        pool.zero_grad()
        x_full = self._x_full
        out = {'loss_dis_council_a2b_s': [0] * self.council_size, 'loss_dis_council_b2a_s': [0] * self.council_size,
               'loss_dis_council_total_s': [0] * self.council_size}
        self._fork()
        for grp in groups:
            g, lead, k0 = len(grp), grp[0], self.shard.local.index(grp[0])
            with self._on(lead, groups), ops.members(g):
                losses = []
                for d in self._dirs:
                    idx, idx_in, tgt_dev, wt_dev, idx_dev = plans[(d, lead)]
                    l = self._nets('disc', d)[lead].calc_dis_loss_planned(x_full[d][lead], x_cmp[d], x[d], idx, idx_in,
                                                                          tgt_dev, wt_dev, idx_dev).view(-1)
                    losses.append(l)
                    for m, i in enumerate(grp):
                        out['loss_dis_council_%s_s' % d][i] = l.detach()[m] / scale
                torch.autograd.backward(losses, [self._const(1.0, g)] * len(losses))
                for m, i in enumerate(grp):
                    tot = None
                    for l in losses:
                        tot = l.detach()[m] if tot is None else tot + l.detach()[m]
                    out['loss_dis_council_total_s'][i] = tot
                ops.wgrad_join()
                self._sync_grads(pool, k0, g)
                self._step(pool, k0, g, hyper)
        self._join()
        return out

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
        if self._graph_mode and self._phase not in (1, 2):
            self._iter_eager = True
        self._phase = 0
        x = {'a2b': self._img(x_a, 'a'), 'b2a': self._img(x_b, 'b')}
        groups = self._plan_groups(x[self._dirs[0]])
        self._e0 = None                    # the generators are about to change: no side stream may start from the old mark
        # ---- host part: style noise, both drawn, s_a first (:284-285); schedules (:323-326, 541-555)
        s_a = self._style(x_a.size(0))
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

        s_dev = {(d, g): self._hin.stage('gen/s/%s/%d' % (d, g), s[d].repeat(g, 1, 1, 1) if g > 1 else s[d])
                 for d in self._dirs for g in sorted({len(grp) for grp in groups})}
        pos_dev = {}
        for grp in groups:                 # loss-matching ring write positions (:518-524, 576-586) of each group's members
            for d in self._dirs:
                pos_dev[(d, grp[0])] = self._hin.stage('gen/pos/%s/%d' % (d, grp[0]), torch.tensor(
                    [self._ring_pos[d][grp[0]], self._ring_pos_c[d][grp[0]]], dtype=torch.int32))
        hyper = self._stage_hyper('gen', groups)
        flags = dict(focus_on=bool(focus_on), council_on=bool(council_on), gan_w=float(hp['gan_w']),
                     council_w=float(hp['council_w']), zo_w=float(hp['mask_zero_or_one_w']), total_w=float(hp['mask_total_w']),
                     tv_w=float(hp['mask_tv_w']), center=float(fl['mask_zero_or_one_center']),
                     eps=float(fl['mask_zero_or_one_epsilon']), use_abs=bool(fl['mask_small_use_abs']),
                     use_square=bool(fl['mask_small_use_square']), match=bool(self.do_w_loss_matching))
        key = ('gen', tuple(x[self._dirs[0]].shape), tuple(map(tuple, groups)), tuple(sorted(flags.items())), self._opt_key('gen'))
        with self._fresh_mirrors('gen', 'dis', 'disc'):
            self._run(key, lambda: self._gen_body(x, groups, flags, s_dev, pos_dev, hyper))
        self._auto_end()

    def _gen_body(self, x, groups, f, s_dev, pos_dev, hyper):
        lib = hip.load()
        pool = self._pools['gen']
        pool.zero_grad()
        C = self.council_size
        ab = {'a2b': 'ab', 'b2a': 'ba'}
        out = {'loss_gen_total_s': [0] * C}
        for d in self._dirs:
            out['loss_gen_adv_%s_s' % d] = [0] * C
            out['loss_gen_mask_zero_one_%s_s' % ab[d]] = [0] * C if f['focus_on'] and f['zo_w'] != 0 else []
            out['loss_gen_mask_total_%s_s' % ab[d]] = [0] * C
            out['loss_gen_mask_TV_%s_s' % ab[d]] = [0] * C
            out['council_loss_%s_s' % ab[d]] = [0] * C
        for d in ('a2b', 'b2a'):
            if d not in self._dirs:
                out['loss_gen_adv_%s_s' % d] = [0] * C

        frozen = []
        for i in self.shard.local:       # no weight gradients for D / council-D in this update
            for d in self._dirs:
                for kind in ('dis',) + (('disc',) if self.do_dis_council else ()):
                    for p in self._nets(kind, d)[i].parameters():
                        if p.requires_grad:
                            p.requires_grad_(False)
                            frozen.append(p)
        n_ring = self._ring_n
        # the data-gradient weight mirrors this update's backward will ask for (new generator / discriminator versions since the
        # last gen_update): prepared now, on their own stream, beside the forward pass (ops.SplitWeights.prefetch_dgrad)
        prep_ev = None
        if self._dgrad_prefetch and not self._graph_mode:      # (a second branch inside a hipGraph costs more than it hides: 8.90 vs 8.42 ms at 128x128)
            cur = torch.cuda.current_stream()
            self._prep_stream.wait_stream(cur)
            built = sum(self._pools[k].split.prefetch_dgrad(self._prep_stream) for k in ('gen', 'dis', 'disc')
                        if k in self._pools and self._pools[k].split is not None)
            prep_ev = torch.cuda.Event()
            prep_ev.record(self._prep_stream)
            if not built:
                cur.wait_event(prep_ev)          # nothing forked (first iteration): close the branch at once
                prep_ev = None
        self._fork()
        try:
            for grp in groups:
                g, lead, k0 = len(grp), grp[0], self.shard.local.index(grp[0])
                with self._on(lead, groups), ops.members(g):
                    roots, ups, totals, early = [], [], [], []
                    for d in self._dirs:
                        gen = self._nets('gen', d)[lead]
                        xr = self._rep(x[d], g)
                        content_in = self._content(d, grp, xr, need_grad=True)
                        if self.shard.dp > 1 and self._dp_overlap and g == 1:
                            # replicas: the decoder's share of the flat gradient (about half of it) is complete when the content
                            # code's gradient arrives -- start its all-reduce there, under the encoder's backward
                            rng = self._dec_bucket(d, lead)
                            if rng is not None:
                                def _start(grad, rng=rng):
                                    # the decoder's weight gradients may still be accumulating on the companion stream
                                    # (CG_WGRAD_STREAM=1): the in-place all-reduce of their bucket starts behind them
                                    ops.wgrad_join()
                                    early.append(rng + (self.shard.replica_mean_begin(pool.grad[rng[0]:rng[1]]),))
                                    return grad
                                content_in.register_hook(_start)
                        x_fake = gen.decode(content_in, s_dev[(d, g)], xr)
                        mask = gen.dec.mask_s
                        ftot = adv = lc = w_dev = None
                        if f['focus_on']:                                                   # :390-451
                            ftot, parts = ops.focus_loss(mask, f['center'], f['eps'], f['zo_w'], f['total_w'], f['tv_w'],
                                                         f['use_abs'], f['use_square'],
                                                         reduce=self.shard.replica_mean_ if self.shard.dp > 1 else None)
                            roots.append(ftot.view(-1))
                            ups.append(self._const(1.0, g))
                            parts = parts.view(g, 3)
                            for m, i in enumerate(grp):
                                if f['zo_w'] != 0:
                                    out['loss_gen_mask_zero_one_%s_s' % ab[d]][i] = parts[m, 0]
                                if f['total_w'] != 0:
                                    out['loss_gen_mask_total_%s_s' % ab[d]][i] = parts[m, 1]
                                if f['tv_w'] != 0:
                                    out['loss_gen_mask_TV_%s_s' % ab[d]][i] = parts[m, 2]
                        ring_g, ring_c, w_all = self._rings[d]
                        rg = ring_g[k0 * n_ring:(k0 + g) * n_ring]
                        rc = ring_c[k0 * n_ring:(k0 + g) * n_ring]
                        pd = pos_dev[(d, lead)]
                        # The adversarial and the council term read the same x_fake through two different discriminators:
                        # their forward passes go to the two side streams (autograd runs each node's backward on its forward's
                        # stream, so the backward passes follow), and the HBM-bound passes of one branch -- splits, activation
                        # backward -- run under the convolutions of the other, as in the two discriminator-side updates.
                        fork = (self._gen_overlap and bool(self._side) and not self._graph_mode and f['gan_w'] != 0
                                and f['council_on'])
                        if fork:
                            main = torch.cuda.current_stream()
                            ev = torch.cuda.Event()
                            ev.record(main)
                            for st in self._side:
                                st.wait_event(ev)
                                x_fake.record_stream(st)
                            xr.record_stream(self._side[1])
                        if f['gan_w'] != 0:                                           # :498-529
                            with (torch.cuda.stream(self._side[0]) if fork else contextlib.nullcontext()):
                                adv = self._nets('dis', d)[lead].calc_gen_loss(x_fake).view(-1)
                            if fork:
                                # same creation order as on one stream (adversarial term first): autograd then adds the two
                                # branches' gradients of x_fake in the same order, bit for bit
                                with torch.cuda.stream(self._side[1]):
                                    lc = self._nets('disc', d)[lead].calc_gen_loss(x_fake, xr).view(-1)
                                adv.record_stream(main)
                                lc.record_stream(main)
                                main.wait_stream(self._side[0])
                                main.wait_stream(self._side[1])
                            adv_full = self._full_batch(adv)
                            for m, i in enumerate(grp):
                                out['loss_gen_adv_%s_s' % d][i] = adv_full[m]
                            if f['match']:
                                check(lib.cg_ring_push_dev(ptr(rg), n_ring, ptr(pd[0:1]), ptr(adv_full), g, stream()), "cg_ring_push")
                                self._effect(lambda d=d, grp=grp: [self._ring_pos[d].__setitem__(i, self._ring_pos[d][i] + 1)
                                                                   for i in grp])
                            roots.append(adv)
                            ups.append(self._const(f['gan_w'], g))
                        if f['council_on']:                                                 # :558-624
                            if not fork:
                                lc = self._nets('disc', d)[lead].calc_gen_loss(x_fake, xr).view(-1)
                            if f['match']:
                                w_dev = w_all[k0:k0 + g]
                                check(lib.cg_loss_match_dev(ptr(rg), ptr(rc), n_ring, ptr(pd[1:2]), ptr(self._full_batch(lc)),
                                                            ptr(w_dev), g, stream()), "cg_loss_match")
                                self._effect(lambda d=d, grp=grp: [self._ring_pos_c[d].__setitem__(i, self._ring_pos_c[d][i] + 1)
                                                                   for i in grp])
                                out['w_match_%s_conf' % d] = w_dev[0]
                        # per-member objective + the upstream gradient of the council term (council_w x matching weight)
                        total, council, gcouncil = ops.gen_total(ftot, adv.detach() if adv is not None else None,
                                                                 lc.detach() if lc is not None else None, w_dev,
                                                                 f['gan_w'], f['council_w'], g)
                        if f['council_on']:
                            roots.append(lc)
                            ups.append(gcouncil)
                            for m, i in enumerate(grp):
                                out['council_loss_%s_s' % ab[d]][i] = council[m]
                        totals.append(total)
                    for m, i in enumerate(grp):
                        tot = None
                        for t in totals:
                            tot = t[m] if tot is None else tot + t[m]
                        out['loss_gen_total_s'][i] = tot
                    if prep_ev is not None:
                        torch.cuda.current_stream().wait_event(prep_ev)
                    torch.autograd.backward(roots, ups)
                    ops.wgrad_join()
                    self._sync_grads(pool, k0, g, early)
                    self._step(pool, k0, g, hyper)
        finally:
            self._join()
            for p in frozen:
                p.requires_grad_(True)
        return out

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
