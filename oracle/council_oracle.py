"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the Council-GAN hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this file, and only as the *checker* (or the timed CPU baseline).  The product path
(`council_gan_amd`) never imports it and fails loudly when the HIP library is missing.

What this is: a from-scratch, functional (state_dict-in, tensors-out) restatement of
`Council_Trainer.dis_update / dis_council_update / gen_update` and of the networks they
drive (`AdaINGen`, `MsImageDis`, `MsImageDisCouncil`), written against plain
`torch.nn.functional` on CPU in whatever dtype the state_dict carries (fp32 for parity,
fp64 for the gradient noise floor).  Every function cites the reference
file:line it follows.

Parity pinning: the arithmetic of the reference lives in PyTorch/ATen (third-party, pinned
`pytorch=1.5.0`, conda_requirements.yml:81) and the reference ships no tests or golden
vectors.  This oracle is therefore pinned against outputs of the reference *itself*, run in
the build container through `oracle/ref_shim.py`; the generating script is
`oracle/make_golden.py`, the fixtures are `tests/golden/*.npz`, and
`tests/test_oracle_golden.py` checks this file against every one of them.

Scope (matches DESIGN.md "scope"): everything reachable from the three shipped configs --
LSGAN, zero padding, relu/lrelu/tanh, norm in {none, in, adain} (+ the `ln` LayerNorm
operator), one- or two-directional training with recon/cyc/vgg/abs/council_abs weights 0.
Unreachable reference branches raise NotImplementedError here too.
"""
import math
import random
from collections import deque

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------
def _act(y, activ):
    """networks.py:494-507 (Conv2dBlock activation table; reachable subset)."""
    if activ == 'relu':
        return F.relu(y)
    if activ == 'lrelu':
        return F.leaky_relu(y, 0.2)
    if activ == 'tanh':
        return torch.tanh(y)
    if activ in ('none', None):
        return y
    raise NotImplementedError("activation %r is not reachable from the shipped configs" % activ)


def instance_norm(y, eps=1e-5):
    """networks.py:483 nn.InstanceNorm2d(affine=False): biased variance, eps inside sqrt."""
    return F.instance_norm(y, eps=eps)


def adain(y, weight, bias, eps=1e-5):
    """networks.py:640-653: batch_norm on the view (1, B*C, H, W), training=True.
    weight/bias are flat [B*C] (networks.py:310-311)."""
    b, c, h, w = y.shape
    out = F.batch_norm(y.contiguous().view(1, b * c, h, w), None, None, weight, bias, True, 0.1, eps)
    return out.view(b, c, h, w)


def layer_norm(x, gamma, beta, eps=1e-5):
    """networks.py:670-686: per-sample mean and *unbiased* std over C*H*W, eps added to std."""
    n = x.size(0)
    mean = x.reshape(n, -1).mean(1).view(n, 1, 1, 1)
    std = x.reshape(n, -1).std(1).view(n, 1, 1, 1)
    x = (x - mean) / (std + eps)
    return x * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def conv_block(x, sd, prefix, stride, pad, norm='none', activ='relu', adain_wb=None):
    """networks.py:515-521: ZeroPad2d -> Conv2d(bias) -> norm -> activation."""
    y = F.conv2d(F.pad(x, (pad, pad, pad, pad)), sd[prefix + 'conv.weight'], sd[prefix + 'conv.bias'],
                 stride=stride)
    if norm == 'in':
        y = instance_norm(y)
    elif norm == 'adain':
        y = adain(y, adain_wb[0], adain_wb[1])
    elif norm == 'ln':
        y = layer_norm(y, sd[prefix + 'norm.gamma'], sd[prefix + 'norm.beta'])
    elif norm != 'none':
        raise NotImplementedError("norm %r is not reachable from the shipped configs" % norm)
    return _act(y, activ)


def avgpool3s2(x):
    """networks.py:32,129: AvgPool2d(3, stride=2, padding=1, count_include_pad=False)."""
    return F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)


# --------------------------------------------------------------------------------------
# generator (AdaINGen, networks.py:223-330)
# --------------------------------------------------------------------------------------
class OracleGen:
    def __init__(self, sd, gen_hp):
        self.sd = sd
        self.dim = gen_hp['dim']
        self.n_down = gen_hp['n_downsample']
        self.n_res = gen_hp['n_res']
        self.activ = gen_hp['activ']
        self.nmask = gen_hp['num_of_mask_dim_to_add']
        if gen_hp['do_my_style']:
            raise NotImplementedError("gen.do_my_style is False in every shipped config")
        if gen_hp['pad_type'] != 'zero':
            raise NotImplementedError("only zero padding is reachable")
        self.mask_s = None

    # networks.py:337-353
    def encode_style(self, x):
        sd, a = self.sd, self.activ
        y = conv_block(x, sd, 'enc_style.model.0.', 1, 3, 'none', a)
        for i in range(1, 5):      # 2 widening + (4-2) same-width stride-2 blocks
            y = conv_block(y, sd, 'enc_style.model.%d.' % i, 2, 1, 'none', a)
        y = F.adaptive_avg_pool2d(y, 1)
        return F.conv2d(y, sd['enc_style.model.6.weight'], sd['enc_style.model.6.bias'])

    # networks.py:355-369 + ResBlock networks.py:448-461
    def encode_content(self, x):
        sd, a = self.sd, self.activ
        y = conv_block(x, sd, 'enc_content.model.0.', 1, 3, 'in', a)
        for i in range(self.n_down):
            y = conv_block(y, sd, 'enc_content.model.%d.' % (i + 1), 2, 1, 'in', a)
        rp = 'enc_content.model.%d.model.' % (self.n_down + 1)
        for r in range(self.n_res):
            res = y
            y = conv_block(y, sd, rp + '%d.model.0.' % r, 1, 1, 'in', a)
            y = conv_block(y, sd, rp + '%d.model.1.' % r, 1, 1, 'in', 'none')
            y = y + res
        return y

    # networks.py:278-283
    def encode(self, x):
        return self.encode_content(x), self.encode_style(x)

    def adain_layout(self):
        """Order and widths of the AdaIN layers as `model.modules()` visits them
        (networks.py:303-312): 2 per decoder ResBlock, then 2 per upsampling stage."""
        c = self.dim * (2 ** self.n_down)
        widths = [c] * (2 * self.n_res)
        for _ in range(self.n_down):
            c //= 2
            widths += [c, c]
        return widths

    def mlp(self, style):
        """networks.py:432-443: Linear-activ, (n_blk-2) x Linear-activ, Linear (activ = gen.activ, networks.py:251-254)."""
        sd = self.sd
        h = style.view(style.size(0), -1)
        h = _act(F.linear(h, sd['mlp.model.0.fc.weight'], sd['mlp.model.0.fc.bias']), self.activ)
        h = _act(F.linear(h, sd['mlp.model.1.fc.weight'], sd['mlp.model.1.fc.bias']), self.activ)
        return F.linear(h, sd['mlp.model.2.fc.weight'], sd['mlp.model.2.fc.bias'])

    # networks.py:285-301 + Decoder_V2_atten networks.py:374-415
    def decode(self, content, style, images, return_mask=False):
        sd, a = self.sd, self.activ
        p = self.mlp(style)
        # assign_adain_params, networks.py:303-312: [mean(bias) | std(weight)] per layer
        wb, off = [], 0
        for c in self.adain_layout():
            bias = p[:, off:off + c].contiguous().view(-1)
            weight = p[:, off + c:off + 2 * c].contiguous().view(-1)
            wb.append((weight, bias))
            off += 2 * c
        li = 0
        y = content
        for r in range(self.n_res):
            res = y
            y = conv_block(y, sd, 'dec.model.0.model.%d.model.0.' % r, 1, 1, 'adain', a, wb[li]); li += 1
            y = conv_block(y, sd, 'dec.model.0.model.%d.model.1.' % r, 1, 1, 'adain', 'none', wb[li]); li += 1
            y = y + res
        idx = 1
        for _ in range(self.n_down):
            y = F.interpolate(y, scale_factor=2, mode='nearest')     # networks.py:385
            y = conv_block(y, sd, 'dec.model.%d.' % (idx + 1), 1, 1, 'adain', a, wb[li]); li += 1
            y = conv_block(y, sd, 'dec.model.%d.' % (idx + 2), 1, 1, 'adain', a, wb[li]); li += 1
            idx += 3
        y = conv_block(y, sd, 'dec.model.%d.' % idx, 1, 0, 'none', a)
        y = conv_block(y, sd, 'dec.model.%d.' % (idx + 1), 1, 0, 'none', a)
        new_x = conv_block(y, sd, 'dec.model.%d.' % (idx + 2), 1, 0, 'none', 'tanh')
        # mask/blend head, networks.py:398-407
        k = self.nmask
        out_dim = images.shape[1]
        self.mask_s = (torch.tanh(10 * new_x[:, -k:]) + 1) / 2
        im = images
        for j in range(k):
            m = self.mask_s[:, j:j + 1]
            im = (1 - m) * im + m * new_x[:, out_dim * j:out_dim * (j + 1)]
        if return_mask:
            mask = self.mask_s
            if mask.shape[1] != 3:                                   # networks.py:410-412
                mask = (mask.sum(1, keepdim=True) / mask.shape[1]).repeat(1, 3, 1, 1)
                self.mask_s = mask
            return im, mask
        return im


# --------------------------------------------------------------------------------------
# discriminators (networks.py:17-215)
# --------------------------------------------------------------------------------------
def _lsgan_dis(outs0, outs1):
    """networks.py:62-64 / 164-166."""
    loss = 0
    for o0, o1 in zip(outs0, outs1):
        loss = loss + torch.mean((o0 - 0) ** 2) + torch.mean((o1 - 1) ** 2)
    return loss


def _lsgan_gen(outs0):
    """networks.py:88-90 / 192-194."""
    loss = 0
    for o0 in outs0:
        loss = loss + torch.mean((o0 - 1) ** 2)
    return loss


class OracleDis:
    """MsImageDis, networks.py:17-110."""

    def __init__(self, sd, dis_hp):
        self.sd = sd
        self.hp = dis_hp
        if dis_hp['gan_type'] != 'lsgan':
            raise NotImplementedError("only lsgan is reachable (nsgan unused; the relativistic "
                                      "branch hard-codes batch 10, networks.py:73)")

    def forward(self, x):
        sd, hp = self.sd, self.hp
        outs = []
        for s in range(hp['num_scales']):
            p = 'cnns.%d.' % s
            y = conv_block(x, sd, p + '0.', 2, 1, 'none', hp['activ'])
            for l in range(1, hp['n_layer']):
                y = conv_block(y, sd, p + '%d.' % l, 2, 1, hp['norm'], hp['activ'])
            n = hp['n_layer']
            outs.append(F.conv2d(y, sd[p + '%d.weight' % n], sd[p + '%d.bias' % n]))
            x = avgpool3s2(x)
        return outs

    def calc_dis_loss(self, fake, real):
        return _lsgan_dis(self.forward(fake), self.forward(real))

    def calc_gen_loss(self, fake):
        return _lsgan_gen(self.forward(fake))


class OracleDisCouncil:
    """MsImageDisCouncil, networks.py:116-215 (6-channel conditional input, 3x3 s1 first conv,
    two 1x1 convs with no activation in between)."""

    def __init__(self, sd, dis_hp):
        self.sd = sd
        self.hp = dis_hp
        if dis_hp['gan_type'] != 'lsgan':
            raise NotImplementedError("only lsgan is reachable")

    def forward(self, x, x_input):
        sd, hp = self.sd, self.hp
        outs = []
        for s in range(hp['num_scales']):
            p = 'cnns.%d.' % s
            y = conv_block(torch.cat((x, x_input), 1), sd, p + '0.', 1, 1, 'none', hp['activ'])
            for l in range(1, hp['n_layer']):
                y = conv_block(y, sd, p + '%d.' % l, 2, 1, hp['norm'], hp['activ'])
            n = hp['n_layer']
            y = F.conv2d(y, sd[p + '%d.weight' % n], sd[p + '%d.bias' % n])
            outs.append(F.conv2d(y, sd[p + '%d.weight' % (n + 1)], sd[p + '%d.bias' % (n + 1)]))
            x = avgpool3s2(x)
            x_input = avgpool3s2(x_input)
        return outs

    def calc_dis_loss(self, fake, real, inp):
        return _lsgan_dis(self.forward(fake, inp), self.forward(real, inp))

    def calc_gen_loss(self, fake, inp):
        return _lsgan_gen(self.forward(fake, inp))


# --------------------------------------------------------------------------------------
# focus-loss criteria (trainer_council.py:230-250)
# --------------------------------------------------------------------------------------
def mask_zero_one(mask, center, eps):
    return torch.sum(1 / (torch.abs(mask - center) + eps)) / mask.numel()


def mask_small(mask, use_abs, use_square):
    assert use_abs or use_square
    loss = 0
    if use_abs:
        loss = loss + torch.abs(torch.sum(mask)) / mask.numel()
    if use_square:
        loss = loss + (torch.sum(mask) / mask.numel()) ** 2
    return loss


def mask_tv(mask):
    return (torch.sum(torch.abs(mask[:, :, 1:, :] - mask[:, :, :-1, :])) +
            torch.sum(torch.abs(mask[:, :, :, 1:] - mask[:, :, :, :-1]))) / mask.numel()


# --------------------------------------------------------------------------------------
# schedules (host integers)
# --------------------------------------------------------------------------------------
def council_flip_state(hp):
    """trainer_council.py:541-550 / 787-796: the on/off flip-flop window."""
    c = hp['council']
    cycle = c['flipOnOff_On_iteration'] + c['flipOnOff_Off_iteration']
    cur = hp['iteration'] % cycle
    start = c['flipOnOff_On_iteration'] if c['flipOnOff_start_with'] else c['flipOnOff_Off_iteration']
    return c['flipOnOff_start_with'] if cur < start else (not c['flipOnOff_start_with'])


def dis_council_active(hp, council_size):
    """trainer_council.py:784-801."""
    c = hp['council']
    if council_size <= 1 or c['numberOfCouncil_dis_relative_iteration'] == 0:
        return False
    do = council_flip_state(hp)
    if not c['flipOnOff']:
        do = c['flipOnOff_start_with']          # :797-798 (sic)
    return bool(do) and hp['council_w'] != 0 and hp['iteration'] >= c['council_start_at_iter']


def gen_council_active(hp, council_size):
    """trainer_council.py:549-559."""
    c = hp['council']
    do = council_flip_state(hp)
    if not c['flipOnOff']:
        do = True
    if hp['iteration'] < c['council_start_at_iter']:
        do = False
    return (hp['council_w'] != 0 or hp['council_abs_w'] != 0) and bool(do) and council_size > 1


def draw_colleagues(i, council_size, n_rel):
    """trainer_council.py:861-868: colleague picks for member i (Python global RNG)."""
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


def check_supported(hp):
    for k in ('recon_x_w', 'recon_s_w', 'recon_c_w', 'recon_x_cyc_w', 'vgg_w', 'abs_beginning_end',
              'council_abs_w'):
        if hp.get(k, 0) != 0:
            raise NotImplementedError("%s != 0 is outside the shipped-config hot path (SURVEY 8a R2)" % k)
    if hp['gen']['useRandomDis'] or hp['dis']['useRandomGen'] or hp['dis']['do_Dis_only_gray']:
        raise NotImplementedError("useRandomDis/useRandomGen/do_Dis_only_gray are False in every shipped config")
    if hp['focus_loss']['do_w_loss_matching_focus']:
        raise NotImplementedError("do_w_loss_matching_focus is False in every shipped config")
    if not (hp['do_a2b'] or hp['do_b2a']):
        raise ValueError("at least one of do_a2b / do_b2a")


# --------------------------------------------------------------------------------------
# trainer (Council_Trainer, trainer_council.py:20-883) -- update steps only
# --------------------------------------------------------------------------------------
class OracleTrainer:
    """State: per member and direction a state_dict of leaf tensors for gen / dis / dis_council,
    three torch.optim.Adam per member (trainer_council.py:170-179), loss-matching deques
    (:71-92).  `state` = {'a2b': {'gen': [sd...], 'dis': [...], 'dis_council': [...]}, 'b2a': ...}."""

    def __init__(self, hp, state, dtype=torch.float32):
        check_supported(hp)
        self.hp = hp
        self.C = hp['council']['council_size']
        self.dirs = [d for d in ('a2b', 'b2a') if hp['do_' + d]]
        self.do_dis_council = hp['council_w'] != 0
        self.style_dim = hp['gen']['style_dim']
        self.dtype = dtype
        self.sd = {}
        for d in self.dirs:
            self.sd[d] = {}
            for net in ('gen', 'dis') + (('dis_council',) if self.do_dis_council else ()):
                self.sd[d][net] = []
                for i in range(self.C):
                    leaf = {}
                    for k, v in state[d][net][i].items():
                        t = torch.as_tensor(np.asarray(v)).to(dtype).clone()
                        is_param = not (k.endswith('running_mean') or k.endswith('running_var'))
                        leaf[k] = t.requires_grad_(is_param)
                    self.sd[d][net].append(leaf)
        lr, betas, wd = hp['lr'], (hp['beta1'], hp['beta2']), hp['weight_decay']

        def params(net, i):
            ps = []
            for d in self.dirs:                       # :156-165: a2b params first, then b2a
                ps += [t for t in self.sd[d][net][i].values() if t.requires_grad]
            return ps
        self.dis_opt = [torch.optim.Adam(params('dis', i), lr=lr, betas=betas, weight_decay=wd) for i in range(self.C)]
        self.gen_opt = [torch.optim.Adam(params('gen', i), lr=lr, betas=betas, weight_decay=wd) for i in range(self.C)]
        if self.do_dis_council:
            self.disc_opt = [torch.optim.Adam(params('dis_council', i), lr=lr, betas=betas, weight_decay=wd)
                             for i in range(self.C)]
        n = hp['loss_matching_hist_size']
        self.hist_gan = {d: [deque(np.ones(n)) for _ in range(self.C)] for d in self.dirs}
        self.hist_council = {d: [deque(np.ones(n)) for _ in range(self.C)] for d in self.dirs}
        self.out = {}

    # -- helpers --------------------------------------------------------------------
    def gen(self, d, i):
        return OracleGen(self.sd[d]['gen'][i], self.hp['gen'])

    def dis(self, d, i):
        return OracleDis(self.sd[d]['dis'][i], self.hp['dis'])

    def disc(self, d, i):
        return OracleDisCouncil(self.sd[d]['dis_council'][i], self.hp['dis'])

    def _src(self, d, x_a, x_b):
        return x_a if d == 'a2b' else x_b

    def _dst(self, d, x_a, x_b):
        return x_b if d == 'a2b' else x_a

    # -- dis_update, trainer_council.py:735-780 ---------------------------------------
    def dis_update(self, x_a, x_b, hp):
        x_a, x_b = x_a.to(self.dtype), x_b.to(self.dtype)
        for o in self.dis_opt:
            o.zero_grad()
        s = {}
        if 'a2b' in self.dirs:                        # :740-741 (s_b first)
            s['a2b'] = torch.randn(x_b.size(0), self.style_dim, 1, 1).to(self.dtype)
        if 'b2a' in self.dirs:                        # :743-744
            s['b2a'] = torch.randn(x_a.size(0), self.style_dim, 1, 1).to(self.dtype)
        self.loss_dis = {d: [] for d in self.dirs}
        self.loss_dis_total = []
        for i in range(self.C):
            total = 0
            for d in self.dirs:
                src, dst = self._src(d, x_a, x_b), self._dst(d, x_a, x_b)
                g = self.gen(d, i)
                with torch.no_grad():                  # :754-760 (graph built then detached at :769)
                    x_fake = g.decode(g.encode_content(src), s[d], src)
                l = self.dis(d, i).calc_dis_loss(x_fake.detach(), dst)
                self.loss_dis[d].append(l)
                # :775-777 -- the b2a term is NOT scaled by gan_w (reference quirk, kept)
                total = total + (hp['gan_w'] * l if d == 'a2b' else l)
            self.loss_dis_total.append(total)
            total.backward()
            self.dis_opt[i].step()

    # -- dis_council_update, trainer_council.py:782-883 --------------------------------
    def dis_council_update(self, x_a, x_b, hp):
        if not dis_council_active(hp, self.C):
            return False
        x_a, x_b = x_a.to(self.dtype), x_b.to(self.dtype)
        for o in self.disc_opt:
            o.zero_grad()
        s = {}
        if 'b2a' in self.dirs:                        # :806-807 (s_a first here)
            s['b2a'] = torch.randn(x_a.size(0), self.style_dim, 1, 1).to(self.dtype)
        if 'a2b' in self.dirs:                        # :808-809
            s['a2b'] = torch.randn(x_b.size(0), self.style_dim, 1, 1).to(self.dtype)
        less = hp['council']['discriminetro_less_style_by']
        n_rel = hp['council']['numberOfCouncil_dis_relative_iteration']
        x_full = {d: [] for d in self.dirs}
        x_cmp = {d: [] for d in self.dirs}
        with torch.no_grad():
            for i in range(self.C):                   # :826-851
                for d in self.dirs:
                    src = self._src(d, x_a, x_b)
                    g = self.gen(d, i)
                    c = g.encode_content(src)
                    x_full[d].append(g.decode(c, s[d], src))
                    x_cmp[d].append(g.decode(c, s[d] * less, src) if less != 0 else x_full[d][-1])
        self.x_council_full, self.x_council_cmp = x_full, x_cmp
        self.loss_disc = {d: [] for d in self.dirs}
        self.loss_disc_total = []
        self.council_picks = []
        for i in range(self.C):                       # :858-883
            picks = draw_colleagues(i, self.C, n_rel)
            self.council_picks.append(picks)
            total = 0
            for d in self.dirs:
                src = self._src(d, x_a, x_b)
                dc = self.disc(d, i)
                l = 0
                for j in picks:                       # :872-874
                    l = l + dc.calc_dis_loss(x_full[d][i].detach(), x_cmp[d][j].detach(), src)
                self.loss_disc[d].append(l)
                total = total + hp['council_w'] * l / n_rel     # :878-880
            self.loss_disc_total.append(total)
            total.backward()
            self.disc_opt[i].step()
        return True

    # -- gen_update, trainer_council.py:280-634 ----------------------------------------
    def gen_update(self, x_a, x_b, hp, iterations=0):
        x_a, x_b = x_a.to(self.dtype), x_b.to(self.dtype)
        for o in self.gen_opt:
            o.zero_grad()
        # :284-285 -- both drawn, s_a then s_b, whatever the direction
        s_a = torch.randn(x_a.size(0), self.style_dim, 1, 1).to(self.dtype)
        s_b = torch.randn(x_b.size(0), self.style_dim, 1, 1).to(self.dtype)
        s = {'a2b': s_b, 'b2a': s_a}
        fl = hp['focus_loss']
        focus_on = hp['iteration'] > fl['focus_loss_start_at_iter'] and \
            (hp['mask_zero_or_one_w'] != 0 or hp['mask_total_w'] != 0)       # :390
        self.x_fake = {d: [] for d in self.dirs}
        self.mask = {d: [] for d in self.dirs}
        self.loss_gen_total = []
        self.loss_gen_adv = {d: [] for d in self.dirs}
        self.loss_mask_zero_one = {d: [] for d in self.dirs}
        self.loss_mask_total = {d: [] for d in self.dirs}
        self.loss_mask_tv = {d: [] for d in self.dirs}
        for i in range(self.C):                       # :328-538
            total = 0
            for d in self.dirs:
                src = self._src(d, x_a, x_b)
                g = self.gen(d, i)
                x = g.decode(g.encode_content(src), s[d], src)
                self.x_fake[d].append(x)
                self.mask[d].append(g.mask_s)
            for d in self.dirs:                                            # :381-387
                self.loss_mask_tv[d].append(0)
                self.loss_mask_total[d].append(0)
            if focus_on:
                for d in self.dirs:
                    if hp['mask_zero_or_one_w'] != 0:                      # :392-415
                        l = mask_zero_one(self.mask[d][i], fl['mask_zero_or_one_center'],
                                          fl['mask_zero_or_one_epsilon'])
                        self.loss_mask_zero_one[d].append(l)
                        total = total + hp['mask_zero_or_one_w'] * l
                for d in self.dirs:
                    if hp['mask_total_w'] != 0:                            # :418-422
                        self.loss_mask_total[d][i] = self.loss_mask_total[d][i] + mask_small(
                            self.mask[d][i], fl['mask_small_use_abs'], fl['mask_small_use_square'])
                for d in self.dirs:
                    if hp['mask_tv_w'] != 0:                               # :425-431
                        self.loss_mask_tv[d][i] = self.loss_mask_tv[d][i] + mask_tv(self.mask[d][i])
                        total = total + hp['mask_tv_w'] * self.loss_mask_tv[d][i]
                for d in self.dirs:                                        # :447-451
                    total = total + hp['mask_total_w'] * self.loss_mask_total[d][i]
            if hp['gan_w'] != 0:                                           # :498-529
                for d in self.dirs:
                    l = self.dis(d, i).calc_gen_loss(self.x_fake[d][i])
                    self.loss_gen_adv[d].append(l)
                    if hp['do_w_loss_matching']:                           # :518-524
                        self.hist_gan[d][i].append(l.detach().cpu().numpy())
                        self.hist_gan[d][i].popleft()
                for d in self.dirs:
                    total = total + hp['gan_w'] * self.loss_gen_adv[d][i]
            self.loss_gen_total.append(total)
        do_council = gen_council_active(hp, self.C)
        self.council_loss = {d: [] for d in self.dirs}
        self.w_match = {d: [] for d in self.dirs}
        for i in range(self.C):                       # :558-634
            if do_council and self.do_dis_council:
                for d in self.dirs:
                    src = self._src(d, x_a, x_b)
                    l = self.disc(d, i).calc_gen_loss(self.x_fake[d][i], src)      # :569-573
                    w = 1.0
                    if hp['do_w_loss_matching']:                           # :576-586
                        self.hist_council[d][i].append(l.detach().cpu().numpy())
                        self.hist_council[d][i].popleft()
                        w = np.mean(self.hist_gan[d][i]) / np.mean(self.hist_council[d][i])
                        l = l * w
                    self.w_match[d].append(w)
                    l = l * hp['council_w']                                # :588-593
                    self.council_loss[d].append(l)
                    self.loss_gen_total[i] = self.loss_gen_total[i] + l    # :621-624
            else:
                for d in self.dirs:
                    self.council_loss[d].append(0)
            self.loss_gen_total[i].backward()
            self.gen_opt[i].step()


    # -- sample, trainer_council.py:643-733 ---------------------------------------------
    @torch.no_grad()
    def sample(self, x_a, x_b, s_a_fixed, s_b_fixed, members=None, return_mask=True):
        """The reference's 8-tuple: per direction (inputs repeated per member, masks | reconstructions, translation with
        the fixed display style, translation with a fresh style).  Fresh styles: s_b2 then s_a2 (:648, :654)."""
        members = range(self.C) if members is None else members
        fresh = {}
        if 'a2b' in self.dirs:
            fresh['a2b'] = torch.randn(x_a.size(0), self.style_dim, 1, 1).to(self.dtype)
        if 'b2a' in self.dirs:
            fresh['b2a'] = torch.randn(x_b.size(0), self.style_dim, 1, 1).to(self.dtype)
        out = {}
        for d, xin, s1 in (('a2b', x_a, s_b_fixed), ('b2a', x_b, s_a_fixed)):
            if d not in self.dirs:
                continue
            xin, s1 = xin.to(self.dtype), s1.to(self.dtype)
            xs, second, x1, x2 = [], [], [], []
            for n in range(xin.size(0)):
                for j in members:
                    g = self.gen(d, j)
                    xi = xin[n:n + 1]
                    xs.append(xi)
                    c, s_fake = g.encode(xi)                                    # :660, :671
                    if not return_mask:
                        second.append(g.decode(c, s_fake, xi))                  # :662, :673
                        x1.append(g.decode(c, s1[n:n + 1], xi))
                    else:
                        im, m = g.decode(c, s1[n:n + 1], xi, return_mask=True)  # :666, :677
                        x1.append(im)
                        second.append(m)
                    x2.append(g.decode(c, fresh[d][n:n + 1], xi))
            out[d] = (torch.cat(xs), torch.cat(second), torch.cat(x1), torch.cat(x2))
        none4 = (None, None, None, None)
        return out.get('a2b', none4) + out.get('b2a', none4)

    # -- resume, trainer_council.py:898-967 ----------------------------------------------
    def resume(self, checkpoint_dir):
        """Weights and Adam state from a checkpoint set in the reference's layout (`{d}_{gen|dis|dis_council}_{i}_{it:08d}.pt`
        holding {d: state_dict}, `optimizer_{i}.pt` holding {'gen','dis','dis_council'}); returns the iteration."""
        import os
        names = {'gen': 'gen', 'dis': 'dis', 'dis_council': 'dis_council'}
        iterations = 0
        for i in range(self.C):
            for net in ('gen', 'dis') + (('dis_council',) if self.do_dis_council else ()):
                for d in self.dirs:
                    key = '%s_%s_%d_' % (d, names[net], i)
                    files = sorted(f for f in os.listdir(checkpoint_dir) if f.startswith(key) and f.endswith('.pt'))
                    sd = torch.load(os.path.join(checkpoint_dir, files[-1]), map_location='cpu')[d]
                    with torch.no_grad():
                        for k, t in self.sd[d][net][i].items():
                            t.copy_(sd[k].to(self.dtype))
                    if net == 'gen':
                        iterations = int(files[-1][-11:-3])
            osd = torch.load(os.path.join(checkpoint_dir, 'optimizer_%d.pt' % i), map_location='cpu')
            self.gen_opt[i].load_state_dict(osd['gen'])
            self.dis_opt[i].load_state_dict(osd['dis'])
            if self.do_dis_council:
                self.disc_opt[i].load_state_dict(osd['dis_council'])
        return iterations


# --------------------------------------------------------------------------------------
# small utilities shared by tests / bench
# --------------------------------------------------------------------------------------
def to_numpy_state(module_state_dict):
    return {k: v.detach().cpu().numpy() for k, v in module_state_dict.items()}


def synthetic_batch(batch, size, seed=7):
    """fp32 images in [-1, 1) -- the range Normalize(0.5, 0.5) produces (utils.py:124-126)."""
    g = torch.Generator().manual_seed(seed)
    x_a = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    x_b = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    return x_a, x_b


def seed_all(seed=1):
    """train.py:55-62."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
