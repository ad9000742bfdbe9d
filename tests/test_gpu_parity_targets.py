"""Parity on the configurations the targets are quoted on (round-3 additions):

  * BASELINE.json configs[4]'s problem -- anime2face, council 8 -- one whole iteration against the oracle: two member
    groups per launch (CG_GROUP = 4), two member streams, 4-of-7 colleague picks without refill, the [own | colleagues]
    council batch of 5 B samples per member (trainer_council.py:858-868);
  * the generator's BACKWARD CHAIN alone under a smooth loss: fixed random upstream gradients for (image, mask) instead of
    the LSGAN / focus heads, at full width, member-batched, on the benchmarked (split-precision) datapath -- every
    generator tensor against the fp64 oracle, tensor by tensor;
  * member grouping on the exact-fp32 datapath: CG_GROUP = 1 (member by member) against CG_GROUP = 4 (one launch per
    layer) must give the same generator / discriminator gradients to round-off -- a wrong upstream scale or a member
    mix-up in the batched back-propagation cannot hide behind the chaotic generator-gradient comparison.
Criteria for whole iterations: tests/parity_util.py."""
import copy
import os
import random

import numpy as np
import pytest
import torch
import yaml

import parity_util as P
from oracle import council_oracle as O

pytestmark = pytest.mark.gpu
CONFIGS = os.path.join(os.path.dirname(__file__), "..", "configs")


@pytest.fixture(scope="module")
def cga():
    import council_gan_amd
    council_gan_amd.hip.load()
    return council_gan_amd


def _cfg(name, council, iteration=60000):
    cfg = yaml.safe_load(open(os.path.join(CONFIGS, name)))
    cfg['council']['council_size'] = council
    cfg['iteration'] = iteration
    return cfg


def test_council8_iteration_vs_oracle(cga):
    """anime2face (b2a, three mask channels), council 8, full widths, 64x64, batch 1."""
    cfg = _cfg("anime2face_council_folder.yaml", 8)
    assert cfg['council']['numberOfCouncil_dis_relative_iteration'] == 4       # 4 of 7 colleagues, no refill
    random.seed(5)
    picks = cga.Council_Trainer.draw_colleagues(2, 8, 4)
    assert len(picks) == len(set(picks)) == 4 and 2 not in picks
    seen = {}
    orig = cga.Council_Trainer._plan_groups

    def spy(self, x):
        groups = orig(self, x)
        seen['groups'] = [list(g) for g in groups]
        seen['streams'] = len(self._streams)
        return groups
    cga.Council_Trainer._plan_groups = spy
    try:
        P.iteration_vs_oracle(cga, cfg, 64, 1, seed=6, report="anime2face 64^2 council 8 B1")
    finally:
        cga.Council_Trainer._plan_groups = orig
    if int(os.environ.get('CG_GROUP', '4')) == 4:
        assert seen['groups'] == [[0, 1, 2, 3], [4, 5, 6, 7]], seen      # two launches of four members


def test_generator_backward_chain_smooth_loss(cga):
    """Every generator tensor's gradient against the fp64 oracle under a linear objective  sum(G_img * image) +
    sum(G_mask * mask)  with fixed random G: no LSGAN / focus head, so no chaotic amplification -- what is compared is the
    backward chain itself (mask/blend head, 1x1 head, upsampling convolutions, AdaIN + MLP, residual blocks, strided
    convolutions, the 7x7 first layer), member-batched (two members in one launch) on the split-precision datapath.
    The tanh head is kept out of saturation (the last layer's weights are scaled by 0.02).  With the saturated head of a
    freshly initialised generator the upstream gradient field is carried by the few unsaturated pixels, and the ReLU sign
    decisions that differ between ANY two fp32 evaluations (|y| ~ 1e-6 elements of the 1x1 head at full resolution) then
    move every upstream tensor by the same ~2e-3 -- measured on the reference's own arithmetic (CPU oracle fp32 vs fp64:
    2.2e-3 on all 60 tensors; an exact tanh derivative does not change it) -- which would drown the comparison; out of
    saturation that floor is 1.5e-4."""
    from council_gan_amd import ops
    cfg = _cfg("male2female_council_folder.yaml", 2)
    cfg['batch_size'] = 2
    B, S, C = 2, 64, 2
    O.seed_all(21)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    assert tr._split_fwd
    with torch.no_grad():
        for gen in tr.gen_a2b_s:
            gen.dec.model[9].conv.weight.mul_(0.02)
    state = P.host_state(tr)
    tr.cuda('cuda:0')
    tr._hp_last = cfg
    tr._ready()
    x_a, _ = O.synthetic_batch(B, S)
    g = torch.Generator().manual_seed(5)
    style = torch.randn(B, cfg['gen']['style_dim'], 1, 1, generator=g)
    up_im = torch.randn(C, B, 3, S, S, generator=g)
    up_mask = torch.randn(C, B, 3, S, S, generator=g)

    x = tr._img(x_a, 'a')
    groups = tr._plan_groups(x)
    assert [list(grp) for grp in groups] == [[0, 1]]
    pool = tr._pools['gen']
    pool.zero_grad()
    with tr._fresh_mirrors('gen'), ops.members(C):
        xr = tr._rep(x, C)
        gen = tr.gen_a2b_s[0]
        s_dev = style.repeat(C, 1, 1, 1).cuda()
        fake = gen.decode(tr._content('a2b', groups[0], xr, need_grad=True), s_dev, xr)
        mask = gen.dec.mask_s
        assert fake.shape == (C * B, 3, S, S) and mask.shape == (C * B, 3, S, S)
        cl = torch.channels_last
        torch.autograd.backward([fake, mask], [up_im.view(C * B, 3, S, S).cuda().contiguous(memory_format=cl),
                                               up_mask.view(C * B, 3, S, S).cuda().contiguous(memory_format=cl)])
    torch.cuda.synchronize()
    # a member-batched launch flags the lead member's gradient views only (optim.ParamPool.step): read every member's slice
    got = []
    for m, net in enumerate(tr.gen_a2b_s):
        touched = {k for k, p in tr.gen_a2b_s[0].named_parameters() if p._cg_grad._cg_touched}
        got.append({k: P.np_(p._cg_grad) for k, p in net.named_parameters() if k in touched})

    worst = {}
    for m in range(C):
        sd = {k: torch.as_tensor(np.asarray(v)).double().clone().requires_grad_(not k.endswith(('running_mean', 'running_var')))
              for k, v in state['a2b']['gen'][m].items()}
        og = O.OracleGen(sd, cfg['gen'])
        xd = x_a.double()
        im = og.decode(og.encode_content(xd), style.double(), xd)
        torch.autograd.backward([im, og.mask_s], [up_im[m].double(), up_mask[m].double()])
        ref = {k: v.grad.numpy() for k, v in sd.items() if v.requires_grad and v.grad is not None}
        assert set(got[m]) == set(ref), set(got[m]) ^ set(ref)
        for k, r in ref.items():
            n = float(np.sqrt((r ** 2).sum()))
            e = float(np.sqrt(((got[m][k].astype(np.float64) - r) ** 2).sum()))
            if k.endswith('conv.bias') and n < 1e-6 * float(np.sqrt((ref[k[:-4] + 'weight'] ** 2).sum())):
                # a bias in front of an instance norm: its gradient is identically zero, what is left is round-off --
                # ours must be as negligible against the layer's weight gradient as the oracle's
                wn = float(np.sqrt((ref[k[:-4] + 'weight'] ** 2).sum()))
                assert e <= 1e-5 * wn, ("zero-gradient bias", m, k, e, wn)
                continue
            worst[(m, k)] = e / n
    bad = {k: v for k, v in worst.items() if not v <= P.ACT_TOL}
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
    print("\n[smooth-loss generator backward] %d tensors, worst l2-rel vs fp64: %s"
          % (len(worst), [("%d/%s" % k, "%.1e" % v) for k, v in top]))
    assert not bad, bad


def test_member_grouping_is_exact_on_fp32_datapath(cga):
    """One whole iteration, exact-fp32 datapath, council 4 at full width: members one by one (CG_GROUP = 1: the reference's
    loop order, trainer_council.py:328,558,747,826,858) against all four in one launch per layer (CG_GROUP = 4).  Every
    loss and every gradient of every network must agree to round-off (measured: bit for bit on most tensors) -- the
    batched back-propagation's upstream scales (gan_w, council_w x matching weight) and member-to-slice mapping are
    pinned without the chaos band of the generator-gradient comparison against the oracle."""
    cfg = _cfg("male2female_council_folder.yaml", 4)
    cfg['batch_size'] = 1
    cfg['cg_forward_precision'] = 'fp32'
    x_a, x_b = O.synthetic_batch(1, 64)

    def run(group):
        O.seed_all(31)
        tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
        tr.cuda('cuda:0')
        tr._group_max = group
        O.seed_all(32)
        grads, losses = {}, {}

        def snap(kind, attr):
            for i in range(tr.council_size):
                net = getattr(tr, attr)[i]
                lead = getattr(tr, attr)[tr._groups[1][[i in g for g in tr._groups[1]].index(True)][0]]
                touched = {k for k, p in lead.named_parameters() if p._cg_grad._cg_touched}
                grads[(kind, i)] = {k: P.np_(p._cg_grad) for k, p in net.named_parameters() if k in touched}
        tr.dis_update(x_a, x_b, cfg); snap("dis", "dis_a2b_s")
        tr.dis_council_update(x_a, x_b, cfg); snap("disc", "dis_council_a2b_s")
        tr.gen_update(x_a, x_b, cfg, cfg['iteration']); snap("gen", "gen_a2b_s")
        torch.cuda.synchronize()
        for n in ('loss_dis_total_s', 'loss_dis_council_total_s', 'loss_gen_total_s', 'loss_gen_adv_a2b_s', 'council_loss_ab_s'):
            losses[n] = P.lossvec(getattr(tr, n))
        ngroups = len(tr._groups[1])
        del tr
        return grads, losses, ngroups

    try:
        g1, l1, n1 = run(1)
        g4, l4, n4 = run(4)
    finally:
        cga.ops.X3_FORWARD = cga.ops.X3_BACKWARD = cga.ops.X3_DYNAMIC_INPUT = True
    assert (n1, n4) == (4, 1)
    for n in l1:
        np.testing.assert_allclose(l4[n], l1[n], rtol=2e-6, err_msg=n)
    worst = 0.0
    for key in g1:
        assert set(g1[key]) == set(g4[key]) and len(g1[key]) > 0, key
        e = P.l2rel(g4[key], g1[key])
        worst = max(worst, e)
        assert e <= 1e-5, ("grouped vs member-by-member gradients", key, e)
    print("\n[member grouping, fp32 datapath] worst l2-rel gradient difference CG_GROUP=4 vs 1: %.1e" % worst)
