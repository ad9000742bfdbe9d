"""GPU parity of the HIP networks and of whole training iterations against (a) the golden fixtures
recorded from the REAL reference (tests/golden/*.npz) and (b) the oracle at full network widths.

Tolerances: activations / images / masks / losses / discriminator gradients
<= 1e-3 max-abs over max-abs (measured: ~1e-6); generator gradients are judged against an fp64
oracle because the reference's own fp32-vs-fp64 gap is 2-4e-3: err(ours, fp64) <= 2 * err(ref32, fp64)
(with a 2e-4 floor for nets where the reference gap happens to be tiny)."""
import copy
import random

import numpy as np
import pytest
import torch

import parity_util as P
from golden_util import Golden, case_names, rel_err, summary
from oracle import council_oracle as O

pytestmark = pytest.mark.gpu

ACT_TOL = 1e-3


@pytest.fixture(scope="module")
def cga():
    import council_gan_amd
    council_gan_amd.hip.load()
    return council_gan_amd


_orig_randn, _orig_choice = torch.randn, random.choice


@pytest.fixture(autouse=True)
def _restore_rng_functions():
    yield
    torch.randn, random.choice = _orig_randn, _orig_choice


def patch_randn(queue):
    def randn(*shape, **k):
        return torch.from_numpy(np.array(queue.pop(0)))
    torch.randn = randn


def patch_choice(queue):
    def choice(seq):
        c = int(queue.pop(0))
        assert c in seq
        return c
    random.choice = choice


def build_trainer(cga, cfg, state):
    """Council_Trainer with the fixture's weights loaded through load_state_dict (checkpoint path)."""
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    for d in state:
        for net, attr in (('gen', 'gen_%s_s'), ('dis', 'dis_%s_s'), ('dis_council', 'dis_council_%s_s')):
            if net not in state[d]:
                continue
            for i, sd in enumerate(state[d][net]):
                getattr(tr, attr % d)[i].load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    tr.cuda('cuda:0')
    return tr


def np_(t):
    return t.detach().float().cpu().numpy()


@pytest.mark.parametrize("name", case_names())
def test_probe_forward_vs_reference(cga, name):
    g = Golden(name)
    tr = build_trainer(cga, g.cfg, g.init_state())
    s = torch.from_numpy(g["probe/style"]).cuda()
    x = {"a2b": torch.from_numpy(g["x_a"]).cuda(), "b2a": torch.from_numpy(g["x_b"]).cuda()}
    errs = {}
    with torch.no_grad():
        for d in g.dirs:
            gen = getattr(tr, 'gen_%s_s' % d)[0]
            xi = tr._img(x[d])
            c, s_fake = gen.encode(xi)
            img, mask = gen.decode(c, s, xi, return_mask=True)
            errs[d + '/content'] = rel_err(np_(c), g["probe/%s/content" % d])
            errs[d + '/style'] = rel_err(np_(s_fake), g["probe/%s/style_fake" % d])
            errs[d + '/image'] = rel_err(np_(img), g["probe/%s/image" % d])
            errs[d + '/mask'] = rel_err(np_(mask), g["probe/%s/mask" % d])
            for sc, o in enumerate(getattr(tr, 'dis_%s_s' % d)[0].forward(img)):
                errs[d + '/dis%d' % sc] = rel_err(np_(o), g["probe/%s/dis_out%d" % (d, sc)])
            if "dis_council" in g.nets:
                for sc, o in enumerate(getattr(tr, 'dis_council_%s_s' % d)[0].forward(img, xi)):
                    errs[d + '/disc%d' % sc] = rel_err(np_(o), g["probe/%s/disc_out%d" % (d, sc)])
    assert max(errs.values()) < ACT_TOL, errs


def grads_of(net):
    out = {}
    for k, p in net.named_parameters():
        gbuf = getattr(p, '_cg_grad', None)
        if gbuf is not None and gbuf._cg_touched:
            out[k] = np_(gbuf)
    return out


def weights_of(net):
    return {k: np_(v) for k, v in net.state_dict().items() if 'running_' not in k}


def l2rel(a, b):
    num = np.sqrt(sum(float(((a[k].astype(np.float64) - b[k].astype(np.float64)) ** 2).sum()) for k in b))
    den = np.sqrt(sum(float((b[k].astype(np.float64) ** 2).sum()) for k in b))
    return num / max(den, 1e-30)


def run_iteration(g, tr, cfg, it, pre, x_a, x_b, snap):
    patch_randn(list(g[pre + "dis/randn"]))
    tr.dis_update(x_a, x_b, cfg)
    snap("dis")
    if "dis_council" in g.nets and bool(g[pre + "disc/ran"]):
        patch_randn(list(g[pre + "disc/randn"]))
        patch_choice(list(g[pre + "disc/choice"]))
        tr.dis_council_update(x_a, x_b, cfg)
        snap("disc")
    elif "dis_council" in g.nets:
        tr.dis_council_update(x_a, x_b, cfg)      # must early-out exactly like the reference
    patch_randn(list(g[pre + "gen/randn"]))
    tr.gen_update(x_a, x_b, cfg, cfg["iteration"])
    snap("gen")


@pytest.mark.parametrize("name", case_names())
def test_two_iterations_vs_reference(cga, name):
    g = Golden(name)
    cfg = copy.deepcopy(g.cfg)
    tr = build_trainer(cga, cfg, g.init_state())
    # fp64 oracle twin for the generator-gradient noise floor
    otr64 = O.OracleTrainer(copy.deepcopy(cfg), g.init_state(), dtype=torch.float64)
    x_a, x_b = torch.from_numpy(g["x_a"]), torch.from_numpy(g["x_b"])
    base = cfg["iteration"]
    kinds = {"dis": ("dis", "dis_%s_s"), "disc": ("dis_council", "dis_council_%s_s"), "gen": ("gen", "gen_%s_s")}
    report = {}
    for it in range(2):
        cfg["iteration"] = base + it
        pre = "it%d/" % it
        got = {}

        def snap(kind):
            net, attr = kinds[kind]
            for d in g.dirs:
                for i in range(g.C):
                    m = getattr(tr, attr % d)[i]
                    got[(kind, d, i)] = (grads_of(m), weights_of(m))
        run_iteration(g, tr, cfg, it, pre, x_a, x_b, snap)

        # fp64 oracle, same host-RNG replay; gradients are snapshotted right after each update (gen_update's
        # backward also deposits gradients in the discriminators, in the oracle as in the reference)
        ocfg = copy.deepcopy(cfg)
        g64 = {}

        def snap64(kind):
            net = kinds[kind][0]
            for d in g.dirs:
                for i in range(g.C):
                    g64[(kind, d, i)] = {k: t.grad.numpy().copy() for k, t in otr64.sd[d][net][i].items()
                                         if t.requires_grad and t.grad is not None}
        patch_randn(list(g[pre + "dis/randn"])); otr64.dis_update(x_a, x_b, ocfg); snap64("dis")
        if "dis_council" in g.nets and bool(g[pre + "disc/ran"]):
            patch_randn(list(g[pre + "disc/randn"])); patch_choice(list(g[pre + "disc/choice"]))
            otr64.dis_council_update(x_a, x_b, ocfg); snap64("disc")
        patch_randn(list(g[pre + "gen/randn"])); otr64.gen_update(x_a, x_b, ocfg, ocfg["iteration"]); snap64("gen")
        torch.randn, random.choice = _orig_randn, _orig_choice

        # ---- losses -------------------------------------------------------------------------
        def lossvec(v):
            return np.array([float(t.detach()) if torch.is_tensor(t) else float(t) for t in v])
        # Full-width fixture, SECOND iteration: Adam's first step moved all 17 M generator weights by +-lr, including
        # those whose gradient is round-off noise, and the oracle itself (same ATen as the reference) is then off the
        # reference by up to 2e-4 in the losses, 5e-4 in discriminator gradients and 4-8e-2 in generator gradient norms
        # (tests/test_oracle_golden.py): iteration 0 is the strict pin, iteration 1 a 2e-3 / 5e-3 consistency check.
        chaotic = g.from_seed and it > 0
        LT = 2e-3 if chaotic else ACT_TOL
        np.testing.assert_allclose(lossvec(tr.loss_dis_total_s), g[pre + "dis/loss_total"], rtol=LT)
        if (pre + "disc/loss_total") in g:
            np.testing.assert_allclose(lossvec(tr.loss_dis_council_total_s), g[pre + "disc/loss_total"], rtol=LT)
        np.testing.assert_allclose(lossvec(tr.loss_gen_total_s), g[pre + "gen/loss_total"], rtol=LT)
        for d in g.dirs:
            ab = 'ab' if d == 'a2b' else 'ba'
            np.testing.assert_allclose(lossvec(getattr(tr, 'loss_gen_adv_%s_s' % d)), g[pre + "gen/loss_adv_%s" % d],
                                       rtol=LT)
            np.testing.assert_allclose(lossvec(getattr(tr, 'council_loss_%s_s' % ab)),
                                       g[pre + "gen/council_loss_%s" % d], rtol=LT, atol=1e-7)
            for nm, attr in (("mask_zero_one", "loss_gen_mask_zero_one_%s_s"), ("mask_total", "loss_gen_mask_total_%s_s"),
                             ("mask_tv", "loss_gen_mask_TV_%s_s")):
                ref = g[pre + "gen/%s_%s" % (nm, d)]
                if not len(ref):
                    continue
                mine = lossvec(getattr(tr, attr % ab))
                if nm == "mask_zero_one":
                    # mean 1/(|m-c|+eps) amplifies a mask perturbation by up to 1/eps^2: the reference's own fp32
                    # value is off its fp64 value by more than 1e-3 on some fixtures, so this criterion is judged
                    # like the generator gradients -- distance to the fp64 oracle, at most twice the reference's
                    r64 = lossvec(otr64.loss_mask_zero_one[d])
                    tol = P.mask_zo_tol(ref, r64, LT)
                    assert np.all(np.abs(mine - r64) <= tol), (nm, d, mine, ref, r64)
                else:
                    np.testing.assert_allclose(mine, ref, rtol=LT, atol=1e-7)

        # ---- gradients and post-step weights --------------------------------------------------
        # the fixture's own fp32-vs-fp64 generator-gradient gap (largest over the members: they draw from one lottery), from the
        # tensors it stores whole -- what the level criterion and the norm band below are derived from
        import parity_util
        xa = g["x_a"]
        pixels = int(xa.shape[0] * xa.shape[2] * xa.shape[3])
        e_ref_run = 0.0
        for (kind, d, i) in got:
            if kind == "gen":
                rf = g.sub(pre + "gen/grad/%s/%d/" % (d, i))
                if rf:
                    e_ref_run = max(e_ref_run, l2rel(rf, {k: v for k, v in g64[(kind, d, i)].items() if k in rf}))
        for (kind, d, i), (gs, ws) in got.items():
            net = kinds[kind][0]
            ref_sum = g[pre + "%s/gradsum/%s/%d" % (kind, d, i)]
            assert len(gs) == ref_sum.shape[0], "set of tensors that received a gradient differs (%s %s %d)" % (kind, d, i)
            scale = ref_sum[:, 1].max()
            mine = summary(gs)
            if kind != "gen":
                # discriminators: clean fp32 gradients
                bad = np.abs(mine[:, 1] - ref_sum[:, 1]) > (5e-3 if chaotic else ACT_TOL) * ref_sum[:, 1] + 1e-5 * scale
                assert not bad.any(), (kind, d, i, mine[bad], ref_sum[bad])
            elif g.from_seed and it == 0:
                # full-width generator: every tensor's gradient norm against the reference's (the 17 M-element tensors
                # are not stored whole).  | ||ours|| - ||ref|| | <= ||ours - ref|| <= ||ours - fp64|| + ||ref - fp64||: the
                # band is the level the whole-tensor criterion allows THIS run (parity_util.gen_grad_limit on the fixture's
                # own fp32-vs-fp64 gap) plus that gap -- not a constant (it was 6e-2 until round 4)
                band = parity_util.gen_grad_limit(pixels, e_ref_run) + e_ref_run
                dev_n = np.abs(mine[:, 1] - ref_sum[:, 1]) / (ref_sum[:, 1] + 1e-4 * scale)
                print("[norm band] %s %s %d: largest per-tensor norm deviation %.2e, band %.2e (fixture gap %.2e)"
                      % (kind, d, i, float(dev_n.max()), band, e_ref_run))
                bad = np.abs(mine[:, 1] - ref_sum[:, 1]) > band * ref_sum[:, 1] + 1e-4 * scale
                assert not bad.any(), (kind, d, i, mine[bad], ref_sum[bad])
            ref_full = g.sub(pre + "%s/grad/%s/%d/" % (kind, d, i))
            if ref_full:
                # (a full-width fixture stores only the tensors with <= 16384 elements whole)
                r64 = {k: v for k, v in g64[(kind, d, i)].items() if k in ref_full}
                e_ref = l2rel(ref_full, r64)
                e_ours = l2rel({k: gs[k] for k in r64}, r64)
                report[(it, kind, d, i)] = (e_ours, e_ref)
                if kind == "gen":
                    # north star: within 1e-3 rel-fp32; where the reference's own fp32-vs-fp64 gap is larger than that
                    # (steep mask head) the measured chaos band of the reference arithmetic
                    # level + per-tensor uniformity, see parity_util.check_gen_grad
                    parity_util.check_gen_grad({k: gs[k] for k in r64}, ref_full, r64, (it, d, i), pixels=pixels,
                                               e_ref_run=e_ref_run)
                else:
                    assert e_ours <= ACT_TOL, ("discriminator gradient", kind, it, d, i, e_ours, e_ref)
            # post-step weights: one Adam step moves every weight by <= lr; compare the bulk
            ref_ws = g[pre + "%s/postsum/%s/%d" % (kind, d, i)]
            minew = summary(ws)
            # tensors whose gradient is round-off noise (conv biases feeding an instance norm: the exact gradient is zero) take
            # Adam-normalised random steps of size ~lr in the reference too -- their norm after two steps is a random walk
            # (lr * sqrt(n) * O(1)), not a quantity two fp32 evaluations share: excluded here and below
            gkeys = sorted(gs)
            noisy = {k for k, r in zip(gkeys, ref_sum) if r[1] < 1e-6 * scale}
            off = np.abs(minew[:, 1] - ref_ws[:, 1]) > 1e-4 * ref_ws[:, 1] + 3e-4
            bad_w = [(k, float(a), float(b)) for k, a, b, o in zip(sorted(ws), minew[:, 1], ref_ws[:, 1], off) if o and k not in noisy]
            assert not bad_w, (kind, d, i, bad_w)
            ref_wfull = g.sub(pre + "%s/post/%s/%d/" % (kind, d, i))
            for k, v in ref_wfull.items():
                if k not in noisy:
                    assert float(np.abs(ws[k] - v).mean()) < 2e-6, (kind, d, i, k)
    print("\n[grad l2-rel vs fp64]", {k: ("%.2e" % v[0], "%.2e" % v[1]) for k, v in report.items()})


def test_full_width_iteration_vs_oracle(cga):
    """Real channel widths (gen dim 64 / dis dim 64, every FAST tile path) at 64x64, batch 2, council 2: one full
    iteration against the oracle built from the trainer's own seeded weights -- losses and discriminator gradients
    <= 1e-3, generator gradients within twice the fp32 oracle's own distance to its fp64 twin, post-step weights
    (criteria: tests/parity_util.py)."""
    import os
    import yaml
    import parity_util as P
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "male2female_council_folder.yaml")))
    cfg['council']['council_size'] = 2
    cfg['iteration'] = 60000
    P.iteration_vs_oracle(cga, cfg, 64, 2, seed=1, report="full width 64^2 council 2 B2")


def test_content_cache_is_invalidated(cga):
    """The encoder runs once per (batch, generator weights): a batch modified in place, a new batch object and a
    generator step must each force a re-encode (trainer._content)."""
    import os
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "male2female_council_folder.yaml")))
    cfg['gen'].update(dim=8, mlp_dim=16, n_res=1)
    cfg['dis'].update(dim=8)
    cfg['council']['council_size'] = 2
    cfg['batch_size'] = 2
    cfg['iteration'] = 60000
    O.seed_all(3)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    tr.cuda('cuda:0')
    x_a, x_b = O.synthetic_batch(2, 32)
    x_a, x_b = x_a.cuda(), x_b.cuda()

    def fresh(x):
        with torch.no_grad():                       # members one by one, stacked member-major
            return torch.cat([g.encode_content(tr._img(x)) for g in tr.gen_a2b_s]).clone()

    def cached():
        grp = tr._plan_groups(tr._img(x_a, 'a'))[0]
        assert grp == [0, 1]                        # both members run as one member-batched launch
        return tr._content('a2b', grp, tr._rep(tr._img(x_a, 'a'), len(grp)), need_grad=False)

    def same(a, b):     # member-batched encoder == the members one by one (tile shapes, hence the order in which the
        return float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())      # fp64 norm partials combine, may differ)

    tr.dis_update(x_a, x_b, cfg)
    c0 = cached()
    assert same(c0, fresh(x_a))
    x_a.mul_(0.5)                                   # in-place edit of the same tensor object
    tr.dis_council_update(x_a, x_b, cfg)
    c1 = cached()
    assert same(c1, fresh(x_a)) and not same(c0, c1)
    tr.gen_update(x_a, x_b, cfg, 60000)             # generator step: weights changed, tape consumed
    assert not tr._enc_cache
    tr.dis_update(x_a, x_b, cfg)
    c2 = cached()
    assert same(c2, fresh(x_a)) and not same(c1, c2)


def test_split_precision_decoder_matches_fp32(cga):
    """The tape-free decoder passes run their 3x3 convolutions as fp16 x 3 MFMA products (ops.conv2d_x3): the
    generated image must agree with the exact-fp32 path far inside the 1e-3 tolerance (measured ~1e-5)."""
    import os
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "male2female_council_folder.yaml")))
    cfg['council']['council_size'] = 2
    cfg['batch_size'] = 2
    cfg['iteration'] = 60000
    O.seed_all(5)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    tr.cuda('cuda:0')
    if not tr._split_fwd:
        pytest.skip("split-precision path disabled (CG_FORWARD_PRECISION=fp32)")
    x_a, _ = O.synthetic_batch(2, 64)
    x = tr._img(x_a)
    s = torch.randn(2, 64, 1, 1).cuda()
    gen = tr.gen_a2b_s[0]
    with torch.no_grad():
        cga.ops.X3_FORWARD = False                 # exact-fp32 reference: fp32 MFMA in every convolution
        c = gen.encode_content(x)
        ref_img = gen.decode(c, s, x).clone()
        ref_mask = gen.dec.mask_s.clone()
        cga.ops.X3_FORWARD = True
        c3 = gen.encode_content(x)                 # split-precision forward of the (tape-capable) general path
        e_c = float((c3 - c).abs().max() / c.abs().max())
        assert 0 < e_c < 2e-4, e_c
        with tr._split_decode('a2b', 0):
            assert gen.dec.split_active
            img = gen.decode(c, s, x)
            mask = gen.dec.mask_s
        assert not gen.dec.split_active
    assert not torch.equal(img, ref_img), "the split-precision trunk did not run"
    e_img = float((img - ref_img).abs().max() / ref_img.abs().max())
    e_mask = float((mask - ref_mask).abs().max() / ref_mask.abs().max())
    print("\n[split-precision decoder vs fp32] image %.2e  mask %.2e" % (e_img, e_mask))
    assert e_img < 2e-4 and e_mask < 2e-4, (e_img, e_mask)
    # a generator step invalidates the split weights: the next use must re-split them
    mgr = gen.dec._split_blocks()[0]._cg_wmgr
    v0 = mgr.version
    tr.dis_update(x_a, x_a, cfg); tr.dis_council_update(x_a, x_a, cfg); tr.gen_update(x_a, x_a, cfg, 60000)
    with tr._split_decode('a2b', 0):
        pass
    assert mgr.version != v0 and mgr.version == tr._pools['gen'].version


def _small_cfg():
    import os
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "male2female_council_folder.yaml")))
    cfg['gen'].update(dim=16, mlp_dim=32, n_res=2)
    cfg['dis'].update(dim=16)
    cfg['council']['council_size'] = 2
    cfg['batch_size'] = 2
    cfg['iteration'] = 60000
    cfg['display_size'] = 2
    return cfg


def test_sample_layout_and_values(cga):
    """`sample()` (trainer_council.py:643-733) returns the reference's 8-tuple; the a2b half is
    (inputs repeated per member, masks, translation with the fixed style, translation with a fresh style)."""
    cfg = _small_cfg()
    O.seed_all(2)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    tr.cuda('cuda:0')
    x_a, x_b = O.synthetic_batch(2, 64)
    out = tr.sample(x_a.cuda(), x_b.cuda())
    assert len(out) == 8 and all(o is None for o in out[4:])            # b2a disabled in this config
    xs, masks, x1, x2 = out[:4]
    assert tuple(xs.shape) == tuple(x1.shape) == tuple(x2.shape) == (4, 3, 64, 64) and tuple(masks.shape) == (4, 3, 64, 64)
    assert torch.equal(xs[0], xs[1]) and torch.equal(xs[0].cpu(), x_a[0])           # sample n, members 0 and 1
    for t in (masks, x1, x2):
        assert torch.isfinite(t).all()
    assert float(masks.min()) >= 0.0 and float(masks.max()) <= 1.0
    with torch.no_grad():                                                          # fixed-style output = a plain decode
        gen = tr.gen_a2b_s[1]
        xi = tr._img(x_a[:1].cuda())
        ref = gen.decode(gen.encode_content(xi), tr.s_b[:1], xi)
    assert rel_err(np_(x1[1:2]), np_(ref)) < 1e-5
    assert not torch.equal(x1, x2)


def test_save_resume_roundtrip(cga, tmp_path):
    """checkpoint files, keys and tensors follow trainer_council.py:969-992 / 898-967, and a resumed
    trainer continues bit-identically (weights, Adam moments, step counts).  Loss matching is switched off for the
    continuation check: its 100-deep loss-history deques (trainer_council.py:131-138) are not part of the
    reference's checkpoints either, so a resumed run restarts them at ones by design."""
    import os
    cfg = _small_cfg()
    cfg['do_w_loss_matching'] = False
    O.seed_all(4)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    tr.cuda('cuda:0')
    x_a, x_b = O.synthetic_batch(2, 64)
    x_a, x_b = x_a.cuda(), x_b.cuda()

    def one_iteration(t, seed):
        O.seed_all(seed)
        t.dis_update(x_a, x_b, cfg); t.dis_council_update(x_a, x_b, cfg); t.gen_update(x_a, x_b, cfg, 60000)

    one_iteration(tr, 10)
    tr.save(str(tmp_path), 122)
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(['a2b_gen_%d_00000123.pt' % i for i in range(2)] + ['a2b_dis_%d_00000123.pt' % i for i in range(2)] +
                           ['a2b_dis_council_%d_00000123.pt' % i for i in range(2)] + ['optimizer_%d.pt' % i for i in range(2)])
    ck = torch.load(os.path.join(tmp_path, 'a2b_gen_0_00000123.pt'), map_location='cpu')
    assert list(ck) == ['a2b'] and tuple(ck['a2b']['enc_content.model.0.conv.weight'].shape) == (16, 3, 7, 7)
    opt = torch.load(os.path.join(tmp_path, 'optimizer_1.pt'), map_location='cpu')
    assert sorted(opt) == ['dis', 'dis_council', 'gen'] and 'state' in opt['gen'] and 'param_groups' in opt['gen']

    O.seed_all(99)                                           # different initial weights: everything must come from disk
    tr2 = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    tr2.cuda('cuda:0')
    assert tr2.resume(str(tmp_path), cfg) == 123
    for a, b in ((tr.gen_a2b_s, tr2.gen_a2b_s), (tr.dis_a2b_s, tr2.dis_a2b_s), (tr.dis_council_a2b_s, tr2.dis_council_a2b_s)):
        for m, m2 in zip(a, b):
            sa, sb = m.state_dict(), m2.state_dict()
            assert list(sa) == list(sb)
            for k in sa:
                assert torch.equal(sa[k], sb[k]), k
    one_iteration(tr, 11)
    one_iteration(tr2, 11)
    for m, m2 in zip(tr.gen_a2b_s, tr2.gen_a2b_s):
        for (k, v), (_, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
            assert torch.equal(v, v2), k
    f = lambda v: [float(t) for t in v]
    assert f(tr.loss_gen_total_s) == f(tr2.loss_gen_total_s)


def test_bench_size_iteration_split_vs_fp32_datapath(cga):
    """BASELINE.json's full configuration (256x256, council 4, batch 4 -- every tile configuration the bench uses): one
    whole iteration on the split-precision datapath against the same iteration on exact fp32 MFMA, same seeds.  The
    oracle needs minutes at this size; the property checked instead is datapath independence: every loss of every member
    of two consecutive iterations agrees to 2e-4 (tolerance of the path: 1e-3), and a repeated run is bit-identical (no atomics, fixed reduction
    orders)."""
    import os
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "male2female_council_folder.yaml")))
    cfg['council']['council_size'] = 4
    cfg['batch_size'] = 4
    cfg['iteration'] = 60000
    x_a, x_b = O.synthetic_batch(4, 256)
    x_a, x_b = x_a.cuda(), x_b.cuda()
    names = ('loss_dis_total_s', 'loss_dis_council_total_s', 'loss_gen_total_s', 'loss_gen_adv_a2b_s',
             'council_loss_ab_s', 'loss_gen_mask_zero_one_ab_s', 'loss_gen_mask_total_ab_s')

    def run(precision):
        c = copy.deepcopy(cfg)
        c['cg_forward_precision'] = precision
        O.seed_all(7)
        tr = cga.Council_Trainer(c, 'cuda:0')
        tr.cuda('cuda:0')
        O.seed_all(8)
        out = {}
        for it in range(2):          # the second iteration's losses see the first one's generator / discriminator steps
            tr.dis_update(x_a, x_b, c); tr.dis_council_update(x_a, x_b, c); tr.gen_update(x_a, x_b, c, 60000 + it)
            out.update({"%s@%d" % (n, it): [float(v) for v in getattr(tr, n)] for n in names})
        w = [float(m.state_dict()['dec.model.0.model.0.model.0.conv.weight'].double().sum()) for m in tr.gen_a2b_s]
        del tr
        return out, w

    try:
        split, w_split = run('split')
        again, w_again = run('split')
        exact, _ = run('fp32')
    finally:
        cga.ops.X3_FORWARD = cga.ops.X3_BACKWARD = cga.ops.X3_DYNAMIC_INPUT = True
    assert split == again and w_split == w_again, "the iteration is not reproducible"
    per = {}
    for n in split:
        for a, b in zip(split[n], exact[n]):
            assert np.isfinite(a) and np.isfinite(b)
            per[n] = max(per.get(n, 0.0), abs(a - b) / max(abs(b), 1e-6))
    print("\n[bench-size iteration] relative loss difference split vs fp32 datapath:", {k: "%.1e" % v for k, v in per.items()})
    # after a generator step the two mask criteria are not comparable at this level: Adam's first step is +-lr for
    # every weight whatever the gradient's size, so round-off-sized gradients flip signs, and the mask = (tanh(10 x) + 1) / 2
    # amplifies that into 1e-3..1e-2 of its mean (the reference on two different BLAS back-ends behaves the same)
    worst = max(v for k, v in per.items() if not (k.endswith("@1") and "mask" in k))
    assert worst < 2e-4, per
    assert max(per.values()) < 5e-2, per


def test_bench_size_generator_forward_vs_oracle(cga):
    """Generator forward at the bench resolution (256x256, full widths, batch 2) against the oracle in fp64, on all three
    datapaths: the tape-free split-precision trunk, the general split-precision path, exact fp32 MFMA."""
    import os
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "male2female_council_folder.yaml")))
    cfg['council']['council_size'] = 2
    cfg['batch_size'] = 2
    cfg['iteration'] = 60000
    O.seed_all(11)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    sd = O.to_numpy_state(tr.gen_a2b_s[0].state_dict())
    tr.cuda('cuda:0')
    x_a, _ = O.synthetic_batch(2, 256)
    s = torch.randn(2, cfg['gen']['style_dim'], 1, 1)
    og = O.OracleGen({k: torch.as_tensor(np.asarray(v)).double() for k, v in sd.items()}, cfg['gen'])
    with torch.no_grad():
        oc = og.encode_content(x_a.double())
        oimg, omask = og.decode(oc, s.double(), x_a.double(), return_mask=True)
    gen = tr.gen_a2b_s[0]
    x = tr._img(x_a)
    sg = s.cuda()
    res = {}
    try:
        with torch.no_grad():
            for name, x3, trunk in (("fp32", False, False), ("split-general", True, False), ("split-trunk", True, True)):
                cga.ops.X3_FORWARD = cga.ops.X3_DYNAMIC_INPUT = x3
                c = gen.encode_content(x)
                if trunk:
                    with tr._split_decode('a2b', 0):
                        img = gen.decode(c, sg, x)
                        mask = gen.dec.mask_s
                else:
                    img = gen.decode(c, sg, x)
                    mask = gen.dec.mask_s
                res[name] = (rel_err(np_(c), oc.numpy()), rel_err(np_(img), oimg.numpy()), rel_err(np_(mask), omask.numpy()))
    finally:
        cga.ops.X3_FORWARD = cga.ops.X3_DYNAMIC_INPUT = tr._split_fwd
    print("\n[256x256 generator forward vs fp64 oracle: content / image / mask]",
          {k: tuple("%.1e" % e for e in v) for k, v in res.items()})
    for k, v in res.items():
        assert max(v) < ACT_TOL, (k, v)
