"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REAL reference.

Run in the build container (needs /root/reference):

    python oracle/make_golden.py            # rewrites every fixture under tests/golden/

For each case it builds the reference `Council_Trainer` (through oracle/ref_shim.py) from a
shipped YAML with small overrides (tiny widths so the fixtures stay small), records the
initial state_dicts, a fixed-noise probe forward, and then runs TWO iterations of
dis_update -> dis_council_update -> gen_update (train.py:237-250), snapshotting after each
call: losses, clean gradients (before the next call pollutes them), style noise
drawn from the CPU RNG, colleague picks drawn from Python's RNG, and post-step weights.

Fixtures are consumed by tests/test_oracle_golden.py (oracle vs reference) and by the GPU
parity tests (HIP path vs reference at the same inputs).
"""
import copy
import json
import os
import random
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT_DIR = os.path.join(ROOT, "tests", "golden")

SMALL = {'gen': {'dim': 4, 'mlp_dim': 8, 'n_res': 2}, 'dis': {'dim': 4}, 'display_size': 2}

# cases whose gradients / post-step weights are stored as per-tensor summaries only
LITE = ('glasses_c1', 'm2f_early', 'bidir_c2')

# FULL-WIDTH cases (gen.dim 64 / dis.dim 64 -- the widths of the shipped configs, so every layer with >= 32 channels
# reaches the split-precision / pipelined kernels of the HIP path).  Too large to store whole: the initial weights are
# NOT stored (the HIP trainer constructed from the same seed has bit-identical ones,
# tests/test_host_cpu.py::test_trainer_init_matches_reference; a per-tensor summary is stored as a guard), gradients
# and post-step weights are per-tensor summaries plus the full arrays of every tensor with <= SMALL_TENSOR elements.
WIDE = ('m2f_w64',)
WIDE_OVERRIDES = {'gen': {'dim': 64, 'mlp_dim': 256, 'n_res': 4}, 'dis': {'dim': 64}, 'display_size': 1}
SMALL_TENSOR = 16384

# case that also carries a checkpoint set written by the reference's save() and the iteration a FRESH reference
# trainer computes after resume() from it (trainer_council.py:898-992)
CKPT = ('ckpt_c2',)
CKPT_OVERRIDES = {'gen': {'dim': 4, 'mlp_dim': 8, 'n_res': 1}, 'dis': {'dim': 4}, 'display_size': 2}

CASES = {
    # name: (yaml, overrides, image size, batch)
    'm2f_c3': ('male2female_council_folder.yaml',
               {'council': {'council_size': 3}, 'iteration': 60000}, 32, 2),
    'anime_c2': ('anime2face_council_folder.yaml',
                 {'council': {'council_size': 2}, 'iteration': 60000}, 32, 2),
    'glasses_c2': ('galsses_council_folder.yaml',
                   {'council': {'council_size': 2}, 'iteration': 60000}, 64, 1),
    'glasses_c1': ('galsses_council_folder.yaml',
                   {'council': {'council_size': 1}, 'iteration': 60000}, 32, 3),
    'm2f_early': ('male2female_council_folder.yaml',
                  {'council': {'council_size': 2}, 'iteration': 100}, 32, 2),
    'bidir_c2': ('male2female_council_folder.yaml',
                 {'council': {'council_size': 2}, 'iteration': 60000, 'do_b2a': True}, 32, 2),
    'm2f_w64': ('male2female_council_folder.yaml',
                {'council': {'council_size': 2}, 'iteration': 60000}, 64, 1),
    'ckpt_c2': ('male2female_council_folder.yaml',
                {'council': {'council_size': 2}, 'iteration': 60000}, 32, 2),
}


def deep_update(d, u):
    for k, v in u.items():
        if isinstance(v, dict):
            deep_update(d[k], v)
        else:
            d[k] = v


def build_config(yaml_name, overrides, size, batch, widths=None):
    cfg = yaml.safe_load(open(os.path.join(ref_shim.REFERENCE_ROOT, 'configs', yaml_name)))
    deep_update(cfg, copy.deepcopy(widths if widths is not None else SMALL))
    deep_update(cfg, copy.deepcopy(overrides))
    cfg['batch_size'] = batch
    cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = size
    return cfg


class Recorder:
    """Wraps torch.randn and random.choice so the fixture records every host-RNG draw."""

    def __init__(self):
        self.randn, self.choice = [], []
        self._randn, self._choice = torch.randn, random.choice

    def __enter__(self):
        def randn(*a, **k):
            t = self._randn(*a, **k)
            self.randn.append(t.clone())
            return t

        def choice(seq):
            c = self._choice(seq)
            self.choice.append(int(c))
            return c
        torch.randn, random.choice = randn, choice
        return self

    def __exit__(self, *exc):
        torch.randn, random.choice = self._randn, self._choice


def sd_np(m):
    return {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}


def grads_np(m):
    return {k: (p.grad.detach().cpu().numpy().copy() if p.grad is not None else None)
            for k, p in m.named_parameters()}


def summary(d):
    """Per-tensor [sum, l2 norm, abs-max] in sorted-key order -- a compact pin for big dicts."""
    keys = sorted(k for k, v in d.items() if v is not None)
    return np.array([[float(d[k].astype(np.float64).sum()), float(np.sqrt((d[k].astype(np.float64) ** 2).sum())),
                      float(np.abs(d[k]).max())] for k in keys], dtype=np.float64)


def scalar(x):
    return float(x.detach()) if torch.is_tensor(x) else float(x)


def record_sample(t, x_a, x_b, put):
    """sample() (trainer_council.py:643-733) on the initial weights: the 8-tuple for return_mask True and False, the
    fixed display styles it uses and the fresh style noise it drew.  The host RNG is restored afterwards so the
    training iterations recorded next see the stream they always saw."""
    st_t, st_r, st_n = torch.get_rng_state(), random.getstate(), np.random.get_state()
    put('sample/s_a', t.s_a.numpy()); put('sample/s_b', t.s_b.numpy())
    n = min(x_a.size(0), t.s_a.size(0))          # train.py hands sample() display_size images (the fixed styles' count)
    x_a, x_b = x_a[:n], x_b[:n]
    put('sample/n', n)
    for tag, rm in (('mask', True), ('recon', False)):
        with Recorder() as rec, torch.no_grad():
            out = t.sample(x_a, x_b, return_mask=rm)
        put('sample/%s/randn' % tag, np.stack([r.numpy() for r in rec.randn]))
        assert len(out) == 8
        put('sample/%s/none' % tag, [int(o is None) for o in out])
        for k, o in enumerate(out):
            if o is not None:
                put('sample/%s/%d' % (tag, k), o.numpy())
    torch.set_rng_state(st_t); random.setstate(st_r); np.random.set_state(st_n)


def run_iteration(t, cfg, x_a, x_b, nets, dirs, C, pre, put, snap):
    """One train.py:237-250 iteration of the reference with every host-RNG draw, loss and network snapshot recorded."""
    with Recorder() as rec:
        t.dis_update(x_a, x_b, cfg)
    put(pre + 'dis/randn', np.stack([r.numpy() for r in rec.randn]))
    put(pre + 'dis/loss_total', [scalar(v) for v in t.loss_dis_total_s])
    for d in dirs:
        put(pre + 'dis/loss_%s' % d, [scalar(v) for v in getattr(t, 'loss_dis_%s_s' % d)])
        for i in range(C):
            snap(pre + 'dis/', d, 'dis', i)

    if 'dis_council' in nets:
        with Recorder() as rec:
            t.dis_council_update(x_a, x_b, cfg)
        ran = len(rec.randn) > 0
        put(pre + 'disc/ran', int(ran))
        if ran:
            put(pre + 'disc/randn', np.stack([r.numpy() for r in rec.randn]))
            put(pre + 'disc/choice', rec.choice)
            put(pre + 'disc/loss_total', [scalar(v) for v in t.loss_dis_council_total_s])
            for d in dirs:
                for i in range(C):
                    snap(pre + 'disc/', d, 'dis_council', i)

    with Recorder() as rec:
        t.gen_update(x_a, x_b, cfg, cfg['iteration'])
    put(pre + 'gen/randn', np.stack([r.numpy() for r in rec.randn]))
    put(pre + 'gen/loss_total', [scalar(v) for v in t.loss_gen_total_s])
    for d in dirs:
        ab = 'ab' if d == 'a2b' else 'ba'
        put(pre + 'gen/loss_adv_%s' % d, [scalar(v) for v in getattr(t, 'loss_gen_adv_%s_s' % d)])
        put(pre + 'gen/council_loss_%s' % d, [scalar(v) for v in getattr(t, 'council_loss_%s_s' % ab)])
        for nm, attr in (('mask_zero_one', 'loss_gen_mask_zero_one_%s_s'), ('mask_total', 'loss_gen_mask_total_%s_s'),
                         ('mask_tv', 'loss_gen_mask_TV_%s_s')):
            put(pre + 'gen/%s_%s' % (nm, d), [scalar(v) for v in getattr(t, attr % ab)])
        for i in range(C):
            snap(pre + 'gen/', d, 'gen', i)


def make_case(name):
    yaml_name, overrides, size, batch = CASES[name]
    wide, ckpt = name in WIDE, name in CKPT
    cfg = build_config(yaml_name, overrides, size, batch, WIDE_OVERRIDES if wide else CKPT_OVERRIDES if ckpt else None)
    Trainer = ref_shim.reference_trainer_cls()
    random.seed(1); np.random.seed(1); torch.manual_seed(1)          # train.py:55-62
    t = Trainer(cfg, 'cpu')
    C = cfg['council']['council_size']
    dirs = [d for d in ('a2b', 'b2a') if cfg['do_' + d]]
    nets = {'gen': 'gen_%s_s', 'dis': 'dis_%s_s'}
    if cfg['council_w'] != 0:
        nets['dis_council'] = 'dis_council_%s_s'
    out = {}

    def put(key, arr):
        out[key] = np.asarray(arr)

    def mods(d, net, tr=None):
        return getattr(tr if tr is not None else t, nets[net] % d)

    for d in dirs:
        for net in nets:
            for i in range(C):
                if wide or ckpt:      # ckpt: the test takes every weight from the checkpoint files
                    put('initsum/%s/%s/%d' % (d, net, i), summary(sd_np(mods(d, net)[i])))
                    continue
                for k, v in sd_np(mods(d, net)[i]).items():
                    put('init/%s/%s/%d/%s' % (d, net, i, k), v)
    if wide:
        put('init_from_seed', 1)

    g = torch.Generator().manual_seed(7)
    x_a = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    x_b = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    put('x_a', x_a.numpy()); put('x_b', x_b.numpy())

    # probe forward with recorded style noise: pins encode/decode/mask/D/council-D forward
    gp = torch.Generator().manual_seed(11)
    s_probe = torch.randn(batch, cfg['gen']['style_dim'], 1, 1, generator=gp)
    put('probe/style', s_probe.numpy())
    with torch.no_grad():
        for d in dirs:
            src = x_a if d == 'a2b' else x_b
            gen = mods(d, 'gen')[0]
            c, s_fake = gen.encode(src)
            img, mask = gen.decode(c, s_probe, src, return_mask=True)
            put('probe/%s/content' % d, c.numpy()); put('probe/%s/style_fake' % d, s_fake.numpy())
            put('probe/%s/image' % d, img.numpy()); put('probe/%s/mask' % d, mask.numpy())
            for s, o in enumerate(mods(d, 'dis')[0].forward(img)):
                put('probe/%s/dis_out%d' % (d, s), o.numpy())
            if 'dis_council' in nets:
                for s, o in enumerate(mods(d, 'dis_council')[0].forward(img, src)):
                    put('probe/%s/disc_out%d' % (d, s), o.numpy())

    record_sample(t, x_a, x_b, put)

    lite = name in LITE or ckpt

    def make_snap(tr, it):
        def snap(pre, d, net, i):
            """grads + post-step weights of one network: full arrays at iteration 0 of a full case (small tensors only
            for a full-width case), per-tensor summaries always."""
            m = mods(d, net, tr)[i]
            g = {k: v for k, v in grads_np(m).items() if v is not None}
            w = {k: v for k, v in sd_np(m).items() if 'running_' not in k}
            put(pre + 'gradsum/%s/%d' % (d, i), summary(g))
            put(pre + 'postsum/%s/%d' % (d, i), summary(w))
            if it == 0 and not lite:
                for k, v in g.items():
                    if not wide or v.size <= SMALL_TENSOR:
                        put(pre + 'grad/%s/%d/%s' % (d, i, k), v)
            if it == 0 and not lite and i == 0:
                for k, v in w.items():
                    if not wide or v.size <= SMALL_TENSOR:
                        put(pre + 'post/%s/%d/%s' % (d, i, k), v)
        return snap

    n_it = 1 if ckpt else 2
    for it in range(n_it):
        cfg['iteration'] = overrides['iteration'] + it
        run_iteration(t, cfg, x_a, x_b, nets, dirs, C, 'it%d/' % it, put, make_snap(t, it))

    if ckpt:
        # the reference writes its checkpoint set; a FRESH reference trainer (different seed: every weight and Adam moment
        # must come from the files) resumes from it and runs one more iteration.  The files travel inside the fixture.
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            t.save(tmp, cfg['iteration'])
            for fn in sorted(os.listdir(tmp)):
                put('ckpt/' + fn, np.frombuffer(open(os.path.join(tmp, fn), 'rb').read(), dtype=np.uint8))
            random.seed(99); np.random.seed(99); torch.manual_seed(99)
            t2 = Trainer(cfg, 'cpu')
            put('resume/iterations', int(t2.resume(tmp, cfg)))
        cfg['iteration'] = overrides['iteration'] + 1
        run_iteration(t2, cfg, x_a, x_b, nets, dirs, C, 'resume/', put, make_snap(t2, 0))

    cfg['iteration'] = overrides['iteration']
    out['config_json'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-12s %4d arrays  %.2f MB' % (name, len(out), os.path.getsize(path) / 1e6))


if __name__ == '__main__':
    for name in (sys.argv[1:] or CASES):
        make_case(name)
