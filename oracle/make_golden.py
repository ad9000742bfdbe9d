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
}


def deep_update(d, u):
    for k, v in u.items():
        if isinstance(v, dict):
            deep_update(d[k], v)
        else:
            d[k] = v


def build_config(yaml_name, overrides, size, batch):
    cfg = yaml.safe_load(open(os.path.join(ref_shim.REFERENCE_ROOT, 'configs', yaml_name)))
    deep_update(cfg, copy.deepcopy(SMALL))
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


def make_case(name):
    yaml_name, overrides, size, batch = CASES[name]
    cfg = build_config(yaml_name, overrides, size, batch)
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

    def mods(d, net):
        return getattr(t, nets[net] % d)

    for d in dirs:
        for net in nets:
            for i in range(C):
                for k, v in sd_np(mods(d, net)[i]).items():
                    put('init/%s/%s/%d/%s' % (d, net, i, k), v)

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

    lite = name in LITE

    def snap(pre, d, net, i, it):
        """grads + post-step weights of one network: full arrays at iteration 0 of a full case,
        per-tensor summaries always."""
        m = mods(d, net)[i]
        g = {k: v for k, v in grads_np(m).items() if v is not None}
        w = {k: v for k, v in sd_np(m).items() if 'running_' not in k}
        put(pre + 'gradsum/%s/%d' % (d, i), summary(g))
        put(pre + 'postsum/%s/%d' % (d, i), summary(w))
        if it == 0 and not lite:
            for k, v in g.items():
                put(pre + 'grad/%s/%d/%s' % (d, i, k), v)
        if it == 0 and not lite and i == 0:
            for k, v in w.items():
                put(pre + 'post/%s/%d/%s' % (d, i, k), v)

    for it in range(2):
        cfg['iteration'] = overrides['iteration'] + it
        pre = 'it%d/' % it
        with Recorder() as rec:
            t.dis_update(x_a, x_b, cfg)
        put(pre + 'dis/randn', np.stack([r.numpy() for r in rec.randn]))
        put(pre + 'dis/loss_total', [scalar(v) for v in t.loss_dis_total_s])
        for d in dirs:
            put(pre + 'dis/loss_%s' % d, [scalar(v) for v in getattr(t, 'loss_dis_%s_s' % d)])
            for i in range(C):
                snap(pre + 'dis/', d, 'dis', i, it)

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
                        snap(pre + 'disc/', d, 'dis_council', i, it)

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
                snap(pre + 'gen/', d, 'gen', i, it)

    cfg['iteration'] = overrides['iteration']
    out['config_json'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-12s %4d arrays  %.2f MB' % (name, len(out), os.path.getsize(path) / 1e6))


if __name__ == '__main__':
    for name in (sys.argv[1:] or CASES):
        make_case(name)
