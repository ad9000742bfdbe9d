"""Layer-by-layer forward error of the generator: HIP path (split and exact-fp32 datapaths) and the fp32 oracle, each
against the fp64 oracle, after every Conv2dBlock (conv + norm + activation) in execution order.

    python tools/diag_forward.py [config yaml] [size] [batch]

Development aid (uses the oracle: never part of the product path)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

import council_gan_amd as cga  # noqa: E402
from council_gan_amd import networks as N  # noqa: E402
from oracle import council_oracle as O  # noqa: E402


def oracle_taps(sd, hp, x, s, dtype):
    taps = []
    orig = O.conv_block

    def rec(*a, **k):
        y = orig(*a, **k)
        taps.append(y.detach().double().numpy())
        return y
    O.conv_block = rec
    try:
        g = O.OracleGen({k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in sd.items()}, hp)
        with torch.no_grad():
            img, mask = g.decode(g.encode_content(x.to(dtype)), s.to(dtype), x.to(dtype), return_mask=True)
    finally:
        O.conv_block = orig
    taps.append(mask.double().numpy())
    taps.append(img.double().numpy())
    return taps


def ours_taps(tr, gen, x, s):
    cga.ops.NORM_NO_F32 = False      # the hooks below read every block's fp32 output
    taps, hooks = [], []
    for m in list(gen.enc_content.modules()) + list(gen.dec.modules()):
        if isinstance(m, N.Conv2dBlock):
            hooks.append(m.register_forward_hook(lambda mod, inp, out: taps.append(out.detach().double().cpu().numpy())))
    with torch.no_grad():
        xd = tr._img(x)
        img, mask = gen.decode(gen.encode_content(xd), s.cuda(), xd, return_mask=True)
    for h in hooks:
        h.remove()
    taps.append(mask.double().cpu().numpy())
    taps.append(img.double().cpu().numpy())
    return taps


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "male2female_council_folder.yaml"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", name)))
    cfg['council']['council_size'] = 1
    cfg['batch_size'] = batch
    cfg['iteration'] = 60000
    d = 'a2b' if cfg['do_a2b'] else 'b2a'
    O.seed_all(1)
    res = {}
    x, _ = O.synthetic_batch(batch, size)
    s = torch.randn(batch, cfg['gen']['style_dim'], 1, 1)
    sd = None
    for prec in ('split', 'fp32'):
        c = copy.deepcopy(cfg)
        c['cg_forward_precision'] = prec
        O.seed_all(1)
        tr = cga.Council_Trainer(c, 'cuda:0')
        gen = tr._nets('gen', d)[0]
        if sd is None:
            sd = O.to_numpy_state(gen.state_dict())
        tr.cuda('cuda:0')
        tr._ready()
        res['hip ' + prec] = ours_taps(tr, gen, x, s)
        del tr
    ref = oracle_taps(sd, cfg['gen'], x, s, torch.float64)
    res['oracle fp32'] = oracle_taps(sd, cfg['gen'], x, s, torch.float32)
    names = ["block %2d" % i for i in range(len(ref) - 2)] + ["mask", "image"]
    print("config %s %dx%d batch %d: l2-rel (max-abs/max) error vs the fp64 oracle after every Conv2dBlock" % (name, size, size, batch))
    print("%-10s %-18s " % ("", "shape") + " ".join("%22s" % k for k in res))
    for i, nm in enumerate(names):
        r = ref[i]
        row = []
        for k in res:
            a = res[k][i]
            l2 = np.sqrt(((a - r) ** 2).sum() / max((r ** 2).sum(), 1e-300))
            mx = np.abs(a - r).max() / max(np.abs(r).max(), 1e-300)
            row.append("%9.2e (%8.2e)" % (l2, mx))
        print("%-10s %-18s " % (nm, "x".join(map(str, r.shape))) + " ".join("%22s" % v for v in row))


if __name__ == "__main__":
    main()
