"""Where does the generator-gradient error come from?  One iteration (HIP trainer vs fp32 oracle vs fp64 oracle, same
weights / inputs / host RNG) with a per-tensor breakdown of err(ours, fp64) against err(fp32 oracle, fp64):

    python tools/diag_gengrad.py [config yaml] [size] [council] [batch]

Development aid (uses the oracle: never part of the product path)."""
import copy
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

import council_gan_amd as cga  # noqa: E402
import parity_util as P  # noqa: E402
from oracle import council_oracle as O  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "male2female_council_folder.yaml"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    council = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", name)))
    cfg['council']['council_size'] = council
    cfg['iteration'] = 60000
    cfg['batch_size'] = batch
    cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = size
    O.seed_all(1)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    state = P.host_state(tr)
    tr.cuda('cuda:0')
    x_a, x_b = O.synthetic_batch(batch, size)
    rng = (random.getstate(), torch.get_rng_state())
    tr.dis_update(x_a, x_b, cfg)
    tr.dis_council_update(x_a, x_b, cfg)
    tr.gen_update(x_a, x_b, cfg, cfg['iteration'])
    torch.cuda.synchronize()
    d = tr._dirs[0]
    ours = {i: P.grads_of(tr._nets('gen', d)[i]) for i in range(council)}
    _, g32, _, _ = P.run_oracle(cfg, state, x_a, x_b, rng, torch.float32)
    _, g64, _, _ = P.run_oracle(cfg, state, x_a, x_b, rng, torch.float64)
    print("config %s %dx%d council %d batch %d  precision %s  group %s" %
          (name, size, size, council, batch, "split" if tr._split_fwd else "fp32", tr._groups[1]))
    for i in range(council):
        a, r32, r64 = ours[i], g32[('gen', d, i)], g64[('gen', d, i)]
        tot = sum(float((r64[k].astype(np.float64) ** 2).sum()) for k in r64)
        print("member %d: l2rel ours %.3e  fp32-oracle %.3e" % (i, P.l2rel(a, r64), P.l2rel(r32, r64)))
        rows = []
        for k in r64:
            e_o = float(((a[k].astype(np.float64) - r64[k]) ** 2).sum())
            e_r = float(((r32[k].astype(np.float64) - r64[k]) ** 2).sum())
            n2 = float((r64[k].astype(np.float64) ** 2).sum())
            rows.append((e_o / tot, k, np.sqrt(e_o / max(n2, 1e-300)), np.sqrt(e_r / max(n2, 1e-300)), np.sqrt(n2 / tot)))
        rows.sort(reverse=True)
        print("  %-46s %10s %10s %10s %10s" % ("tensor (by share of ours' squared error)", "share", "ours rel", "fp32 rel", "|g| share"))
        for sh, k, eo, er, ns in rows[:14]:
            print("  %-46s %10.3e %10.3e %10.3e %10.3e" % (k, sh, eo, er, ns))


if __name__ == "__main__":
    main()
