"""How chaotic is the generator gradient with respect to forward round-off?  CPU only (oracle only; development aid).

The fp32 oracle is run K times with every Conv2dBlock output multiplied by (1 + eps * N(0,1)) -- a stand-in for "the same
arithmetic with a different summation order" at a per-layer noise level eps -- and the generator-gradient error against
the fp64 oracle is compared with the unperturbed fp32 oracle's own error:

    python tools/diag_gengrad_lottery.py [config yaml] [size] [council] [batch] [eps] [K]

Prints, per council member, err(fp32)/1 and the K ratios err(perturbed fp32) / err(fp32)."""
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
    eps = float(sys.argv[5]) if len(sys.argv) > 5 else 3e-7
    K = int(sys.argv[6]) if len(sys.argv) > 6 else 8
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", name)))
    cfg['council']['council_size'] = council
    cfg['iteration'] = 60000
    cfg['batch_size'] = batch
    cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = size
    O.seed_all(1)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')      # host-side construction only
    state = P.host_state(tr)
    d = tr._dirs[0]
    x_a, x_b = O.synthetic_batch(batch, size)
    rng = (random.getstate(), torch.get_rng_state())
    _, g64, _, _ = P.run_oracle(cfg, state, x_a, x_b, rng, torch.float64)
    _, g32, _, _ = P.run_oracle(cfg, state, x_a, x_b, rng, torch.float32)
    base = [P.l2rel(g32[('gen', d, i)], g64[('gen', d, i)]) for i in range(council)]
    print("config %s %dx%d council %d batch %d, per-layer relative noise %.1e" % (name, size, size, council, batch, eps))
    print("fp32 oracle vs fp64:", ["%.2e" % b for b in base])
    orig = O.conv_block
    gen_noise = torch.Generator()
    ratios = [[] for _ in range(council)]
    for k in range(K):
        gen_noise.manual_seed(1000 + k)

        def noisy(*a, **kw):
            y = orig(*a, **kw)
            if y.dtype == torch.float32:
                y = y * (1.0 + eps * torch.randn(y.shape, generator=gen_noise))
            return y
        O.conv_block = noisy
        try:
            _, gp, _, _ = P.run_oracle(cfg, state, x_a, x_b, rng, torch.float32)
        finally:
            O.conv_block = orig
        for i in range(council):
            ratios[i].append(P.l2rel(gp[('gen', d, i)], g64[('gen', d, i)]) / base[i])
        print("  draw %d:" % k, " ".join("%.2f" % r[-1] for r in ratios), flush=True)
    allr = np.array([v for r in ratios for v in r])
    print("ratio err(perturbed)/err(fp32): median %.2f  90th pct %.2f  max %.2f" % (np.median(allr), np.percentile(allr, 90), allr.max()))


if __name__ == "__main__":
    main()
