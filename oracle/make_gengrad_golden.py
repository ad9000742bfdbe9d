"""TEST INFRASTRUCTURE -- generates tests/golden/pin_gengrad_b4.npz: the ORACLE side of the generator-gradient statistic at the
benchmark's own shape (BASELINE.json configs[2]: male2female 256x256, council 4, batch 4) for seeds {1, 2, 3}.

    python oracle/make_gengrad_golden.py [seed ...]        # ~10 minutes and ~45 GB of host memory per seed (fp32 + fp64 oracle)

Why a fixture: the fp32 oracle and its fp64 twin cost ~7 minutes of host time per seed at this shape, which kept
tests/test_gpu_parity_full.py::test_bench_batch_generator_gradient_ratio opt-in and out of the driver's run (missing 3).
With the oracle side committed, the GPU side is two HIP iterations per seed (seconds).

What is stored per seed and generator (member): the fp64 generator gradient on a fixed, seeded SUBSAMPLE of every tensor
(tests/golden_util.py::grad_subsample_index: >= 256 elements or 1 / 256 of the tensor, whichever is larger; stored as fp32 --
6e-8 relative, four orders below the errors measured), every tensor's full squared norm, the fp32 oracle's own error against
fp64 computed BOTH on the full tensors and with the subsample estimator (so the estimator's accuracy is on record), the fp32
oracle's generator losses, and a per-tensor checksum of the initial weights (the GPU side re-derives them from the seed:
tests/test_host_cpu.py::test_trainer_init_matches_reference)."""
import copy
import os
import random
import resource
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import council_oracle as O  # noqa: E402
import golden_util as GU  # noqa: E402
import parity_util as P  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "pin_gengrad_b4.npz")
SIZE, BATCH, COUNCIL = 256, 4, 4


def one_seed(seed, out):
    import council_gan_amd as cga
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
    cfg['council']['council_size'] = COUNCIL
    cfg['iteration'] = 60000
    cfg['batch_size'] = BATCH
    cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = SIZE
    O.seed_all(seed)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')          # host-side construction: the reference's RNG stream
    state = P.host_state(tr)
    del tr
    x_a, x_b = O.synthetic_batch(BATCH, SIZE, seed=GU.gengrad_image_seed(seed))
    rng = (random.getstate(), torch.get_rng_state())
    pre = "s%d/" % seed
    t0 = time.time()
    o32, g32, _, _ = P.run_oracle(cfg, state, x_a, x_b, rng, torch.float32)
    t1 = time.time()
    out[pre + "loss_gen_total"] = P.lossvec(o32.loss_gen_total)
    g32 = {k: v for k, v in g32.items() if k[0] == "gen"}
    del o32
    _, g64, _, _ = P.run_oracle(cfg, state, x_a, x_b, rng, torch.float64)
    t2 = time.time()
    for d in state:
        for i in range(COUNCIL):
            k = ("gen", d, i)
            names = sorted(g64[k])
            sub64, sub32, norms, numels = [], [], [], []
            for n in names:
                a64, a32 = g64[k][n].reshape(-1), g32[k][n].reshape(-1)
                idx = GU.grad_subsample_index(n, a64.size)
                sub64.append(a64[idx].astype(np.float32))
                sub32.append(a32[idx].astype(np.float32))
                norms.append(float((a64.astype(np.float64) ** 2).sum()))
                numels.append(a64.size)
            mp = pre + "%s/%d/" % (d, i)
            out[mp + "names"] = np.array(names)
            out[mp + "numel"] = np.array(numels, dtype=np.int64)
            out[mp + "norm2_64"] = np.array(norms, dtype=np.float64)
            out[mp + "sub64"] = np.concatenate(sub64)
            e_full = P.l2rel(g32[k], g64[k])
            e_sub = GU.subsample_l2rel({n: g32[k][n] for n in names}, names, numels, np.concatenate(sub64), np.array(norms))
            out[mp + "err_ref_full"] = np.float64(e_full)
            out[mp + "err_ref_sub"] = np.float64(e_sub)
            init = state[d]['gen'][i]
            out[mp + "init_sum"] = np.array([float(np.asarray(init[n], dtype=np.float64).sum()) for n in sorted(init)])
            print("seed %d %s/%d: fp32 oracle vs fp64 %.3e (full)  %.3e (subsample estimator, %d of %d values)"
                  % (seed, d, i, e_full, e_sub, sum(len(s) for s in sub64), sum(numels)), flush=True)
    print("seed %d: fp32 oracle %.0f s, fp64 oracle %.0f s, peak RSS %.1f GB" % (seed, t1 - t0, t2 - t1,
          resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)


def main():
    seeds = [int(s) for s in sys.argv[1:]] or [1, 2, 3]
    out = {}
    if os.path.exists(OUT):                       # seeds can be (re)generated one at a time
        z = np.load(OUT)
        out = {k: z[k] for k in z.files if not any(k.startswith("s%d/" % s) for s in seeds)}
    for s in seeds:
        one_seed(s, out)
        out["seeds"] = np.array(sorted({int(k[1:k.index('/')]) for k in out if k.startswith("s") and '/' in k}))
        np.savez_compressed(OUT, **out)
        print("wrote", OUT, os.path.getsize(OUT), "bytes", flush=True)


if __name__ == "__main__":
    main()
