"""TEST INFRASTRUCTURE (build container only: needs /root/reference).  Run as a subprocess by
tests/test_ref_interop_cpu.py because the import shim patches torch globally.

Checkpoint interoperability with the REAL reference, both directions:
  1. the reference's own checkpoint set (from tests/golden/ckpt_c2.npz) is loaded by OUR resume();
  2. OUR save() writes a new set from that state;
  3. the reference's resume() (trainer_council.py:898-967) loads OUR files;
and every weight, Adam moment, step count and hyper-parameter must equal what the reference holds after resuming
from its own files."""
import copy
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_util import Golden  # noqa: E402


def main():
    g = Golden("ckpt_c2")
    cfg = g.cfg
    import council_gan_amd as cga
    with tempfile.TemporaryDirectory() as dir_ref, tempfile.TemporaryDirectory() as dir_ours:
        for k in g.z.files:
            if k.startswith("ckpt/"):
                open(os.path.join(dir_ref, k[5:]), "wb").write(g[k].tobytes())
        cga.seed_everything(5)
        ours = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')        # stays on the host: save / resume are file I/O
        it = ours.resume(dir_ref, cfg)
        assert it == int(g["resume/iterations"]), it
        ours.save(dir_ours, it - 1)
        assert sorted(os.listdir(dir_ours)) == sorted(os.listdir(dir_ref)), (os.listdir(dir_ours), os.listdir(dir_ref))
        # file sizes: independent contiguous tensors, not views into the flat optimizer buffer
        for fn in os.listdir(dir_ref):
            a, b = os.path.getsize(os.path.join(dir_ref, fn)), os.path.getsize(os.path.join(dir_ours, fn))
            assert abs(a - b) <= 0.05 * a + 4096, (fn, a, b)

        from oracle import ref_shim
        Trainer = ref_shim.reference_trainer_cls()
        refs = []
        for seed, d in ((11, dir_ref), (12, dir_ours)):
            torch.manual_seed(seed)
            t = Trainer(copy.deepcopy(cfg), 'cpu')
            assert t.resume(d, cfg) == it
            refs.append(t)
        a, b = refs
        n = 0
        for attr in ('gen_a2b_s', 'dis_a2b_s', 'dis_council_a2b_s'):
            for ma, mb in zip(getattr(a, attr), getattr(b, attr)):
                sa, sb = ma.state_dict(), mb.state_dict()
                assert list(sa) == list(sb)
                for k in sa:
                    assert torch.equal(sa[k], sb[k]), (attr, k)
                    n += 1
        for attr in ('gen_opt_s', 'dis_opt_s', 'dis_council_opt_s'):
            for oa, ob in zip(getattr(a, attr), getattr(b, attr)):
                sa, sb = oa.state_dict(), ob.state_dict()
                assert sorted(sa['state']) == sorted(sb['state']), attr
                for i in sa['state']:
                    for k, v in sa['state'][i].items():
                        w = sb['state'][i][k]
                        assert (torch.equal(v, w) if torch.is_tensor(v) else v == w), (attr, i, k)
                        n += 1
                for ga, gb in zip(sa['param_groups'], sb['param_groups']):
                    for k in ('lr', 'betas', 'eps', 'weight_decay', 'amsgrad', 'initial_lr'):
                        assert ga.get(k) == gb.get(k), (attr, k, ga.get(k), gb.get(k))
        print("INTEROP_OK %d tensors" % n)


if __name__ == "__main__":
    main()
