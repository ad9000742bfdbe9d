"""Loader for tests/golden/*.npz (written by oracle/make_golden.py from the real reference)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


CKPT_CASES = ("ckpt_c2",)       # fixtures built around a reference-written checkpoint set (oracle/make_golden.py CKPT)


def case_names(kind="train"):
    """Fixture names.  "train": the two-iteration training cases (including the full-width one, whose initial weights
    are re-derived from the seed); "ckpt": the reference-checkpoint cases; "all": both."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    # input_pil.npz: the loader-tail fixture; pin_*.npz: operator / statistic pins (LayerNorm, bench-shape gradients) -- not trainer cases
    names = [n for n in names if not n.startswith(("input_", "pin_"))]
    if kind == "all":
        return names
    return [n for n in names if (n in CKPT_CASES) == (kind == "ckpt")]


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.cfg = json.loads(bytes(self.z["config_json"]).decode())
        self.C = self.cfg["council"]["council_size"]
        self.dirs = [d for d in ("a2b", "b2a") if self.cfg["do_" + d]]
        self.nets = ["gen", "dis"] + (["dis_council"] if self.cfg["council_w"] != 0 else [])

    def __contains__(self, k):
        return k in self.z.files

    def __getitem__(self, k):
        return self.z[k]

    def sub(self, prefix):
        """All arrays under `prefix` as {suffix: array}."""
        return {k[len(prefix):]: self.z[k] for k in self.z.files if k.startswith(prefix)}

    @property
    def from_seed(self):
        return "init_from_seed" in self.z.files

    def init_state(self):
        """{'a2b': {'gen': [sd, ...], 'dis': [...], 'dis_council': [...]}, ...}.  Full-width fixtures do not store their
        initial weights: the HIP trainer's host-side constructor under the reference's seeds (train.py:55-62) reproduces
        them bit for bit (tests/test_host_cpu.py::test_trainer_init_matches_reference); the per-tensor summary stored
        in the fixture guards that assumption."""
        if self.from_seed:
            return self._seeded_state()
        st = {}
        for d in self.dirs:
            st[d] = {}
            for net in self.nets:
                st[d][net] = [self.sub("init/%s/%s/%d/" % (d, net, i)) for i in range(self.C)]
        return st


def _seeded(self):
    import copy
    import council_gan_amd as cga
    cga.seed_everything(int(self.z["init_from_seed"]))
    tr = cga.Council_Trainer(copy.deepcopy(self.cfg), 'cuda:0')          # host-side construction only: no GPU involved
    attr = {"gen": "gen_%s_s", "dis": "dis_%s_s", "dis_council": "dis_council_%s_s"}
    st = {}
    for d in self.dirs:
        st[d] = {}
        for net in self.nets:
            st[d][net] = [{k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
                          for m in getattr(tr, attr[net] % d)]
            for i, sd in enumerate(st[d][net]):
                ref = self.z["initsum/%s/%s/%d" % (d, net, i)]
                assert np.array_equal(summary(sd), ref), "seeded initial weights differ from the reference's (%s %s %d)" % (d, net, i)
    return st


Golden._seeded_state = _seeded


def summary(d):
    keys = sorted(d)
    return np.array([[float(np.asarray(d[k], dtype=np.float64).sum()),
                      float(np.sqrt((np.asarray(d[k], dtype=np.float64) ** 2).sum())),
                      float(np.abs(np.asarray(d[k])).max())] for k in keys], dtype=np.float64)


def rel_err(a, b):
    """max-abs difference over max-abs reference (the 'max-abs / max-abs' criterion of SURVEY section 7)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / den)


# ---- generator-gradient statistic at the benchmark's shape (tests/golden/pin_gengrad_b4.npz, oracle/make_gengrad_golden.py) ----
def gengrad_image_seed(seed):
    """Seed of the synthetic batch of trainer seed `seed` (seed 1 -> 7: SURVEY.md 8d's batch, the one bench.py uses)."""
    return 6 + seed


def grad_subsample_index(name, numel):
    """The fixed subsample of a gradient tensor both sides of the fixture use: sorted indices into the flattened tensor
    (physical order of the state_dict tensor), >= 256 elements or 1 / 256 of it, whichever is larger, the whole tensor when it
    is that small; drawn from a generator seeded by the tensor's name and size."""
    import zlib
    n_sub = min(numel, max(256, numel // 256))
    if n_sub == numel:
        return np.arange(numel)
    rs = np.random.RandomState(zlib.crc32(("%s:%d" % (name, numel)).encode()) & 0x7fffffff)
    return np.sort(rs.choice(numel, size=n_sub, replace=False))


def subsample_l2rel(grads, names, numels, sub64, norm2_64):
    """Estimator of || g - g64 || / || g64 || over all tensors of one generator from the subsample: every tensor's squared
    error on its subsample is scaled by numel / subsample size; the denominator is the stored FULL norm."""
    num, off = 0.0, 0
    for n, numel in zip(names, numels):
        idx = grad_subsample_index(str(n), int(numel))
        ref = sub64[off:off + len(idx)].astype(np.float64)
        off += len(idx)
        got = np.asarray(grads[str(n)]).reshape(-1)[idx].astype(np.float64)
        num += float(((got - ref) ** 2).sum()) * (int(numel) / len(idx))
    assert off == len(sub64)
    return float(np.sqrt(num) / max(np.sqrt(float(np.sum(norm2_64))), 1e-30))
