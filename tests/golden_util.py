"""Loader for tests/golden/*.npz (written by oracle/make_golden.py from the real reference)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


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

    def init_state(self):
        """{'a2b': {'gen': [sd, ...], 'dis': [...], 'dis_council': [...]}, ...}"""
        st = {}
        for d in self.dirs:
            st[d] = {}
            for net in self.nets:
                st[d][net] = [self.sub("init/%s/%s/%d/" % (d, net, i)) for i in range(self.C)]
        return st


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
