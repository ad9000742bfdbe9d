"""Builds the gfx950 shared library IN-TREE: council-gan_amd/lib/libcouncilgan_hip.so.

    python council-gan_amd/build_hip.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("conv_gemm.hip", "norm.hip", "elementwise.hip", "collective.hip", "head.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "cg_common.h"), os.path.join(HERE, "csrc", "conv_x3.inc"), os.path.join(HERE, "..", "include", "council_gan_hip.h")]
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libcouncilgan_hip.so")
STAMP = OUT + ".stamp"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"] + os.environ.get("CG_HIPCC_FLAGS", "").split()


def _digest():
    h = hashlib.sha256()
    for p in DEPS:
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + SRC + ["-ldl", "-o", OUT]
    if verbose:
        print("[build_hip]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(STAMP, "w").write(dig)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
