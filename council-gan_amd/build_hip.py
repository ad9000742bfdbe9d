"""Builds the gfx950 shared library IN-TREE: council-gan_amd/lib/libcouncilgan_hip.so.

    python council-gan_amd/build_hip.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.  Every translation unit is
compiled to its own object (in parallel, cached by a digest of the file, the shared headers and the flags under lib/obj/) and
the objects are linked: an edit to one kernel file recompiles that file only."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ("conv_gemm.hip", "conv_x3.hip", "norm.hip", "elementwise.hip", "collective.hip", "head.hip")
SRC = [os.path.join(CSRC, f) for f in UNITS if os.path.exists(os.path.join(CSRC, f))]
# headers / include files every unit may see (a change recompiles everything)
SHARED = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))] + \
         [os.path.join(HERE, "..", "include", "council_gan_hip.h")]
DEPS = SRC + SHARED
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
OUT = os.path.join(OUT_DIR, "libcouncilgan_hip.so")
STAMP = OUT + ".stamp"
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("CG_HIPCC_FLAGS", "").split()
FLAGS = CFLAGS + ["-shared"]


def _digest(paths=DEPS):
    h = hashlib.sha256()
    for p in paths:
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(hipcc, src, verbose):
    dig = _digest([src] + SHARED)
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    tag = obj + ".stamp"
    if os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == dig:
        return obj
    cmd = [hipcc] + CFLAGS + ["-c", src, "-o", obj]
    if verbose:
        print("[build_hip]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(tag, "w").write(dig)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(len(SRC), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(hipcc, s, verbose), SRC))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", OUT]
    if verbose:
        print("[build_hip]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(STAMP, "w").write(dig)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
