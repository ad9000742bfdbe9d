"""Where does the generator-gradient error of the HIP path come from?  The smooth-loss backward-chain comparison
(tests/parity_util.py: smooth_backward_errors) under different datapaths, per tensor.
Usage: python tools/diag_smooth.py [size=64] [batch=2]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import yaml  # noqa: E402
import council_gan_amd as cga  # noqa: E402
from council_gan_amd import ops  # noqa: E402
import parity_util as P  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
cfg['council']['council_size'] = 2
cfg['iteration'] = 60000
ORDER = None


def show(tag, errs, fwd):
    global ORDER
    if ORDER is None:
        ORDER = [k for (m, k) in errs if m == 0]
    print("== %s   forward (image, mask) err: %s" % (tag, {m: ("%.1e" % a, "%.1e" % b) for m, (a, b) in fwd.items()}))
    for k in ORDER:
        print("   %-46s %s" % (k, "  ".join("%.2e" % errs[(m, k)] for m in sorted({m for m, _ in errs}))))
    sys.stdout.flush()


modes = [("split, two members per launch, gen.activ = tanh (smooth network)", {}, {'activ': 'tanh'}),
         ("exact fp32 datapath, gen.activ = tanh", {'cg_forward_precision': 'fp32'}, {'activ': 'tanh'}),
         ("split, two members per launch", {}, {}),
         ("exact fp32 datapath", {'cg_forward_precision': 'fp32'}, {}),
         ("split, member by member", {}, {'group_max': 1})]
for tag, over, kw in modes:
    c = copy.deepcopy(cfg)
    c.update(over)
    show(tag, *P.smooth_backward_errors(cga, c, size=size, batch=batch, **kw))
# split forward, fp32 backward
orig = cga.Council_Trainer._ready


def ready_fp32_bwd(self):
    orig(self)
    ops.X3_BACKWARD = False


cga.Council_Trainer._ready = ready_fp32_bwd
try:
    show("split forward, exact-fp32 backward", *P.smooth_backward_errors(cga, copy.deepcopy(cfg), size=size, batch=batch))
finally:
    cga.Council_Trainer._ready = orig
    ops.X3_BACKWARD = True
