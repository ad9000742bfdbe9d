"""Eager vs hipGraph mode, iteration by iteration: which logged value differs first?  (development aid)
Usage: python tools/diag_graph.py [warmup iterations before capture, default 1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402
import test_gpu_graph as T  # noqa: E402

os.environ['CG_DIAG_POOLS'] = '1'
os.environ['CG_DIAG_DUMP_IT'] = '2'
if len(sys.argv) > 1:
    os.environ['CG_GRAPH_WARMUP'] = sys.argv[1]
cfg = T._tiny("male2female_council_folder.yaml", 2, 1)
cfg['do_b2a'] = True
if os.environ.get('DIAG_FP32'):
    cfg['cg_forward_precision'] = 'fp32'
if os.environ.get('DIAG_ONE_DIR'):
    cfg['do_b2a'] = False
if os.environ.get('DIAG_FULL'):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
    cfg['council']['council_size'] = 4
    cfg['batch_size'] = 1
    cfg['iteration'] = 60000
NIT, SIZE = int(os.environ.get('DIAG_ITERS', '5')), int(os.environ.get('DIAG_SIZE', '64'))
e = T._run(cga, cfg, False, NIT, SIZE)
g = T._run(cga, cfg, True, NIT, SIZE)
print("captured segments:", g[4])
for it, (a, b) in enumerate(zip(e[0], g[0])):
    if '_dump' in a:
        da, db = a.pop('_dump'), b.pop('_dump')
        for k in da:
            x, y = da[k].double(), db[k].double()
            if not torch.equal(x, y):
                print("   grad %-58s rel diff %.2e  (|eager| %.3e)" % (k, float((x - y).norm() / max(float(x.norm()), 1e-30)), float(x.norm())))
    for n in a:
        if a[n] != b[n] or it < 1:
            print("it %d %-26s %s  eager %s  graph %s" % (it, n, "same" if a[n] == b[n] else "DIFF", a[n], b[n]))
bad = [k for k in e[1] if not torch.equal(e[1][k], g[1][k])]
print("weights differing after 5 iterations: %d of %d" % (len(bad), len(e[1])), bad[:6])
print("steps equal:", e[2] == g[2], " ring equal:", e[3] == g[3])
