"""Which tensors go through a split pass of their own (fp32 -> {hi, lo} planes: ops.split_f16_dynamic / act_bwd_split / split_f16) in one
benchmark iteration, with shape, bytes and the Python call chain -- the list a fused producer epilogue would have to cover
(bounded split):    python tools/split_sites.py      (male2female 256x256, council 4, batch 4)"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import council_gan_amd as cga
from council_gan_amd import ops
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
cfg['council']['council_size'] = 4; cfg['batch_size'] = 4
cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = 256
cfg['iteration'] = 60000
cga.seed_everything(1)
tr = cga.Council_Trainer(cfg, 'cuda:0'); tr.cuda('cuda:0')
x_a, x_b = cga.synthetic_batch(4, 256); x_a, x_b = x_a.cuda(), x_b.cuda()
def step():
    tr.dis_update(x_a, x_b, cfg); tr.dis_council_update(x_a, x_b, cfg); tr.gen_update(x_a, x_b, cfg, 60000)
for _ in range(3): step()
torch.cuda.synchronize()
sites = collections.Counter(); bytes_ = collections.Counter()
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if 'council-gan_amd' in x.filename]
        key = (name, tuple(a[0].shape), ' < '.join('%s:%d' % (x.filename.split('/')[-1], x.lineno) for x in fr[-3:]))
        sites[key] += 1; bytes_[key] += a[0].numel() * 4
        return orig(*a, **k)
    setattr(ops, name, f)
for n in ('split_f16_dynamic', 'act_bwd_split', 'split_f16'):
    wrap(n)
step(); torch.cuda.synchronize()
for k, c in sorted(sites.items(), key=lambda kv: -bytes_[kv[0]]):
    print("%-18s %-22s n=%3d %7.1f MB  %s" % (k[0], k[1], c, bytes_[k] / 1e6, k[2]))
