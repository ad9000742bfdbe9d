"""Which ATen compute kernels run inside one benchmark iteration, and which Python lines launch them (torch profiler with stacks):
    python tools/aten_rows.py      (male2female 256x256, council 4, batch 4 -- bench.py's default workload)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import yaml  # noqa: E402
import council_gan_amd as cga  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
cfg['council']['council_size'] = 4
cfg['batch_size'] = 4
cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = 256
cfg['iteration'] = 60000
cga.seed_everything(1)
tr = cga.Council_Trainer(cfg, 'cuda:0')
tr.cuda('cuda:0')
x_a, x_b = cga.synthetic_batch(4, 256)
x_a, x_b = x_a.cuda(), x_b.cuda()


def step():
    tr.dis_update(x_a, x_b, cfg); tr.dis_council_update(x_a, x_b, cfg); tr.gen_update(x_a, x_b, cfg, 60000)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA or not ev.name.startswith("aten::"):
        continue
    dev = sum(k.duration for k in ev.kernels) if ev.kernels else 0
    if dev <= 0:
        continue
    stack = [f for f in (ev.stack or []) if "council-gan_amd" in f or "bench.py" in f]
    site = stack[0].split("council-gan_amd/")[-1] if stack else "(autograd engine / no python frame)"
    node, par = "", ev.cpu_parent
    while par is not None:
        if "evaluate_function" in par.name or par.name.endswith("Backward"):
            node = par.name.split(": ")[-1]
            break
        par = par.cpu_parent
    shapes = str([tuple(x) for x in (ev.input_shapes or []) if x])
    key = (ev.name, site + "  node=" + node + " shapes=" + shapes)
    c, t = rows.get(key, (0, 0.0))
    rows[key] = (c + 1, t + dev)
print("%-28s %5s %9s  %s" % ("aten op", "calls", "device us", "first council-gan_amd frame"))
for (name, site), (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %5d %9.1f  %s" % (name, c, t, site[:150]))

# Second view: the Python call sites of the ATen operators issued from the MAIN thread (forward passes, loss assembly; the autograd
# engine's own accumulations run on its worker threads and show up only in the table above).
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEW_ONLY = ("view", "reshape", "select", "slice", "detach", "alias", "as_strided", "expand", "permute", "transpose", "t.", "unsqueeze",
             "squeeze", "empty", "_unsafe_view", "unbind", "split", "narrow", "is_", "size", "stride", "record_stream", "_local_scalar")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW_ONLY):
            fr = [f for f in traceback.extract_stack() if "council-gan_amd" in f.filename]
            site = "%s:%d" % (fr[-1].filename.split("council-gan_amd/")[-1], fr[-1].lineno) if fr else "(tool)"
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            key = (name, site, str(shapes))
            self.sites[key] = self.sites.get(key, 0) + 1
        return func(*args, **(kwargs or {}))


with Sites() as sites:
    step()
    torch.cuda.synchronize()
print("\nATen operators dispatched from the main thread in one iteration (views / allocations left out):")
for (name, site, shapes), c in sorted(sites.sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%-34s %4d  %-28s %s" % (name, c, site, shapes[:90]))
