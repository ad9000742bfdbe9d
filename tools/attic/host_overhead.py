"""How long does the HOST need to enqueue one training iteration (no synchronisation inside)?  If this is close to the
GPU time per step, the step is launch-bound.  Usage: python tools/host_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import council_gan_amd as cga
from oracle import council_oracle as O
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "male2female_council_folder.yaml")))
cfg['council']['council_size'] = 4; cfg['batch_size'] = 4; cfg['iteration'] = 60000
O.seed_all(1)
tr = cga.Council_Trainer(cfg, 'cuda:0'); tr.cuda('cuda:0')
x_a, x_b = O.synthetic_batch(4, 256); x_a, x_b = x_a.cuda(), x_b.cuda()
def step():
    tr.dis_update(x_a, x_b, cfg); tr.dis_council_update(x_a, x_b, cfg); tr.gen_update(x_a, x_b, cfg, 60000)
for _ in range(2): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enqueue %.1f ms   until GPU done %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    torch.autograd.set_multithreading_enabled(False)      # backward nodes on this thread: visible to cProfile
    step(); torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
