"""Calibration of bench.py's `cpu_baseline` ("port": the oracle) against the REAL reference, in the build container (needs /root/reference):
the same iteration -- male2female 256x256, council 4, batch 4, dis_update + dis_council_update + gen_update of all members -- timed on the
oracle (oracle/council_oracle.py) and on the unmodified reference trainer (through oracle/ref_shim.py), same threads, same inputs.

    python tools/cpu_calibrate.py [batch=4] [timed iterations=1]

Prints seconds per iteration for both and the ratio; BASELINE-style prose of the result lives in DESIGN.md section 5."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
import yaml  # noqa: E402
from oracle import council_oracle as O  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
timed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
cfg['council']['council_size'] = 4
cfg['batch_size'] = batch
cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = 256
cfg['iteration'] = 60000
x_a, x_b = O.synthetic_batch(batch, 256)
print("threads", torch.get_num_threads(), "cores", os.cpu_count(), "batch", batch, flush=True)

Trainer = ref_shim.reference_trainer_cls()
O.seed_all(1)
ref = Trainer(copy.deepcopy(cfg), 'cpu')
state = {'a2b': {'gen': [O.to_numpy_state(m.state_dict()) for m in ref.gen_a2b_s],
                 'dis': [O.to_numpy_state(m.state_dict()) for m in ref.dis_a2b_s],
                 'dis_council': [O.to_numpy_state(m.state_dict()) for m in ref.dis_council_a2b_s]}}


def time_it(step, name):
    t0 = time.time()
    step()
    warm = time.time() - t0
    s = []
    for _ in range(timed):
        t0 = time.time()
        step()
        s.append(time.time() - t0)
    print("%-10s warm-up %.1f s, timed %s s per iteration -> %.4f images/s" % (name, warm, ["%.1f" % v for v in s], batch / min(s)), flush=True)
    return min(s)


def ref_step():
    c = copy.deepcopy(cfg)
    ref.dis_update(x_a, x_b, c)
    ref.dis_council_update(x_a, x_b, c)
    ref.gen_update(x_a, x_b, c, c['iteration'])


t_ref = time_it(ref_step, "reference")
del ref
otr = O.OracleTrainer(copy.deepcopy(cfg), state)


def ora_step():
    otr.dis_update(x_a, x_b, cfg)
    otr.dis_council_update(x_a, x_b, cfg)
    otr.gen_update(x_a, x_b, cfg, cfg['iteration'])


t_ora = time_it(ora_step, "oracle")
print("oracle / reference time per iteration = %.3f" % (t_ora / t_ref))
