"""Run by tests/test_ref_interop_cpu.py in a subprocess (the reference import shim patches torch process-wide): the oracle's
generator against the REAL reference's AdaINGen (networks.py:223-330) for every activation of gen.activ the hot path supports
(networks.py:494-507: relu, lrelu, tanh) -- the shipped configs only use relu, the smooth-network parity test uses tanh."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
import yaml  # noqa: E402
import networks as RN  # noqa: E402  (the reference's networks.py)
from oracle import council_oracle as O  # noqa: E402

cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
for activ in ("relu", "lrelu", "tanh"):
    hp = dict(cfg['gen'])
    hp.update(dim=16, mlp_dim=32, n_res=2, activ=activ)
    torch.manual_seed(3)
    ref = RN.AdaINGen(3, hp, cuda_device='cpu')
    x = torch.rand(2, 3, 32, 32) * 2 - 1
    s = torch.randn(2, hp['style_dim'], 1, 1)
    with torch.no_grad():
        c, s_fake = ref.encode(x)
        y_ref = ref.decode(c, s, x)
        og = O.OracleGen({k: v.clone() for k, v in ref.state_dict().items()}, hp)
        c_or, s_or = og.encode(x)
        y_or = og.decode(c_or, s, x)
    for name, a, b in (("content", c, c_or), ("style", s_fake, s_or), ("image", y_ref, y_or)):
        d = float((a - b).abs().max())
        assert d <= 1e-6 * max(1.0, float(a.abs().max())), (activ, name, d)
print("ACTIVATIONS_OK")
