"""TEST INFRASTRUCTURE -- generates tests/golden/pin_layernorm.npz from the REAL reference (build container only).

Pins oracle/council_oracle.py::layer_norm to the reference's LayerNorm (networks.py:659-686) -- BOTH branches of its
forward (batch 1: statistics over the flattened tensor, `:673-676`; batch > 1: per sample, `:677-679`; unbiased std, eps added to
the std) -- and the `dis.norm: ln` discriminators that contain it (MsImageDis / MsImageDisCouncil with norm='ln',
networks.py:24-47, 119-146): inputs, parameters, outputs and the gradients of an LSGAN-style scalar, as computed by the
reference's own modules on this container's CPU PyTorch.

    python oracle/make_ln_golden.py

Consumed by tests/test_oracle_golden.py::test_layer_norm_pinned_to_reference (runs everywhere, no reference tree needed)."""
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "pin_layernorm.npz")


def main():
    ref_shim.install()
    import networks as RN  # the reference's networks.py

    out = {}
    torch.manual_seed(11)
    for tag, shape in (("b1", (1, 6, 5, 7)), ("b3", (3, 6, 5, 7))):
        ln = RN.LayerNorm(shape[1])
        with torch.no_grad():
            ln.beta.copy_(torch.randn(shape[1]) * 0.3)
        x = (torch.randn(*shape) * 1.7 + 0.4).requires_grad_(True)
        y = ln(x)
        w = torch.randn(*shape)
        (y * w).sum().backward()
        for k, v in (("x", x), ("gamma", ln.gamma), ("beta", ln.beta), ("y", y), ("w", w), ("dx", x.grad),
                     ("dgamma", ln.gamma.grad), ("dbeta", ln.beta.grad)):
            out["ln/%s/%s" % (tag, k)] = v.detach().numpy().copy()

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))
    hp = dict(cfg['dis'])
    hp.update(dim=4, norm='ln')
    for cls_name, key in (("MsImageDis", "dis"), ("MsImageDisCouncil", "dis_council")):
        torch.manual_seed(5)
        net = getattr(RN, cls_name)(3, hp, cuda_device='cpu')
        x = torch.rand(2, 3, 32, 32) * 2 - 1
        xin = torch.rand(2, 3, 32, 32) * 2 - 1
        outs = net(x) if key == "dis" else net(x, xin)
        loss = sum(torch.mean((o - 1) ** 2) for o in outs)
        loss.backward()
        out[key + "/x"] = x.numpy()
        out[key + "/x_input"] = xin.numpy()
        out[key + "/loss"] = np.float32(loss.item())
        for i, o in enumerate(outs):
            out[key + "/out/%d" % i] = o.detach().numpy()
        for k, v in net.state_dict().items():
            out[key + "/sd/" + k] = v.detach().numpy().copy()
        for k, p in net.named_parameters():
            out[key + "/grad/" + k] = p.grad.numpy().copy()
    import json
    out["dis_hp_json"] = np.frombuffer(json.dumps(hp).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
