"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import shim that lets the *unmodified* reference (`/root/reference`) run on this
container's CPU PyTorch.  It exists for exactly one purpose: generating the golden
fixtures under `tests/golden/` (see `oracle/make_golden.py`) that pin the oracle
restatement in `oracle/council_oracle.py`.

`/root/reference` does not exist on the GPU box, so nothing on a `-m gpu` test,
`smoke()` or `bench.py` path may import this module.

What has to be patched, and why (reference file:line):
  * `networks.py:35,144,238-254` call `.cuda(self.cuda_device)` on modules and
    `networks.py:50,148,258,279,286` call `x.cuda(...)` on tensors -> identity on CPU.
  * `utils.py:6,8,13,19` import torchfile / torchvision (absent here),
    `trainer_council.py:17` imports torchvision.transforms.functional (unused) -> stub modules.
  * `train.py:251` calls `torch.cuda.synchronize` -> no-op.
  * `trainer_council.py:520,578` push `loss.detach().cpu().numpy()` into the loss-matching deques
    and then scale the same loss IN PLACE (`:581,589`).  On CUDA `.cpu()` copies, so the deque
    keeps the unscaled loss; on a CPU tensor `.cpu()` is the identity, the numpy array aliases
    the loss and the deque entry silently becomes loss * w_match * council_w.  The authors ran on
    CUDA, so the shim makes `.cpu()` copy (device semantics), otherwise iteration >= 2 of the
    CPU run diverges from what the reference computes on a GPU.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name):
    m = types.ModuleType(name)
    m.__dict__["__path__"] = []
    sys.modules[name] = m
    return m


def install():
    """Patch torch + sys.modules, put the reference on sys.path. Idempotent."""
    import os
    import torch
    import torch.nn as nn

    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree %s not present (GPU box?) -- the shim is "
                           "container-only test infrastructure" % REFERENCE_ROOT)
    if getattr(install, "_done", False):
        return
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()       # device->host copy semantics
    torch.cuda.manual_seed = lambda *a, **k: None
    for name in ("torchfile", "tensorboardX", "termcolor", "torchvision",
                 "torchvision.models", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchvision.utils"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["torchvision.models"].inception_v3 = None
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True


def reference_trainer_cls():
    install()
    from trainer_council import Council_Trainer  # noqa: reference module
    return Council_Trainer


def reference_networks():
    install()
    import networks  # noqa: reference module
    return networks
