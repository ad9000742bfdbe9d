"""Whole-iteration numerics of candidate convolution datapaths, emulated on the CPU (development aid; uses the oracle,
never part of the product path; no GPU).

One train.py:237-250 iteration (dis_update + dis_council_update + gen_update) of the fp64 oracle, with every convolution
whose contraction runs over a multiple of 32 channels replaced -- forward, data gradient and weight gradient -- by an
emulation of the split-precision product  a.b ~ ah.bh + X(ah).X(bl) + X(al).X(bh)  evaluated in float64 on QUANTISED
operands ({ah, al} = fp16 planes under a per-tensor power-of-two scale, as shipped):

    x3      X = identity                      (the shipped fp16 x 3 datapath: 3 fp16 MFMA passes)
    fp8     X = fp8 e4m3, per-tensor scale    (2 pass equivalents)
    mxfp6   X = fp6 e2m3, e8m0 scale per 32 elements of the contraction index (1.5 pass equivalents)
    mxfp4   X = fp4 e2m1, likewise
    fp16    cross terms dropped               (1 pass)

and prints, per datapath, the error of every loss and of the D / council-D / generator gradients against the exact fp64
iteration, next to the fp32 oracle's own error (the reference's arithmetic).  Accumulation is exact here (the kernels
accumulate in fp32): this isolates the operand-precision question of DESIGN.md section 8 "Open" item 1.

    python tools/probe/emu_datapath_iteration.py [config yaml] [size=64] [council=2] [batch=1] [modes=x3,fp8,mxfp6]
"""
import copy
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import yaml  # noqa: E402

import council_gan_amd as cga  # noqa: E402
import parity_util as P  # noqa: E402
from oracle import council_oracle as O  # noqa: E402

D = torch.float64


def _pow2(t, target):
    m = float(t.abs().max())
    return 2.0 ** np.floor(np.log2(target / m)) if m > 0 else 1.0


def split16(t):
    s = _pow2(t, 8192.0)
    hi = (t * s).to(torch.float16)
    lo = (t * s - hi.to(D)).to(torch.float16)
    return hi.to(D) / s, lo.to(D) / s


def q_fp8(t, kdim):
    s = _pow2(t, 256.0)
    return (t * s).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float32).to(D) / s


def _mx(t, kdim, kind):
    """Blocks of 32 consecutive elements of the contraction index `kdim` (channels; for the weight gradient: positions of
    one image row-major) share a power-of-two scale."""
    if kdim == "pos":
        n, c, h, w = t.shape
        v = t.reshape(n, c, h * w)
        pad = (-v.shape[-1]) % 32
        if pad:
            v = F.pad(v, (0, pad))
        q = _mx_last(v, kind)
        return q[..., :h * w].reshape(n, c, h, w)
    v = t.movedim(kdim, -1)
    return _mx_last(v.contiguous(), kind).movedim(-1, kdim)


def _mx_last(v, kind):
    shp = v.shape
    b = v.reshape(-1, shp[-1] // 32, 32)
    m = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    s = torch.exp2(torch.floor(torch.log2(m)) - 2)
    a = (b / s).abs()
    if kind == "fp6":          # e2m3: max 7.5, 3 mantissa bits, sub-normal step 1/8
        a = a.clamp_max(7.5)
        step = torch.exp2(torch.floor(torch.log2(a.clamp_min(1.0))) - 3)
    else:                      # e2m1: 0 .5 1 1.5 2 3 4 6
        a = a.clamp_max(6.0)
        step = torch.exp2(torch.floor(torch.log2(a.clamp_min(1.0))) - 1)
    return (torch.round(a / step) * step * torch.sign(b) * s).reshape(shp)


def product(op, a, b, ka, kb, mode):
    """op(a, b) bilinear; ka / kb: the contraction index of a / b (a dim number or "pos")."""
    if mode == "exact":
        return op(a, b)
    ah, al = split16(a)
    bh, bl = split16(b)
    y = op(ah, bh)
    if mode == "fp16":
        return y
    if mode == "x3":
        X = lambda t, k: t
    elif mode == "fp8":
        X = q_fp8
    elif mode == "mxfp6":
        X = lambda t, k: _mx(t, k, "fp6")
    elif mode == "mxfp4":
        X = lambda t, k: _mx(t, k, "fp4")
    else:
        raise ValueError(mode)
    return y + op(X(ah, ka), X(bl, kb)) + op(X(al, ka), X(bh, kb))


class EmuConv(torch.autograd.Function):
    MODE = "exact"

    @staticmethod
    def forward(ctx, x, w, stride):
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        mode = EmuConv.MODE if x.shape[1] % 32 == 0 else "exact"
        return product(lambda a, b: F.conv2d(a, b, stride=stride), x, w, 1, 1, mode)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        st = ctx.stride
        dx = dw = None
        if ctx.needs_input_grad[0]:
            mode = EmuConv.MODE if w.shape[0] % 32 == 0 else "exact"
            dx = product(lambda a, b: torch.nn.grad.conv2d_input(x.shape, b, a, stride=st), dy, w, 1, 0, mode)
        if ctx.needs_input_grad[1]:
            mode = EmuConv.MODE if (x.shape[1] % 32 == 0 and w.shape[0] % 8 == 0) else "exact"
            dw = product(lambda a, b: torch.nn.grad.conv2d_weight(a, w.shape, b, stride=st), x, dy, "pos", "pos", mode)
        return dx, dw, None


class FShim:
    """torch.nn.functional with conv2d routed through the emulation (the oracle calls F.conv2d(x, w, bias, stride=...))."""

    def __getattr__(self, name):
        return getattr(F, name)

    @staticmethod
    def conv2d(x, w, bias=None, stride=1, padding=0):
        assert padding == 0
        st = stride if isinstance(stride, int) else stride[0]
        y = EmuConv.apply(x, w, st)
        return y if bias is None else y + bias.view(1, -1, 1, 1)


def losses_of(otr):
    out = {}
    for k, v in vars(otr).items():
        if k.startswith("loss_") and isinstance(v, (list, tuple)) and v and not isinstance(v[0], (list, dict)):
            try:
                out[k] = [float(x) for x in v]
            except (TypeError, ValueError):
                pass
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "male2female_council_folder.yaml"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    council = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    modes = (sys.argv[5] if len(sys.argv) > 5 else "x3,fp8,mxfp6").split(",")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", name)))
    cfg['council']['council_size'] = council
    cfg['iteration'] = 60000
    cfg['batch_size'] = batch
    cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = size
    O.seed_all(1)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')            # host-side construction only
    state = P.host_state(tr)
    x_a, x_b = O.synthetic_batch(batch, size)
    rng = (random.getstate(), torch.get_rng_state())
    print("config %s %dx%d council %d batch %d" % (name, size, size, council, batch), flush=True)

    def run(mode, dtype):
        EmuConv.MODE = mode
        real = O.F
        if mode != "native":
            O.F = FShim()
        try:
            t0 = time.time()
            otr, g, _, _ = P.run_oracle(cfg, state, x_a, x_b, rng, dtype)
            return losses_of(otr), g, time.time() - t0
        finally:
            O.F = real

    l64, g64, dt = run("native", D)
    print("fp64 oracle: %.0f s" % dt, flush=True)
    rows = [("fp32 oracle (the reference's arithmetic)",) + run("native", torch.float32)[:2]]
    chk = run("exact", D)
    worst = max(abs(a - b) / max(abs(b), 1e-30) for k in l64 for a, b in zip(chk[0][k], l64[k]))
    print("emulation plumbing check (exact products through the shim vs native fp64): worst loss difference %.1e" % worst, flush=True)
    for m in modes:
        l, g, dt = run(m, D)
        rows.append(("%s (%.0f s)" % (m, dt), l, g))
        print("  ran %s" % m, flush=True)
    print("%-44s %12s %12s %12s %12s" % ("datapath", "losses", "D grads", "council-D", "G grads"))
    print("%-44s %12s %12s %12s %12s" % ("", "max rel", "l2-rel max", "l2-rel max", "l2-rel min..max"))
    for label, l, g in rows:
        le = max(abs(a - b) / max(abs(b), 1e-30) for k in l64 for a, b in zip(l[k], l64[k]))
        ge = {}
        for kind in ("dis", "disc", "gen"):
            ge[kind] = [P.l2rel(g[key], g64[key]) for key in g64 if key[0] == kind]
        print("%-44s %12.2e %12.2e %12.2e %9.1e..%.1e" % (label, le, max(ge["dis"]), max(ge["disc"]) if ge["disc"] else 0.0,
                                                          min(ge["gen"]), max(ge["gen"])), flush=True)


if __name__ == "__main__":
    main()
