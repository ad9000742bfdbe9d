"""Numerics probe (CPU, no GPU): would the two CROSS terms of the split-precision product survive fp8?

Today a.b ~ ah.bh + ah.bl + al.bh with all four planes fp16 (3 fp16 MFMA passes, 22 significand bits).  The cross terms are
2^-11 of the main term, so their operands need only a few bits: evaluate them as fp8 (e4m3) MFMAs -- twice the fp16 rate on
gfx950 -- and the product costs 1 + 2/2 = 2 fp16-pass equivalents instead of 3 (block-scaled fp6 / fp4 run at four times
the fp16 rate: 1.5 pass equivalents).  This script emulates that on a res-block
sized convolution in float64 arithmetic with quantised operands and prints the error against the exact fp64 result next to
plain fp32, fp16 x 3 and single-pass fp16.

  python tools/probe/emu_fp8_cross_terms.py
"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
D = torch.float64


def pow2_scale(t, target):
    """power-of-two s with max|t|*s in [target/2, target)"""
    m = float(t.abs().max())
    import math
    return 2.0 ** math.floor(math.log2(target / m)) if m > 0 else 1.0


def split16(t):
    s = pow2_scale(t, 8192.0)                    # the shipped convention: peak in [4096, 8192)
    hi = (t * s).to(torch.float16)
    lo = (t * s - hi.to(D)).to(torch.float16)
    return hi.to(D) / s, lo.to(D) / s


def q8(t, target=256.0):
    """e4m3 with a per-tensor power-of-two scale (peak in [128, 256); e4m3 max 448, min normal 2^-6, subnormal 2^-9)."""
    s = pow2_scale(t, target)
    return (t * s).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float32).to(D) / s


def mx(t, cdim, kind):
    """OCP microscaling along the contraction (channel) dimension: blocks of 32 channels share one power-of-two scale
    (e8m0), elements are fp6 e2m3 (max 7.5, step 1/8 below 2) or fp4 e2m1 (max 6: 0, .5, 1, 1.5, 2, 3, 4, 6) or fp8 e4m3."""
    t = t.movedim(cdim, -1)
    shp = t.shape
    b = t.reshape(-1, shp[-1] // 32, 32)
    m = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    emax = {"fp6": 2, "fp4": 2, "fp8": 8}[kind]
    s = torch.exp2(torch.floor(torch.log2(m)) - emax)
    v = b / s
    if kind == "fp8":
        q = v.clamp(-448, 448).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float32).to(D)
    else:
        a = v.abs()
        if kind == "fp6":
            a = a.clamp_max(7.5)
            e = torch.floor(torch.log2(a.clamp_min(1.0)))          # 0, 1, 2 (sub-normals share exponent 0)
            step = torch.exp2(e - 3)
        else:
            a = a.clamp_max(6.0)
            e = torch.floor(torch.log2(a.clamp_min(1.0)))
            step = torch.exp2(e - 1)
        q = torch.round(a / step) * step * torch.sign(v)
    return (q * s).reshape(shp).movedim(-1, cdim)


def conv(a, b):
    return F.conv2d(a, b, padding=1)


def err(y, ref):
    d = (y - ref).abs()
    return float(d.max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


def main():
    for name, x, w in (
        ("normalised activations, kaiming weights",
         torch.randn(2, 256, 32, 32, dtype=D).clamp_min(0), torch.randn(256, 256, 3, 3, dtype=D) * (2.0 / 2304) ** 0.5),
        ("heavy-tailed activations (|N|^3), N(0, 0.02) weights",
         torch.randn(2, 256, 32, 32, dtype=D).pow(3), torch.randn(256, 256, 3, 3, dtype=D) * 0.02),
        ("gradient-like operand (1e-6 scale, 1e3 dynamic range)",
         torch.randn(2, 256, 32, 32, dtype=D) * torch.logspace(-6, -3, 256, dtype=D).view(1, 256, 1, 1),
         torch.randn(256, 256, 3, 3, dtype=D) * 0.03)):
        ref = conv(x, w)
        xh, xl = split16(x)
        wh, wl = split16(w)
        rows = []
        rows.append(("fp32 operands, fp32 accumulate (torch CPU)", conv(x.float(), w.float()).to(D)))
        rows.append(("fp16 single pass (ah.bh)", conv(xh, wh)))
        rows.append(("fp16 x 3 (shipped)", conv(xh, wh) + conv(xh, wl) + conv(xl, wh)))
        rows.append(("fp16 main + fp8 cross terms", conv(xh, wh) + conv(q8(xh), q8(wl)) + conv(q8(xl), q8(wh))))
        rows.append(("fp16 x 2 (one operand unsplit: ah.bh + al.bh)", conv(xh, wh) + conv(xl, wh)))
        for kind in ("fp8", "fp6", "fp4"):
            rows.append(("fp16 main + MX-%s cross terms (32-channel blocks)" % kind,
                         conv(xh, wh) + conv(mx(xh, 1, kind), mx(wl, 1, kind)) + conv(mx(xl, 1, kind), mx(wh, 1, kind))))
        print("== %s" % name)
        for label, y in rows:
            e = err(y, ref)
            print("   %-48s max-rel %.2e   rms-rel %.2e" % (label, e[0], e[1]))


if __name__ == "__main__":
    main()
