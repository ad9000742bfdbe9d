"""A/B of the opt-in thin-layer kernels against the kernels the default build picks, on the step's own shapes
(member-batched: batch 16 = four members x batch 4).  Usage: python tools/ab_thin.py
  * conv_fwd_thin_kernel (tile configuration 40) / conv_wgrad_thin_kernel vs the generic fp32 kernels
  * split-precision forward tiles 20 / 21 (128x32, 256x32) vs 64x64 (cfg 3) on the 64 -> 3 / 12 channel output layers
  * split-precision weight gradient on the 256x128 / 16-wave tile vs 128x128 / 8 waves"""
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from bench_x3 import split, run_x3  # noqa: E402

CL = torch.channels_last


def timed(fn, reps):
    fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def thin_forward(lib):
    print("== thin-input forward: generic fp32 kernel (auto) vs conv_fwd_thin_kernel (cfg 40)")
    shapes = [("gen 7x7 3->64 @256 b16", 16, 256, 3, 0, 7, 1, 3), ("dis 4x4s2 3->64 @256 b16", 16, 256, 3, 0, 4, 2, 1),
              ("dis 4x4s2 3->64 @256 b32", 32, 256, 3, 0, 4, 2, 1), ("dis 4x4s2 3->64 @128 b32", 32, 128, 3, 0, 4, 2, 1),
              ("cdis 3x3 3+3->64 @256 b16", 16, 256, 3, 3, 3, 1, 1), ("cdis 3x3 3+3->64 @256 b64", 64, 256, 3, 3, 3, 1, 1),
              ("cdis 3x3 3+3->64 @128 b64", 64, 128, 3, 3, 3, 1, 1), ("dgrad 1x1 12->64 @256 b16", 16, 256, 12, 0, 1, 1, 0)]
    for name, N, HW, C1, C2, K, stride, pad in shapes:
        g = ops.fwd_geom(N, HW, HW, C1, C2, 0, K, K, stride, pad, 64, ops.ACT["lrelu"])
        x1 = torch.randn(N, C1, HW, HW, device="cuda").contiguous(memory_format=CL)
        x2 = torch.randn(N, C2, HW, HW, device="cuda").contiguous(memory_format=CL) if C2 else None
        w = (torch.randn(64, C1 + C2, K, K, device="cuda") * 0.1).contiguous(memory_format=CL)
        b = torch.randn(64, device="cuda")
        ys = [torch.empty((N, 64, g.Ho, g.Wo), device="cuda").contiguous(memory_format=CL) for _ in range(2)]
        flops = 2.0 * N * g.Ho * g.Wo * 64 * (C1 + C2) * K * K
        out_mb = ys[0].numel() * 4 / 1e6
        res = []
        for y, cfg in zip(ys, (-1, 40)):
            def fn(y=y, cfg=cfg):
                hip.check(lib.cg_conv2d_fwd_tile(byref(g), hip.ptr(x1), hip.ptr(x2), hip.ptr(w), hip.ptr(b), hip.ptr(y), cfg,
                                                 hip.stream()), "fwd")
            ms = timed(fn, 10)
            res.append("%7.1f us %5.1f TF %4.2f TB/s out" % (ms * 1e3, flops / ms / 1e9, out_mb / ms / 1e3 / 1e3))
        d = float((ys[0] - ys[1]).abs().max() / ys[0].abs().max())
        print("%-30s | auto %s | thin %s | rel diff %.1e" % (name, res[0], res[1], d), flush=True)


def thin_wgrad(lib):
    print("== thin-input weight gradient: generic fp32 kernel vs conv_wgrad_thin_kernel (cg_conv2d_wgrad_thin), incl. the reduce")
    shapes = [("gen 7x7 3->64 @256 b16 (4 members)", 16, 256, 3, 0, 7, 1, 3, 4), ("dis 4x4s2 3->64 @256 b32 (4 members)", 32, 256, 3, 0, 4, 2, 1, 4),
              ("dis 4x4s2 3->64 @128 b32 (4 members)", 32, 128, 3, 0, 4, 2, 1, 4), ("cdis 3x3 3+3->64 @256 b64 (4 members)", 64, 256, 3, 3, 3, 1, 1, 4),
              ("cdis 3x3 3+3->64 @128 b64 (4 members)", 64, 128, 3, 3, 3, 1, 1, 4)]
    for name, N, HW, C1, C2, K, stride, pad, nm in shapes:
        g = ops.fwd_geom(N, HW, HW, C1, C2, 0, K, K, stride, pad, 64, 0)
        x1 = torch.randn(N, C1, HW, HW, device="cuda").contiguous(memory_format=CL)
        x2 = torch.randn(N, C2, HW, HW, device="cuda").contiguous(memory_format=CL) if C2 else None
        dz = (torch.randn(N, 64, g.Ho, g.Wo, device="cuda") * 1e-2).contiguous(memory_format=CL)
        nw = 64 * (C1 + C2) * K * K
        stride_el = nw + 64 + 32
        grp = hip.Group(nm, 0, stride_el)
        flops = 2.0 * N * g.Ho * g.Wo * 64 * (C1 + C2) * K * K
        outs, res = [], []
        for on in (0, 1):
            prev = lib.cg_conv2d_wgrad_thin(on)
            try:
                flat = torch.zeros(nm * stride_el, device="cuda")
                wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace_g(byref(g), byref(grp)))

                def fn(flat=flat, wsb=wsb):
                    hip.check(lib.cg_conv2d_wgrad_g(byref(g), byref(grp), hip.ptr(x1), hip.ptr(x2), hip.ptr(dz), hip.ptr(flat[:nw]),
                                                    hip.ptr(flat[nw:]), 0, hip.ptr(wsb), wsb.numel(), hip.stream()), "wgrad")
                ms = timed(fn, 6)
            finally:
                lib.cg_conv2d_wgrad_thin(prev)
            outs.append(flat)
            res.append("%7.1f us %5.1f TF" % (ms * 1e3, flops / ms / 1e9))
        d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
        print("%-40s | generic %s | thin %s | rel diff %.1e" % (name, res[0], res[1], d), flush=True)


def thin_output(lib):
    print("== thin-output split-precision forward: cfg 3 (64x64) vs 20 (128x32) vs 21 (256x32)")
    shapes = [("64->3 3x3 @256 b16", 16, 256, 64, 3, 3, 1, 1), ("64->12 1x1 @256 b16", 16, 256, 64, 12, 1, 1, 0),
              ("64->3 2x2 (dgrad class) @128 b16", 16, 128, 64, 3, 2, 1, 0)]
    for name, N, HW, Cin, Cout, K, stride, pad in shapes:
        g = ops.fwd_geom(N, HW, HW, Cin, 0, 0, K, K, stride, pad, Cout, 0)
        x = torch.randn(N, Cin, HW, HW, device="cuda").contiguous(memory_format=CL)
        w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=CL)
        b = torch.randn(Cout, device="cuda")
        xs, ws = split(lib, x), split(lib, w, hip.X3_WSCALE)
        flops = 2.0 * N * g.Ho * g.Wo * Cout * Cin * K * K
        in_mb = x.numel() * 4 / 1e6
        ys, res = {}, []
        for cfg in (3, 20, 21):
            ys[cfg] = torch.zeros((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=CL)
            ms = timed(lambda cfg=cfg: run_x3(lib, g, xs, ws, b, ys[cfg], cfg), 10)
            res.append("cfg %2d %7.1f us %5.1f TF %4.2f TB/s in" % (cfg, ms * 1e3, flops / ms / 1e9, in_mb / ms / 1e3 / 1e3))
        d = max(float((ys[c] - ys[3]).abs().max()) for c in (20, 21))
        print("%-34s | %s | max diff %.1e" % (name, " | ".join(res), d), flush=True)


def wgrad_multitap(lib):
    print("== split-precision weight gradient, 32 / 64 input channels: one tap per K-tile (64-wide) | K-tiles spanning taps (128-wide), incl. the reduce")
    shapes = [("D 64->128 4x4s2 @256 b32 (4 members)", 32, 256, 64, 128, 4, 2, 1, 4), ("D 64->128 4x4s2 @128 b32 (4 members)", 32, 128, 64, 128, 4, 2, 1, 4),
              ("DC 64->128 4x4s2 @256 b64 (4 members)", 64, 256, 64, 128, 4, 2, 1, 4), ("G 64->128 4x4s2 @256 b16 (4 members)", 16, 256, 64, 128, 4, 2, 1, 4),
              ("dec 64->64 3x3 @256 b16 (4 members)", 16, 256, 64, 64, 3, 1, 1, 4)]
    for name, N, HW, Cin, Cout, K, stride, pad, nm in shapes:
        g = ops.fwd_geom(N, HW, HW, Cin, 0, 0, K, K, stride, pad, Cout, 0)
        x = torch.randn(N, Cin, HW, HW, device="cuda").contiguous(memory_format=CL)
        dz = (torch.randn(N, Cout, g.Ho, g.Wo, device="cuda") * 1e-3).contiguous(memory_format=CL)
        nw = Cout * Cin * K * K
        stride_el = ((nw + Cout + 31) // 32) * 32
        flat = torch.zeros(nm * stride_el, device="cuda")
        grp = hip.Group(nm, 0, stride_el)
        res, outs = [], []
        with torch.no_grad():
            xs, dzs = ops.split_f16_dynamic(x), ops.split_f16_dynamic(dz)
            for on in (0, 1):
                t = hip.tuning(); prev = t.wgrad_x3_multitap; t.wgrad_x3_multitap = on; lib.cg_tuning_set(byref(t))
                try:
                    wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace_g(byref(g), byref(grp)))

                    def run():
                        hip.check(lib.cg_conv2d_wgrad_x3_g(byref(g), byref(grp), xs.hi_ptr(), xs.lo, xs.scale_ptr(), dzs.hi_ptr(),
                                                           dzs.lo, dzs.scale_ptr(), hip.ptr(flat[:nw]), hip.ptr(flat[nw:]), 0,
                                                           hip.ptr(wsb), wsb.numel(), hip.stream()), "wgrad")
                    ms = timed(run, 5)
                    outs.append(flat.clone())
                finally:
                    t = hip.tuning(); t.wgrad_x3_multitap = prev; lib.cg_tuning_set(byref(t))
                fl = 2.0 * N * g.Ho * g.Wo * Cout * Cin * K * K
                res.append("%7.1f us %6.1f TF" % (ms * 1000, fl / ms / 1e9))
        d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
        print("%-40s | %s | %s | max rel diff %.1e" % (name, res[0], res[1], d), flush=True)


def wgrad_256(lib):
    print("== split-precision weight gradient: 128x128 / 8 waves | 256x128 / 16 waves | 256x256 LDS-DMA / 8 waves (experimental), incl. the reduce")
    shapes = [("res 256->256 3x3 @64 b16 (4 members)", 16, 64, 256, 256, 3, 1, 1, 4),
              ("res 256->256 3x3 @64 b4 (1 member)", 4, 64, 256, 256, 3, 1, 1, 1),
              ("128->256 4x4s2 @128 b16 (4 members)", 16, 128, 128, 256, 4, 2, 1, 4),
              ("256->512 4x4s2 @64 b64 (4 members)", 64, 64, 256, 512, 4, 2, 1, 4),
              ("512->512 1x1 @32 b64 (4 members)", 64, 32, 512, 512, 1, 1, 0, 4),
              # one member per rank (tools/one_member_rank.py): a discriminator sees 20 images, the generator 4
              ("res 256->256 3x3 @64 b8 (1 member)", 8, 64, 256, 256, 3, 1, 1, 1),
              ("128->256 4x4s2 @128 b20 (1 member)", 20, 128, 128, 256, 4, 2, 1, 1),
              ("256->512 4x4s2 @64 b20 (1 member)", 20, 64, 256, 512, 4, 2, 1, 1),
              ("256->512 4x4s2 @32 b20 (1 member)", 20, 32, 256, 512, 4, 2, 1, 1),
              ("256->128 3x3 @128 b4 (1 member)", 4, 128, 256, 128, 3, 1, 1, 1)]
    wide_too = "wide" in sys.argv[1:]
    for name, N, HW, Cin, Cout, K, stride, pad, nm in shapes:
        g = ops.fwd_geom(N, HW, HW, Cin, 0, 0, K, K, stride, pad, Cout, 0)
        x = torch.randn(N, Cin, HW, HW, device="cuda").contiguous(memory_format=CL)
        dz = (torch.randn(N, Cout, g.Ho, g.Wo, device="cuda") * 1e-3).contiguous(memory_format=CL)
        nw = Cout * Cin * K * K
        stride_el = nw + Cout + 32
        grp = hip.Group(nm, 0, stride_el)
        flops = 2.0 * N * g.Ho * g.Wo * Cout * Cin * K * K
        outs, res = [], []
        with torch.no_grad():
            xs, dzs = ops.split_f16_dynamic(x), ops.split_f16_dynamic(dz)
            for label, bm256, wide in (("128x128", 0, 0), ("256x128", 1, 0)) + ((("256x256w", 0, 1),) if wide_too and Cin % 256 == 0 else ()):
                prev, prevw = lib.cg_conv2d_wgrad_x3_bm256(bm256), lib.cg_conv2d_wgrad_x3_wide(wide)
                try:
                    flat = torch.zeros(nm * stride_el, device="cuda")
                    wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace_g(byref(g), byref(grp)))

                    def fn(flat=flat, wsb=wsb):
                        hip.check(lib.cg_conv2d_wgrad_x3_g(byref(g), byref(grp), xs.hi_ptr(), xs.lo, xs.scale_ptr(), dzs.hi_ptr(),
                                                           dzs.lo, dzs.scale_ptr(), hip.ptr(flat[:nw]), hip.ptr(flat[nw:]), 0,
                                                           hip.ptr(wsb), wsb.numel(), hip.stream()), "wgrad")
                    ms = timed(fn, 10)
                finally:
                    lib.cg_conv2d_wgrad_x3_bm256(prev)
                    lib.cg_conv2d_wgrad_x3_wide(prevw)
                outs.append(flat)
                res.append("%s %7.1f us %5.1f TF" % (label, ms * 1e3, flops / ms / 1e9))
        d = max(float((o - outs[0]).abs().max() / outs[0].abs().max()) for o in outs[1:])
        print("%-40s | %s | max rel diff %.1e" % (name, " | ".join(res), d), flush=True)


if __name__ == "__main__":
    lib = hip.load()
    if sys.argv[1:2] == ["multitap"]:
        wgrad_multitap(lib)
        sys.exit(0)
    which = sys.argv[1:] or ["thin", "thinw", "out", "wgrad"]
    if "thin" in which:
        thin_forward(lib)
    if "thinw" in which:
        thin_wgrad(lib)
    if "out" in which:
        thin_output(lib)
    if "wgrad" in which:
        wgrad_256(lib)
