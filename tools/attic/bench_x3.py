"""Split-precision (fp16 x 3) forward conv vs the fp32-MFMA kernel: error against an fp64 reference (small
shape, CPU) and throughput on the decoder shapes.  Usage: python tools/bench_x3.py [rounds]"""
import ctypes
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from bench_conv import SHAPES  # noqa: E402

CL = torch.channels_last


def split(lib, t, scale=1.0):
    """fp32 tensor (physical layout kept) -> [2, numel] fp16 planes"""
    out = torch.empty(2 * t.numel(), dtype=torch.float16, device=t.device)
    hip.check(lib.cg_split_f16(hip.ptr(t), hip.ptr(out), t.numel(), ops.x3_lo(t.numel()), scale, hip.stream()), "split")
    return out


def run_x3(lib, g, xs, ws, b, y, cfg):
    hip.check(lib.cg_conv2d_fwd_x3(byref(g), hip.ptr(xs), ops.x3_lo(xs.numel() // 2), hip.ptr(ws), ops.x3_lo(ws.numel() // 2), hip.X3_WSCALE, None,
                                   hip.ptr(b), hip.ptr(y), None, 0, None, 0, None, cfg, None, None, hip.stream()), "x3")


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    lib = hip.load()
    # ---- accuracy on a shape small enough for an fp64 CPU reference
    torch.manual_seed(0)
    N, H, W, Cin, Cout, K = 2, 32, 32, 256, 256, 3
    x = torch.randn(N, Cin, H, W)
    w = torch.randn(Cout, Cin, K, K) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xd, wd, bd = x.cuda().contiguous(memory_format=CL), w.cuda().contiguous(memory_format=CL), b.cuda()
    g = ops.fwd_geom(N, H, W, Cin, 0, 0, K, K, 1, 1, Cout, 0)
    y32 = torch.empty((N, Cout, H, W), device="cuda").contiguous(memory_format=CL)
    hip.check(lib.cg_conv2d_fwd_tile(byref(g), hip.ptr(xd), None, hip.ptr(wd), hip.ptr(bd), hip.ptr(y32), 20, hip.stream()), "fp32")
    xs, ws = split(lib, xd), split(lib, wd, hip.X3_WSCALE)
    scale = float(ref.abs().max())
    e32 = float((y32.cpu().double() - ref).abs().max()) / scale
    r32 = float(((y32.cpu().double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    print("fp32 MFMA            : max-abs/max %.2e   rms-rel %.2e" % (e32, r32))
    for cfg in (0, 1, 2, 3, 5, 6, 12, 13):
        y = torch.empty_like(y32)
        run_x3(lib, g, xs, ws, bd, y, cfg)
        e = float((y.cpu().double() - ref).abs().max()) / scale
        r = float(((y.cpu().double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
        print("fp16x3 cfg %d         : max-abs/max %.2e   rms-rel %.2e" % (cfg, e, r))
    # plain fp16 (hi*hi only) for scale: emulate by zeroing the lo planes
    xs0, ws0 = xs.clone(), ws.clone()
    if ops.x3_interleaved():
        xs0.view(-1, 64)[:, 32:] = 0
        ws0.view(-1, 64)[:, 32:] = 0
    else:
        xs0[xd.numel():] = 0
        ws0[wd.numel():] = 0
    y = torch.empty_like(y32)
    run_x3(lib, g, xs0, ws0, bd, y, 0)
    print("fp16 (hi only)       : max-abs/max %.2e" % (float((y.cpu().double() - ref).abs().max()) / scale))

    # ---- throughput
    print("%-34s %8s | %16s %16s %16s %16s %16s" % ("shape", "GFLOP", "fp32 pipe", "x3 128x128/8w", "x3 128x64/8w", "x3 128x128/16w", "x3 256x128/16w"))
    for si in (0, 1, 2, 3, 4, 12):
        name, N, H, W, Cin, Cout, K, stride, pad, up = SHAPES[si]
        g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 1)
        x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=CL)
        w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=CL)
        b = torch.randn(Cout, device="cuda")
        y = torch.empty((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=CL)
        y2 = torch.empty_like(y)
        xs, ws = split(lib, x), split(lib, w, hip.X3_WSCALE)
        flops = 2.0 * N * g.Ho * g.Wo * Cout * Cin * K * K
        reps = min(50, max(3, int(2e11 / flops / 4)))
        runs = [("fp32", lambda: hip.check(lib.cg_conv2d_fwd(byref(g), hip.ptr(x), None, hip.ptr(w), hip.ptr(b), hip.ptr(y), hip.stream()), "f"))]
        for cfg in (1, 2, 12, 13):
            runs.append(("x3_%d" % cfg, (lambda c: (lambda: run_x3(lib, g, xs, ws, b, y2, c)))(cfg)))
        best = {k: 1e9 for k, _ in runs}
        for r in range(rounds + 1):
            for k, fn in runs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps if r else 1):
                    fn()
                e1.record()
                e1.synchronize()
                if r:
                    best[k] = min(best[k], e0.elapsed_time(e1) / reps)
        d = float((y - y2).abs().max() / y.abs().max())
        print("%-34s %8.2f | " % (name, flops / 1e9) + " ".join("%7.1fTF %5.0fus" % (flops / best[k] / 1e9, best[k] * 1000) for k, _ in runs)
              + "   |fp32-x3| %.1e" % d, flush=True)


if __name__ == "__main__":
    main()
