"""A/B of conv_fwd_x3 tile configurations: time per launch and max difference against the first configuration.
Usage: python tools/ab_x3.py 1,40,0,41 [shape indices, default 0,1,2,3,4,13,14]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from bench_conv import SHAPES  # noqa: E402
from bench_x3 import split, run_x3  # noqa: E402

CL = torch.channels_last


def main():
    cfgs = [int(c) for c in sys.argv[1].split(",")]
    shapes = [int(i) for i in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 13, 14]
    table = list(SHAPES)
    for spec in filter(None, os.environ.get("EXTRA_SHAPES", "").split(";")):     # name,N,H,W,Cin,Cout,K,stride,pad,up
        f = spec.split(",")
        table.append((f[0],) + tuple(int(v) for v in f[1:]))
        shapes.append(len(table) - 1)
    lib = hip.load()
    print("%-38s | " % "shape" + " ".join("%22s" % ("cfg %d" % c) for c in cfgs))
    for si in shapes:
        name, N, H, W, Cin, Cout, K, stride, pad, up = table[si]
        g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 1)
        x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=CL)
        w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=CL)
        b = torch.randn(Cout, device="cuda")
        ys = {c: torch.zeros((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=CL) for c in cfgs}
        xs, ws = split(lib, x), split(lib, w, hip.X3_WSCALE)
        flops = 2.0 * N * g.Ho * g.Wo * Cout * Cin * K * K
        reps = min(50, max(3, int(2e11 / flops / 4)))
        best = {c: 1e9 for c in cfgs}
        for r in range(4):
            for c in cfgs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps if r else 1):
                    run_x3(lib, g, xs, ws, b, ys[c], c)
                e1.record()
                e1.synchronize()
                if r:
                    best[c] = min(best[c], e0.elapsed_time(e1) / reps)
        ref = ys[cfgs[0]]
        print("%-38s | " % name + " ".join("%6.1fus %4.0fTF d%.0e" % (best[c] * 1000, flops / best[c] / 1e9,
              float((ys[c] - ref).abs().max())) for c in cfgs), flush=True)


if __name__ == "__main__":
    main()
