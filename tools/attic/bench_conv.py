"""A/B benchmark of the forward implicit-GEMM tile configurations on the real layer shapes of the
256x256 / batch-4 step.  Interleaved rounds in ONE process (guide rule 24), random
data, HIP events on the launch stream.  Usage: python tools/bench_conv.py [rounds]"""
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402
from council_gan_amd import hip, ops  # noqa: E402

CFG_NAMES = {0: "128x128/4w/1s", 1: "128x64/4w/1s", 2: "128x32/4w/1s", 3: "64x64/4w/1s", 4: "128x128/4w/2s",
             5: "128x64/4w/2s", 6: "128x128/8w/1s", 7: "128x128/8w/2s", 8: "64x128/4w/1s", 9: "64x128/4w/2s",
             10: "64x64/4w/2s", 11: "256x128/8w/1s", 12: "256x128/8w/2s", 13: "128x32/4w/2s", 14: "128x64/8w/1s",
             15: "256x64/8w/1s", 16: "64x128/8w/1s", 17: "128x128/16w", 18: "256x128/16w",
             20: "P128x128/8w", 21: "P128x128/4w", 22: "P128x64/4w", 23: "P64x64/4w", 24: "P256x128/8w",
             25: "P128x64/8w", 26: "P64x128/4w", 27: "P128x128/8w/pf2", 28: "abl:fixedslice", 29: "abl:nostagger",
             30: "P128x128/4w/pf2", 32: "P256x64/8w"}
SKIP = {1, 2, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18}      # double-buffered / 256x128-8w variants: measured, never better (profiles/r01_conv_tiles.txt)

SHAPES = [
    # name, N, H, W, Cin, Cout, K, stride, pad, up
    ("res 256->256 3x3 @64", 4, 64, 64, 256, 256, 3, 1, 1, 0),
    ("up 256->128 3x3 @128", 4, 64, 64, 256, 128, 3, 1, 1, 1),
    ("128->128 3x3 @128", 4, 128, 128, 128, 128, 3, 1, 1, 0),
    ("up 128->64 3x3 @256", 4, 128, 128, 128, 64, 3, 1, 1, 1),
    ("64->64 3x3 @256", 4, 256, 256, 64, 64, 3, 1, 1, 0),
    ("64->128 4x4s2 @256", 4, 256, 256, 64, 128, 4, 2, 1, 0),
    ("128->256 4x4s2 @128", 4, 128, 128, 128, 256, 4, 2, 1, 0),
    ("D 64->128 4x4s2 @128 b8", 8, 128, 128, 64, 128, 4, 2, 1, 0),
    ("D 128->256 4x4s2 @64 b8", 8, 64, 64, 128, 256, 4, 2, 1, 0),
    ("D 256->512 4x4s2 @32 b8", 8, 32, 32, 256, 512, 4, 2, 1, 0),
    ("DC 64->128 4x4s2 @256 b16", 16, 256, 256, 64, 128, 4, 2, 1, 0),
    ("DC 256->512 4x4s2 @64 b16", 16, 64, 64, 256, 512, 4, 2, 1, 0),
    ("1x1 64->64 @256", 4, 256, 256, 64, 64, 1, 1, 0, 0),
    ("dgrad-class 256x4taps->128 @64", 4, 64, 64, 256, 128, 2, 1, 0, 0),
    ("DCdgrad-class 128x4taps->64 @128 b16", 16, 128, 128, 128, 64, 2, 1, 0, 0),
    ("D 512->512 1x1 @32 b16", 16, 32, 32, 512, 512, 1, 1, 0, 0),
    ("Ddgrad-class 512x4taps->256 @16 b4", 4, 16, 16, 512, 256, 2, 1, 0, 0),
]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    lib = hip.load()
    cfgs = [c for c in sorted(CFG_NAMES) if c not in SKIP]
    if os.environ.get("CFGS"):
        cfgs = [int(c) for c in os.environ["CFGS"].split(",")]
    shapes = SHAPES
    if os.environ.get("SHAPES"):
        shapes = [SHAPES[int(i)] for i in os.environ["SHAPES"].split(",")]
    print("%-34s %8s | " % ("shape", "GFLOP") + " ".join("%13s" % CFG_NAMES[c] for c in cfgs))
    for name, N, H, W, Cin, Cout, K, stride, pad, up in shapes:
        g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 1)
        x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device="cuda")
        y = torch.empty((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=torch.channels_last)
        flops = 2.0 * N * g.Ho * g.Wo * Cout * Cin * K * K
        ref = None
        best = {c: 1e9 for c in cfgs}
        ok = {}
        for c in cfgs:           # correctness of every configuration against configuration 0
            rc = lib.cg_conv2d_fwd_tile(byref(g), hip.ptr(x), None, hip.ptr(w), hip.ptr(b), hip.ptr(y), c, hip.stream())
            if rc != 0:
                ok[c] = None
                continue
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            ok[c] = float((y - ref).abs().max())
        reps = max(3, int(2e11 / flops / 4))
        reps = min(reps, 50)
        for r in range(rounds):
            for c in cfgs:
                if ok[c] is None:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    lib.cg_conv2d_fwd_tile(byref(g), hip.ptr(x), None, hip.ptr(w), hip.ptr(b), hip.ptr(y), c, hip.stream())
                e1.record()
                e1.synchronize()
                best[c] = min(best[c], e0.elapsed_time(e1) / reps)
        cells = []
        for c in cfgs:
            if ok[c] is None:
                cells.append("%13s" % "n/a")
            else:
                cells.append("%6.1fTF%s" % (flops / best[c] / 1e9, " " if ok[c] < 1e-3 else "!") + "%5.0fus" % (best[c] * 1000))
        print("%-34s %8.2f | " % (name, flops / 1e9) + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
