"""Split-precision forward convolution, one launch shape at a time: A/B of tile configurations (time per launch, max difference
against the first configuration), or a plain launch loop for a rocprofv3 --pmc pass.

    python tools/ab_x3.py 16,38,40 [shape indices, default 0,1,2,3,4,13,14]     A/B table; EXTRA_SHAPES="name,N,H,W,Cin,Cout,K,stride,pad,up;..."
    python tools/ab_x3.py --launch 16 [batch=16] [reps=8]                        res-block shape, `reps` launches (tools/prof_bench.sh PMC=16)
    PMC_SHAPE="name,N,H,W,Cin,Cout,K,stride,pad,up" python tools/ab_x3.py --launch <cfg>   ... of any other shape (batch argument ignored)
"""
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402

CL = torch.channels_last
SHAPES = [
    # name, N, H, W, Cin, Cout, K, stride, pad, up      (single-member batch 4 unless the name says otherwise)
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


def split(lib, t, scale=1.0):
    """fp32 tensor (physical layout kept) -> {hi, lo} fp16 planes"""
    out = torch.empty(2 * t.numel(), dtype=torch.float16, device=t.device)
    hip.check(lib.cg_split_f16(hip.ptr(t), hip.ptr(out), t.numel(), ops.x3_lo(t.numel()), scale, hip.stream()), "split")
    return out


def run_x3(lib, g, xs, ws, b, y, cfg):
    hip.check(lib.cg_conv2d_fwd_x3(byref(g), hip.ptr(xs), ops.x3_lo(xs.numel() // 2), hip.ptr(ws), ops.x3_lo(ws.numel() // 2),
                                   hip.X3_WSCALE, None, hip.ptr(b), hip.ptr(y), None, 0, None, 0, None, cfg, None, None, hip.stream()),
              "x3")


def operands(lib, N, H, W, Cin, Cout, K, stride, pad, up):
    g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, int(os.environ.get('AB_ACT', '1')))      # AB_ACT=0: no fused activation
    x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=CL)
    w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=CL)
    return g, split(lib, x), split(lib, w, hip.X3_WSCALE), torch.randn(Cout, device="cuda")


def launch_loop(argv):
    cfg = int(argv[0]) if argv else 16
    N = int(argv[1]) if len(argv) > 1 else 16
    reps = int(argv[2]) if len(argv) > 2 else 8
    lib = hip.load()
    shape = (N, 64, 64, 256, 256, 3, 1, 1, 0)
    if os.environ.get("PMC_SHAPE"):
        shape = tuple(int(v) for v in os.environ["PMC_SHAPE"].split(",")[1:])
    g, xs, ws, b = operands(lib, *shape)
    N, H, W, Cin, Cout, K = shape[:6]
    y = torch.empty((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=CL)
    for _ in range(reps):
        run_x3(lib, g, xs, ws, b, y, cfg)
    torch.cuda.synchronize()
    print("cfg", cfg, "shape", shape, "done; %.1f GFLOP; algorithmic bytes per launch: in %.1f MB (hi+lo fp16) + weights %.1f MB + out %.1f MB"
          % (2.0 * y.numel() * Cin * K * K / 1e9, N * Cin * H * W * 4 / 1e6, Cout * Cin * K * K * 4 / 1e6, y.numel() * 4 / 1e6))


def main():
    if sys.argv[1] == "--launch":
        return launch_loop(sys.argv[2:])
    cfgs = [int(c) for c in sys.argv[1].split(",")]
    shapes = [int(i) for i in sys.argv[2].split(",") if i != "-"] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 13, 14]      # "-": EXTRA_SHAPES only
    table = list(SHAPES)
    for spec in filter(None, os.environ.get("EXTRA_SHAPES", "").split(";")):     # name,N,H,W,Cin,Cout,K,stride,pad,up
        f = spec.split(",")
        table.append((f[0],) + tuple(int(v) for v in f[1:]))
        shapes.append(len(table) - 1)
    lib = hip.load()
    print("%-38s | " % "shape" + " ".join("%22s" % ("cfg %d" % c) for c in cfgs))
    for si in shapes:
        name, N, H, W, Cin, Cout, K, stride, pad, up = table[si]
        g, xs, ws, b = operands(lib, N, H, W, Cin, Cout, K, stride, pad, up)
        ys = {c: torch.zeros((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=CL) for c in cfgs}
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
