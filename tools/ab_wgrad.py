"""Split-precision weight gradient, one launch shape at a time: time per launch under the library's tile choices (A/B through
the cg_tuning switches), or a plain launch loop for a rocprofv3 --pmc pass.

    python tools/ab_wgrad.py [shape indices]                 table: default | XCD grouping off | bm256 off | wide off  (max difference vs default)
    python tools/ab_wgrad.py --launch <shape index> [reps]   `reps` launches of one shape (PMC_KERNEL=conv_wgrad PMC_TOOL=... tools/prof_bench.sh)

Shapes are the member-batched launches of the bench step (profiles/r05_final_conv_shapes.txt, family f7): G members x N samples."""
import os
import sys
from ctypes import byref, c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402

CL = torch.channels_last
SHAPES = [
    # name, members, N per member, H, W, Cin, Cout, K, stride, pad
    ("res 256->256 3x3 @64 x4", 4, 4, 64, 64, 256, 256, 3, 1, 1),
    ("64->64 3x3 @256 x4", 4, 4, 256, 256, 64, 64, 3, 1, 1),
    ("128->128 3x3 @128 x4", 4, 4, 128, 128, 128, 128, 3, 1, 1),
    ("64->128 4x4s2 @256 x4", 4, 4, 256, 256, 64, 128, 4, 2, 1),
    ("128->256 4x4s2 @128 x4", 4, 4, 128, 128, 128, 256, 4, 2, 1),
    ("DC 64->128 4x4s2 @256 x4 b16", 4, 16, 256, 256, 64, 128, 4, 2, 1),
    ("DC 128->256 4x4s2 @128 x4 b16", 4, 16, 128, 128, 128, 256, 4, 2, 1),
    ("DC 256->512 4x4s2 @64 x4 b16", 4, 16, 64, 64, 256, 512, 4, 2, 1),
    ("res 256->256 3x3 @64 x1", 1, 4, 64, 64, 256, 256, 3, 1, 1),
]


def operands(lib, G, N, H, W, Cin, Cout, K, stride, pad):
    g = ops.fwd_geom(G * N, H, W, Cin, 0, 0, K, K, stride, pad, Cout, 0)      # the geometry's batch is the whole launch: G members x N samples
    x = torch.randn(G * N, Cin, H, W, device="cuda").contiguous(memory_format=CL)
    dz = (torch.randn(G * N, Cout, g.Ho, g.Wo, device="cuda") * 1e-3).contiguous(memory_format=CL)
    with torch.no_grad():
        xs, dzs = ops.split_f16_dynamic(x), ops.split_f16_dynamic(dz)
    elems = Cout * Cin * K * K
    stride_el = (elems + Cout + 31) // 32 * 32
    grp = hip.Group(G, 0, stride_el)
    dw = torch.zeros(G * stride_el, device="cuda")
    if not lib.cg_conv2d_wgrad_x3_ok_g(byref(g), byref(grp)):
        raise SystemExit("shape not taken by the split-precision weight gradient")
    ws = torch.empty(lib.cg_conv2d_wgrad_workspace_g(byref(g), byref(grp)) // 4 + 1, device="cuda")
    flops = 2.0 * G * N * g.Ho * g.Wo * Cout * Cin * K * K

    def run():
        hip.check(lib.cg_conv2d_wgrad_x3_g(byref(g), byref(grp), xs.hi_ptr(), xs.lo, xs.scale_ptr(), dzs.hi_ptr(), dzs.lo,
                                           dzs.scale_ptr(), hip.ptr(dw), c_void_p(dw.data_ptr() + 4 * elems), 0, hip.ptr(ws),
                                           ws.numel() * 4, hip.stream()), "wgrad_x3")
    return run, dw, flops


def timeit(run, reps):
    best = 1e9
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps if r else 1):
            run()
        e1.record()
        e1.synchronize()
        if r:
            best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    lib = hip.load()
    if len(sys.argv) > 1 and sys.argv[1] == "--launch":
        si = int(sys.argv[2])
        reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
        run, dw, flops = operands(lib, *SHAPES[si][1:])
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        print("wgrad", SHAPES[si][0], "done; %.1f GFLOP per launch" % (flops / 1e9))
        return
    shapes = [int(i) for i in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(len(SHAPES)))
    modes = [("default", {}), ("xcd_group=0", {"wgrad_xcd_group": 0}), ("bm256=0", {"wgrad_x3_bm256": 0}), ("wide=0", {"wgrad_x3_wide": 0})]
    print("%-34s | " % "shape" + " ".join("%24s" % m[0] for m in modes))
    for si in shapes:
        run, dw, flops = operands(lib, *SHAPES[si][1:])
        reps = min(50, max(3, int(2e11 / flops / 4)))
        ref, cells = None, []
        for name, fields in modes:
            with hip.tuned(**fields):
                t = timeit(run, reps)
                torch.cuda.synchronize()
                out = dw.clone()
            ref = out if ref is None else ref
            cells.append("%7.1fus %4.0fTF d%.0e" % (t * 1000, flops / t / 1e9, float((out - ref).abs().max() / ref.abs().max())))
        print("%-34s | " % SHAPES[si][0] + " ".join("%24s" % c for c in cells), flush=True)


if __name__ == "__main__":
    main()
