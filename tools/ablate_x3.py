"""Where does conv_fwd_x3_kernel's time go?  Times measurement-only variants of the kernel (wrong results, built with
CG_HIPCC_FLAGS=-DCG_X3_ABLATION) in which one resource is taken out: B operand's LDS traffic, all LDS stores, the
global loads.  Usage: CG_HIPCC_FLAGS=-DCG_X3_ABLATION python council-gan_amd/build_hip.py; python tools/ablate_x3.py"""
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from bench_conv import SHAPES  # noqa: E402
from bench_x3 import split, run_x3  # noqa: E402

CL = torch.channels_last
CFGS = [(1, "128x128/8w"), (27, "8w L1-hit loads"), (22, "8w no-gload"), (25, "8w mfma+Aread"), (28, "8w +no barrier"),
        (31, "8w mfma+barrier"), (29, "8w pure mfma"), (30, "4w pure mfma")]


def main():
    lib = hip.load()
    print("%-30s | " % "shape" + " ".join("%15s" % n for _, n in CFGS))
    for si in (0, 2, 13):
        name, N, H, W, Cin, Cout, K, stride, pad, up = SHAPES[si]
        g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 1)
        x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=CL)
        w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=CL)
        b = torch.randn(Cout, device="cuda")
        y = torch.empty((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=CL)
        xs, ws = split(lib, x), split(lib, w, hip.X3_WSCALE)
        flops = 2.0 * N * g.Ho * g.Wo * Cout * Cin * K * K
        best = {}
        for cfg, _ in CFGS:
            best[cfg] = 1e9
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20 if r else 1):
                    run_x3(lib, g, xs, ws, b, y, cfg)
                e1.record()
                e1.synchronize()
                if r:
                    best[cfg] = min(best[cfg], e0.elapsed_time(e1) / 20)
        print("%-30s | " % name + " ".join("%6.1fus %5.0fTF" % (best[c] * 1000, flops / best[c] / 1e9) for c, _ in CFGS), flush=True)


if __name__ == "__main__":
    main()
