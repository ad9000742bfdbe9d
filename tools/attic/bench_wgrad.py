"""A/B of the weight-gradient kernels (legacy vs pipelined) on the layer shapes of the 256x256 / batch-4 step.
Usage: python tools/bench_wgrad.py [rounds]"""
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from bench_conv import SHAPES  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    lib = hip.load()
    print("%-34s %8s | %18s %18s   max|diff|" % ("shape", "GFLOP", "legacy", "pipelined"))
    for name, N, H, W, Cin, Cout, K, stride, pad, up in SHAPES:
        g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 0)
        x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        dz = torch.randn(N, Cout, g.Ho, g.Wo, device="cuda").contiguous(memory_format=torch.channels_last)
        dw = [torch.empty(Cout, Cin, K, K, device="cuda").contiguous(memory_format=torch.channels_last) for _ in range(2)]
        db = [torch.empty(Cout, device="cuda") for _ in range(2)]
        ws = hip.workspace(lib.cg_conv2d_wgrad_workspace(byref(g)))
        flops = 2.0 * N * g.Ho * g.Wo * Cout * Cin * K * K
        best = [1e9, 1e9]
        reps = min(50, max(3, int(2e11 / flops / 4)))
        for r in range(rounds + 1):
            for leg in (1, 0):
                lib.cg_conv2d_wgrad_legacy(leg)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps if r else 1):
                    hip.check(lib.cg_conv2d_wgrad(byref(g), hip.ptr(x), None, hip.ptr(dz), hip.ptr(dw[leg]), hip.ptr(db[leg]), 0,
                                                  hip.ptr(ws), ws.numel(), hip.stream()), "wgrad")
                e1.record()
                e1.synchronize()
                if r:
                    best[leg] = min(best[leg], e0.elapsed_time(e1) / reps)
        lib.cg_conv2d_wgrad_legacy(0)
        diff = max(float((dw[0] - dw[1]).abs().max() / dw[1].abs().max()), float((db[0] - db[1]).abs().max() / db[1].abs().max()))
        print("%-34s %8.2f | %8.1fTF %6.0fus %8.1fTF %6.0fus   %.2e" % (name, flops / 1e9, flops / best[1] / 1e9, best[1] * 1000,
                                                                    flops / best[0] / 1e9, best[0] * 1000, diff), flush=True)


if __name__ == "__main__":
    main()
