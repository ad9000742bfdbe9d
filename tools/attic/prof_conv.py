"""Run ONE forward-conv shape / tile configuration a few times (for rocprofv3 --pmc / --kernel-trace).
Usage: python tools/prof_conv.py <shape index into bench_conv.SHAPES> <tile cfg> [reps]"""
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
    si, cfg = int(sys.argv[1]), int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    lib = hip.load()
    name, N, H, W, Cin, Cout, K, stride, pad, up = SHAPES[si]
    g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 1)
    x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device="cuda")
    y = torch.empty((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=torch.channels_last)
    for _ in range(reps):
        hip.check(lib.cg_conv2d_fwd_tile(byref(g), hip.ptr(x), None, hip.ptr(w), hip.ptr(b), hip.ptr(y), cfg, hip.stream()),
                  "conv")
    torch.cuda.synchronize()
    print(name, cfg, "done")


if __name__ == "__main__":
    main()
