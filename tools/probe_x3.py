"""Per-block timeline + in-loop shader clock of conv_fwd_x3_kernel probe variants (library built with
CG_HIPCC_FLAGS=-DCG_X3_ABLATION).  Usage: python tools/probe_x3.py <shape index> <cfg,cfg,...>"""
import os
import sys
from ctypes import c_int64

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from bench_conv import SHAPES  # noqa: E402
from bench_x3 import split, run_x3  # noqa: E402

NAMES = {50: "8w full", 51: "8w pure mfma", 52: "8w mfma+Aread", 53: "8w K64 full", 54: "8w no-gload", 55: "4w full"}


def main():
    si = int(sys.argv[1])
    cfgs = [int(c) for c in sys.argv[2].split(",")]
    lib = hip.load()
    name, N, H, W, Cin, Cout, K, stride, pad, up = SHAPES[si]
    g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 1)
    x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device="cuda")
    y = torch.empty((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=torch.channels_last)
    xs, ws = split(lib, x), split(lib, w, hip.X3_WSCALE)
    M = N * g.Ho * g.Wo
    tiles = ((M + 127) // 128) * ((Cout + 127) // 128)
    n = min(tiles, 4096)
    print("%s: %d tiles" % (name, tiles))
    for cfg in cfgs:
        for _ in range(20):
            run_x3(lib, g, xs, ws, b, y, cfg)
        torch.cuda.synchronize()
        buf = (c_int64 * (n * 16))()
        hip.check(lib.cg_debug_fetch(buf, n * 16), "fetch")
        a = np.array(buf[:], dtype=np.int64).reshape(n, 16)
        clk = a[:, 0:8:2].astype(np.float64)
        wall = a[:, 1:8:2].astype(np.float64) * 10.0      # ns
        t0 = wall[:, 0].min()
        out = "  cfg %d %-14s span %.1f us |" % (cfg, NAMES.get(cfg, ""), (wall[:, 3].max() - t0) / 1e3)
        for nm, i, j in (("pro", 0, 1), ("loop", 1, 2), ("epi", 2, 3)):
            d = (wall[:, j] - wall[:, i]) / 1e3
            ghz = (clk[:, j] - clk[:, i]) / np.maximum(wall[:, j] - wall[:, i], 1)
            out += " %s %.1f us @ %.2f GHz |" % (nm, np.median(d), np.median(ghz))
        print(out, flush=True)


if __name__ == "__main__":
    main()
