"""Per-block timeline of the pipelined conv kernel (tile cfg 31): where does a launch spend its time?
Usage: python tools/probe_conv.py <shape index>"""
import os
import sys
from ctypes import byref, c_int64

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from bench_conv import SHAPES  # noqa: E402


def main():
    si = int(sys.argv[1])
    lib = hip.load()
    name, N, H, W, Cin, Cout, K, stride, pad, up = SHAPES[si]
    g = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 1)
    x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device="cuda")
    y = torch.empty((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=torch.channels_last)
    M = N * g.Ho * g.Wo
    tiles = ((M + 127) // 128) * ((Cout + 127) // 128)
    for _ in range(5):
        hip.check(lib.cg_conv2d_fwd_tile(byref(g), hip.ptr(x), None, hip.ptr(w), hip.ptr(b), hip.ptr(y), 31, hip.stream()), "conv")
    torch.cuda.synchronize()
    n = min(tiles, 4096)
    buf = (c_int64 * (n * 16))()
    hip.check(lib.cg_debug_fetch(buf, n * 16), "fetch")
    a = np.array(buf[:], dtype=np.int64).reshape(n, 16)
    clk = a[:, 0:8:2].astype(np.float64)
    wall = a[:, 1:8:2].astype(np.float64) * 10.0      # ns
    t0 = wall[:, 0].min()
    print("%s: %d tiles; kernel span %.1f us" % (name, n, (wall[:, 3].max() - t0) / 1e3))
    for nm, i, j in (("prologue", 0, 1), ("loop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
        d = (wall[:, j] - wall[:, i]) / 1e3
        c = clk[:, j] - clk[:, i]
        ghz = c / np.maximum(wall[:, j] - wall[:, i], 1)
        print("  %-9s us: min %.1f  med %.1f  max %.1f   | clock GHz med %.3f (min %.3f max %.3f)"
              % (nm, d.min(), np.median(d), d.max(), np.median(ghz), ghz.min(), ghz.max()))
    st = (wall[:, 0] - t0) / 1e3
    en = (wall[:, 3] - t0) / 1e3
    print("  block start us: min %.1f med %.1f max %.1f ; block end us: min %.1f med %.1f max %.1f"
          % (st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max()))
    hw = a[:, 8] & 0xffffffff
    xcc = a[:, 8] >> 32
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print("  distinct (xcc,se,sh,cu) = %d ; blocks per CU histogram: %s" % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
    # loop time by blocks-per-CU class
    per = dict(zip(uniq, cnt))
    cls = np.array([per[k] for k in key])
    for c in sorted(set(cls)):
        d = (wall[cls == c, 2] - wall[cls == c, 1]) / 1e3
        print("  blocks sharing a CU with %d-1 others: n=%d loop med %.1f us" % (c, (cls == c).sum(), np.median(d)))


if __name__ == "__main__":
    main()
