"""Where the wide split-precision tile (conv_fwd_x3w_kernel<256,256>) spends its K loop: tile configuration 29 is cfg 16 with
s_memtime stamps around the DMA-landed wait, the slice barrier and the DMA issue of every wave.
Usage: python tools/probe_x3w_stalls.py [batch=16] [probe cfg=29] [waves per block=8]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from ab_x3 import split, run_x3  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
PCFG = int(sys.argv[2]) if len(sys.argv) > 2 else 29
NWB = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib = hip.load()
g = ops.fwd_geom(N, 64, 64, 256, 0, 0, 3, 3, 1, 1, 256, 1)
x = torch.randn(N, 256, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.randn(256, device="cuda")
y = torch.empty((N, 256, 64, 64), device="cuda").contiguous(memory_format=torch.channels_last)
xs, ws = split(lib, x), split(lib, w, hip.X3_WSCALE)
for cfg in (16, PCFG):
    for _ in range(3):
        run_x3(lib, g, xs, ws, b, y, cfg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run_x3(lib, g, xs, ws, b, y, cfg)
    e1.record()
    e1.synchronize()
    print("cfg %d: %.1f us per launch" % (cfg, e0.elapsed_time(e1) * 100))
tiles = N * 64 * 64 // 256
nw = min(2048, tiles * NWB)
buf = (ctypes.c_longlong * (nw * 16))()
hip.check(lib.cg_debug_fetch(buf, nw * 16), "cg_debug_fetch")
d = np.frombuffer(buf, dtype=np.int64).reshape(nw, 16)
nk = int(d[0, 4])
print("waves %d (%d per block), K slices %d; matrix-pipe demand per slice and SIMD: %d waves x %d MFMA x 32 cycles = 3072 cycles" % (nw, NWB, nk, NWB // 4, 384 // NWB))
print("%-10s %10s %10s %10s %10s %10s %8s" % ("waves", "loop/slice", "dma-wait", "barrier", "dma-issue", "rest", "GHz"))
for name, sel in (("all", d[:, 5] >= 0), ("first half", d[:, 5] < NWB // 2), ("second half", d[:, 5] >= NWB // 2)):
    r = d[sel].astype(np.float64)
    tot, vm, bar, dma = (r[:, k].mean() / nk for k in range(4))
    ghz = (r[:, 0] / (r[:, 6] * 10.0)).mean()      # cycles per ns
    print("%-10s %10.0f %10.0f %10.0f %10.0f %10.0f %8.2f" % (name, tot, vm, bar, dma, tot - vm - bar - dma, ghz))
print("(cycles per K slice and wave; 'rest' = the four MFMA steps with their fragment reads; loop/slice x %d slices / clock = kernel time)" % nk)

# one slice (kt = nk / 2) of one block, wave by wave: absolute shader-clock stamps relative to the block's earliest stamp
full = (ctypes.c_longlong * (4096 * 16))()
hip.check(lib.cg_debug_fetch(full, 4096 * 16), "cg_debug_fetch")
t = np.frombuffer(full, dtype=np.int64).reshape(4096, 16)[2048:]
names = ["slice top", "step0 issued", "step1 issued", "step2 issued", "step3 issued", "before dma-landed wait", "after wait",
         "after barrier", "after early DMA", "late DMA start", "late DMA end"]
for blk in (0, 100):
    w = t[blk * NWB:(blk + 1) * NWB, :11].astype(np.float64)
    base = w[w > 0].min()
    print("block %d (waves 0-3 issue their DMA behind the barrier, waves 4-7 at the next slice's step 0; SIMD = wave %% 4)" % blk)
    print("%-24s" % "event" + " ".join("%8s" % ("w%d" % i) for i in range(NWB)))
    for k, nmk in enumerate(names):
        print("%-24s" % nmk + " ".join("%8s" % ("%.0f" % (w[i, k] - base) if w[i, k] > 0 else "-") for i in range(NWB)))
