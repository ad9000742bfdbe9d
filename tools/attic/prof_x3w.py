"""Launch one split-precision forward tile configuration on the res-block shape a few times (for rocprofv3 --pmc).
Usage: prof_x3w.py [cfg=16] [batch=16] [reps=8]   (batch 16 = four members x batch 4, the shape of a member-batched launch)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402
from bench_x3 import split, run_x3  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
H = W = 64
Cin = Cout = 256
lib = hip.load()
g = ops.fwd_geom(N, H, W, Cin, 0, 0, 3, 3, 1, 1, Cout, 1)
x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.randn(Cout, device="cuda")
y = torch.empty((N, Cout, H, W), device="cuda").contiguous(memory_format=torch.channels_last)
xs, ws = split(lib, x), split(lib, w, hip.X3_WSCALE)
for _ in range(reps):
    run_x3(lib, g, xs, ws, b, y, cfg)
torch.cuda.synchronize()
print("cfg", cfg, "batch", N, "done; algorithmic bytes per launch: in %.1f MB (hi+lo fp16) + weights %.1f MB + out %.1f MB"
      % (x.numel() * 4 / 1e6, w.numel() * 4 / 1e6, y.numel() * 4 / 1e6))
