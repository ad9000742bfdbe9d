"""Run the split-precision forward kernel on one shape a few times (for rocprofv3 --pmc).  Usage: prof_x3.py <shape> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import council_gan_amd as cga  # noqa: F401
from council_gan_amd import hip, ops
from bench_conv import SHAPES
si = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
name, N, H, W, Cin, Cout, K, stride, pad, up = SHAPES[si]
x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(Cout, Cin, K, K, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.randn(Cout, device="cuda")
with torch.no_grad():
    xs, ws = ops.split_f16(x), ops.split_f16(w, hip.X3_WSCALE)
    for _ in range(reps):
        y = ops.conv2d_x3(xs, ws, Cout, K, K, b, stride, pad, "relu", upsample=bool(up))
torch.cuda.synchronize()
print(name, "done")
