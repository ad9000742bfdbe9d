"""Launch the split-precision weight gradient on the member-batched res-block shape a few times (for rocprofv3 --pmc).
Usage: prof_wgrad_x3.py [mode=2] [batch=16] [members=4] [reps=8]   (mode: cg_conv2d_wgrad_x3_bm256 -- 0 = 128x128 tile, 1 = 256x128)"""
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nm = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
HW, Cin, Cout, K = 64, 256, 256, 3
lib = hip.load()
lib.cg_conv2d_wgrad_x3_bm256(mode)
g = ops.fwd_geom(N, HW, HW, Cin, 0, 0, K, K, 1, 1, Cout, 0)
x = torch.randn(N, Cin, HW, HW, device="cuda").contiguous(memory_format=torch.channels_last)
dz = (torch.randn(N, Cout, HW, HW, device="cuda") * 1e-3).contiguous(memory_format=torch.channels_last)
nw = Cout * Cin * K * K
stride_el = nw + Cout + 32
grp = hip.Group(nm, 0, stride_el)
flat = torch.zeros(nm * stride_el, device="cuda")
with torch.no_grad():
    xs, dzs = ops.split_f16_dynamic(x), ops.split_f16_dynamic(dz)
    wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace_g(byref(g), byref(grp)))
    for _ in range(reps):
        hip.check(lib.cg_conv2d_wgrad_x3_g(byref(g), byref(grp), xs.hi_ptr(), xs.lo, xs.scale_ptr(), dzs.hi_ptr(), dzs.lo,
                                           dzs.scale_ptr(), hip.ptr(flat[:nw]), hip.ptr(flat[nw:]), 0, hip.ptr(wsb), wsb.numel(),
                                           hip.stream()), "wgrad")
torch.cuda.synchronize()
print("mode", mode, "batch", N, "members", nm, "done; algorithmic bytes per launch: x %.1f MB + dz %.1f MB (hi+lo fp16) + dw %.1f MB"
      % (x.numel() * 4 / 1e6, dz.numel() * 4 / 1e6, nm * nw * 4 / 1e6))
