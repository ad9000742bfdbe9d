import sys, os
sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F
import council_gan_amd as cga
from council_gan_amd import ops, hip
torch.manual_seed(0)
for (N,H,W,Cin,Cout,K,s,p,up) in [(2,8,8,32,1,1,1,0,0),(2,4,4,32,32,1,1,0,0),(2,8,8,32,32,1,1,0,0),(2,16,16,64,12,1,1,0,0),(2,8,8,32,64,3,1,1,0),(2,8,8,64,64,3,1,1,1),(2,16,16,32,8,4,2,1,0),(3,8,8,512,512,1,1,0,0),(2,8,8,512,1,1,1,0,0)]:
    x = torch.randn(N,Cin,H,W).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout,Cin,K,K)*0.02).cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout).cuda()
    ref = F.conv2d(F.pad(F.interpolate(x.cpu().double(), scale_factor=2) if up else x.cpu().double(), (p,)*4), w.cpu().double(), b.cpu().double(), stride=s)
    with torch.no_grad():
        ops.X3_FORWARD = False
        y32 = ops.conv2d(x, w, b, s, p, 'none', upsample=bool(up))
        xs = ops.split_f16(x); ws = ops.split_f16(w, hip.X3_WSCALE)
        y3 = ops.conv2d_x3(xs, ws, Cout, K, K, b, s, p, 'none', upsample=bool(up))
    sc = float(ref.abs().max())
    print((N,H,W,Cin,Cout,K,s,p,up), "fp32 err %.2e  x3 err %.2e" % (float((y32.cpu().double()-ref).abs().max())/sc, float((y3.cpu().double()-ref).abs().max())/sc))
