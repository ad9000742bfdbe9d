"""Feasibility probe: do this library's ctypes-launched kernels (forward, autograd backward on the engine's thread, the
flat Adam step, a side stream) capture into a hipGraph through torch.cuda.graph and replay correctly with new inputs?
Also measures launch cost: eager enqueue vs graph replay.   Usage: python tools/probe_graph.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402
from council_gan_amd import ops  # noqa: E402

torch.manual_seed(0)
dev = 'cuda:0'
convs = [torch.nn.Conv2d(64, 64, 3, 1, 1) for _ in range(6)]
opt = cga.FlatAdam([p for c in convs for p in c.parameters()], lr=1e-3)
opt.materialize(dev)
mgr = ops.SplitWeights(opt)
x_static = torch.randn(4, 64, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
side = torch.cuda.Stream()


def step():
    opt.zero_grad()
    mgr.refresh()
    h = x_static
    for c in convs:
        st = []
        h = ops.conv2d(h, c.weight, c.bias, 1, 1, 'none', stats=st, wmgr=mgr)
        h = ops.instance_norm(h, act='relu', stats=st, want_split=True)
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        aux = ops.avgpool3s2(h.detach())
    main.wait_stream(side)
    loss = ops.l1_mean(h, torch.zeros_like(h))
    loss.backward()
    opt.step()
    return loss, aux


def run_eager(xs):
    out = []
    for x in xs:
        x_static.copy_(x)
        l, a = step()
        out.append((float(l), float(a.double().sum())))
    return out


xs = [torch.randn_like(x_static) for _ in range(4)]
w0 = opt.flat['data'].clone()
m0, v0 = opt.flat['m'].clone(), opt.flat['v'].clone()
ref = run_eager(xs)
w_ref = opt.flat['data'].clone()
# reset and do the same with a captured graph (Adam's step count is a host integer baked into the launch: keep the probe to
# what is bit-comparable -- replay iteration k with the weights / moments / step count of eager iteration k)
opt.flat['data'].copy_(w0); opt.flat['m'].copy_(m0); opt.flat['v'].copy_(v0)
opt._steps = [0] * len(opt._steps)
opt.version += 1
x_static.copy_(xs[0])
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    opt.flat['data'].copy_(w0)
    with torch.cuda.graph(g, stream=s):
        loss_g, aux_g = step()
torch.cuda.synchronize()
print("captured ok")
opt.flat['data'].copy_(w0); opt.flat['m'].copy_(m0); opt.flat['v'].copy_(v0)
got = []
x_static.copy_(xs[0])
g.replay()
torch.cuda.synchronize()
got.append((float(loss_g), float(aux_g.double().sum())))
print("eager  it0:", ref[0])
print("replay it0:", got[0], "(step count 1 baked: only iteration 0 is comparable)")
ok = abs(got[0][0] - ref[0][0]) <= 1e-6 * abs(ref[0][0]) and abs(got[0][1] - ref[0][1]) <= 1e-6 * abs(ref[0][1])
print("MATCH" if ok else "MISMATCH")
# timing: eager enqueue vs replay
for name, fn in (("eager", lambda: step()), ("replay", lambda: g.replay())):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-6s host %.3f ms/iter   until GPU done %.3f ms/iter" % (name, 1e3 * (t1 - t0) / 20, 1e3 * (t2 - t0) / 20))
