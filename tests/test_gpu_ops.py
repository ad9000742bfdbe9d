"""GPU parity of every HIP operator against a plain PyTorch CPU reference of the same op (fp64
autograd), forward and backward, on the closed shape set of SURVEY.md 2.1 plus ragged / odd /
tiny-channel cases.  Tolerance: max-abs error / max-abs reference <= 2e-5 (fp32 round-off of an
fmaf chain; the 1e-3 budget of the north star is for whole-network outputs)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# kernels / tile configurations written after the round's last GPU call (forward configurations 22-27: channel-slice-major K
# order, 256x64 tile; the 256x256 weight-gradient tile) have not run on a GPU yet: they join the suite with CG_TEST_EXPERIMENTAL=1
_EXPERIMENTAL = pytest.mark.skipif(os.environ.get("CG_TEST_EXPERIMENTAL") != "1", reason="experimental kernel (CG_TEST_EXPERIMENTAL=1)")

TOL = 2e-5


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def dev(t):
    return t.float().cuda()


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.fixture(scope="module")
def cga():
    import council_gan_amd
    council_gan_amd.hip.load()
    return council_gan_amd


def ref_act(y, act):
    return {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.2), "tanh": torch.tanh}[act](y)


CONV_CASES = [
    # name, N, H, W, C1, C2, Cout, K, stride, pad, up, act
    ("3x3_fast_64to128", 2, 16, 16, 64, 0, 128, 3, 1, 1, 0, "relu"),
    ("3x3_fast_256to256_tile128", 2, 64, 64, 256, 0, 256, 3, 1, 1, 0, "none"),
    ("4x4s2_fast_64to128", 2, 32, 32, 64, 0, 128, 4, 2, 1, 0, "lrelu"),
    ("4x4s2_fast_128to256_big", 4, 64, 64, 128, 0, 256, 4, 2, 1, 0, "none"),
    ("7x7_cin3", 2, 32, 32, 3, 0, 64, 7, 1, 3, 0, "relu"),
    ("1x1_64to12_tanh", 2, 16, 16, 64, 0, 12, 1, 1, 0, 0, "tanh"),
    ("1x1_64to64_relu", 2, 64, 64, 64, 0, 64, 1, 1, 0, 0, "relu"),
    ("1x1_512to1", 2, 8, 8, 512, 0, 1, 1, 1, 0, 0, "none"),
    ("1x1_512to512", 3, 8, 8, 512, 0, 512, 1, 1, 0, 0, "none"),
    ("concat_3p3to64", 2, 16, 16, 3, 3, 64, 3, 1, 1, 0, "lrelu"),
    ("up_3x3_128to64", 2, 8, 8, 128, 0, 64, 3, 1, 1, 1, "none"),
    ("odd_3x3_8to16", 2, 9, 7, 8, 0, 16, 3, 1, 1, 0, "relu"),
    ("odd_4x4s2_8to16", 3, 9, 7, 8, 0, 16, 4, 2, 1, 0, "lrelu"),
    ("odd_up_3x3_16to8", 1, 5, 3, 16, 0, 8, 3, 1, 1, 1, "none"),
    ("4x4s2_cin3", 2, 32, 32, 3, 0, 64, 4, 2, 1, 0, "lrelu"),
    ("tiny_1x1_c4", 1, 1, 1, 4, 0, 4, 1, 1, 0, 0, "none"),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(cga, case):
    _, N, H, W, C1, C2, Cout, K, stride, pad, up, act = case
    g = torch.Generator().manual_seed(sum(map(ord, case[0])))
    x = torch.randn(N, C1, H, W, generator=g, dtype=torch.float64)
    x2 = torch.randn(N, C2, H, W, generator=g, dtype=torch.float64) if C2 else None
    w = torch.randn(Cout, C1 + C2, K, K, generator=g, dtype=torch.float64) / np.sqrt((C1 + C2) * K * K)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    x2r = x2.clone().requires_grad_(True) if C2 else None
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    inp = torch.cat((xr, x2r), 1) if C2 else xr
    if up:
        inp = F.interpolate(inp, scale_factor=2, mode="nearest")
    yr = ref_act(F.conv2d(F.pad(inp, (pad,) * 4), wr, br, stride=stride), act)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)

    xd = cl(dev(x)).requires_grad_(True)
    x2d = cl(dev(x2)).requires_grad_(True) if C2 else None
    wd = cl(dev(w)).requires_grad_(True)
    bd = dev(b).requires_grad_(True)
    yd = cga.ops.conv2d(xd, wd, bd, stride, pad, act, x2=x2d, upsample=bool(up))
    assert tuple(yd.shape) == tuple(yr.shape)
    errs = {"fwd": rel(yd, yr)}
    yd.backward(cl(dev(gy)))
    errs["dx"] = rel(xd.grad, xr.grad)
    if C2:
        errs["dx2"] = rel(x2d.grad, x2r.grad)
    errs["dw"] = rel(wd.grad, wr.grad)
    errs["db"] = rel(bd.grad, br.grad)
    assert max(errs.values()) < TOL, errs


STATS_CASES = [
    # name, N, H, W, Cin, Cout, up  (3x3, pad 1): every fp32 tile configuration pick_fwd_cfg can return for a layer that
    # is followed by an instance norm, plus the generic (Cin = 3) kernel that leaves the statistics to the norm
    ("cout64_many_tiles_128x64", 2, 64, 64, 128, 64, 1),
    ("cout64_few_tiles_64x64", 2, 16, 16, 64, 64, 0),
    ("cout128_few_tiles_64x64", 2, 64, 64, 64, 128, 0),
    ("cout256_many_tiles_128x128", 4, 64, 64, 128, 256, 0),
    ("cin3_generic", 2, 32, 32, 3, 64, 0),
]


@pytest.mark.parametrize("case", STATS_CASES, ids=[c[0] for c in STATS_CASES])
def test_conv_epilogue_statistics_feed_instance_norm(cga, case):
    """The {sum, sum of squares} an instance norm needs come out of the producing convolution's epilogue when the tile
    configuration supports it (cg_conv2d_fwd_stats): conv -> IN -> ReLU through that route must equal the torch chain,
    on the exact-fp32 kernels (the split-precision route is covered by the network-level tests)."""
    _, N, H, W, Cin, Cout, up = case
    torch.manual_seed(3)
    x = cl(torch.randn(N, Cin, H, W).cuda())
    w = cl((torch.randn(Cout, Cin, 3, 3) / np.sqrt(Cin * 9)).cuda())
    b = torch.randn(Cout).cuda()
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.relu(F.instance_norm(F.conv2d(xi.double(), w.double(), b.double(), padding=1)))
    with torch.no_grad():
        st = []
        y = cga.ops.conv2d(x, w, b, 1, 1, "none", upsample=bool(up), stats=st)
        fused = cga.ops.instance_norm(y, act="relu", stats=st)
        plain = cga.ops.instance_norm(y, act="relu")
    assert rel(plain, ref) < TOL, rel(plain, ref)
    assert rel(fused, ref) < TOL, (rel(fused, ref), st[0][1] if st else None)


def test_conv2d_accumulates_into_flat_grad(cga):
    """wgrad writes straight into a `_cg_grad` buffer (accumulate) and returns None to autograd."""
    g = torch.Generator().manual_seed(3)
    x = cl(dev(torch.randn(2, 32, 8, 8, generator=g)))
    w = cl(dev(torch.randn(64, 32, 3, 3, generator=g) * 0.1)).requires_grad_(True)
    b = dev(torch.zeros(64)).requires_grad_(True)
    w._cg_grad = torch.ones_like(w)
    b._cg_grad = torch.ones_like(b)
    y = cga.ops.conv2d(x, w, b, 1, 1, "none")
    y.sum().backward()
    assert w.grad is None and b.grad is None
    wc = w.detach().cpu().double().requires_grad_(True)
    ref_w = torch.autograd.grad(F.conv2d(x.cpu().double(), wc, None, padding=1).sum(), wc)[0]
    cga.ops.wgrad_join()          # weight gradients run on a companion stream of the backward's stream (ops._companion)
    assert rel(w._cg_grad - 1, ref_w) < TOL
    assert rel(b._cg_grad - 1, torch.full((64,), 2 * 8 * 8.0)) < TOL
    assert w._cg_grad._cg_touched and b._cg_grad._cg_touched


def test_linear(cga):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 64, generator=g, dtype=torch.float64)
    w = torch.randn(256, 64, generator=g, dtype=torch.float64) / 8
    b = torch.randn(256, generator=g, dtype=torch.float64)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.relu(F.linear(xr, wr, br))
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xd, wd, bd = (dev(t).requires_grad_(True) for t in (x, w, b))
    yd = cga.ops.linear(xd, wd, bd, act="relu")
    yd.backward(dev(gy))
    errs = dict(fwd=rel(yd, yr), dx=rel(xd.grad, xr.grad), dw=rel(wd.grad, wr.grad), db=rel(bd.grad, br.grad))
    assert max(errs.values()) < TOL, errs


NORM_CASES = [("c64_relu_res", 2, 64, 16, 16, "relu", True), ("c256_none_res", 2, 256, 16, 16, "none", True),
              ("c6_odd_relu", 3, 6, 9, 7, "relu", False), ("c128_big", 2, 128, 64, 64, "relu", False),
              ("c8_lrelu", 1, 8, 5, 5, "lrelu", False)]


@pytest.mark.parametrize("case", NORM_CASES, ids=[c[0] for c in NORM_CASES])
@pytest.mark.parametrize("affine", [False, True], ids=["in", "adain"])
def test_instance_norm(cga, case, affine):
    _, N, C, H, W, act, res = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64) * 2 + 0.7
    r = torch.randn(N, C, H, W, generator=g, dtype=torch.float64) if res else None
    P = 2 * C + 5
    params = torch.randn(N, P, generator=g, dtype=torch.float64)
    boff, goff = 3, 3 + C
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    pr = params.clone().requires_grad_(True)
    if affine:
        y = F.batch_norm(xr.view(1, N * C, H, W), None, None, pr[:, goff:goff + C].reshape(-1),
                         pr[:, boff:boff + C].reshape(-1), True, 0.1, 1e-5).view(N, C, H, W)
    else:
        y = F.instance_norm(xr, eps=1e-5)
    yr = ref_act(y, act)
    if res:
        yr = yr + rr
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)

    xd = cl(dev(x)).requires_grad_(True)
    rd = cl(dev(r)).requires_grad_(True) if res else None
    pd = dev(params).requires_grad_(True)
    if affine:
        yd = cga.ops.adain(xd, pd, goff, boff, act=act, residual=rd)
    else:
        yd = cga.ops.instance_norm(xd, act=act, residual=rd)
    yd.backward(cl(dev(gy)))
    errs = dict(fwd=rel(yd, yr), dx=rel(xd.grad, xr.grad))
    if res:
        errs["dres"] = rel(rd.grad, rr.grad)
    if affine:
        errs["dparams"] = rel(pd.grad, pr.grad)
    assert max(errs.values()) < 5e-5, errs


@pytest.mark.parametrize("outside", [False, True], ids=["layers_only", "plus_outside_consumer"])
def test_adain_layers_share_one_parameter_gradient_buffer(cga, outside, monkeypatch):
    """networks.py:303-312: the MLP's [N, P] output is handed to the decoder's AdaIN layers as disjoint column slices.  With
    ops.adain_param_fork every layer's backward writes its columns into ONE shared buffer (ops.ParamGrad) and a fork node reports
    it after the last of them -- no zero-filled [N, P] gradient per layer, no additions by the autograd engine.  The matrix's
    gradient (and everything upstream of it: here a scale factor standing for the MLP) must be BIT-IDENTICAL to the per-layer
    form, also when somebody differentiates through the matrix outside the layers (the reference keeps m.weight / m.bias views),
    and columns no layer owns stay zero."""
    from council_gan_amd import ops
    g = torch.Generator().manual_seed(11)
    N, H, W = 3, 12, 10
    chans = [32, 64, 8]
    P = 2 * sum(chans) + 7                         # 7 columns nobody owns
    w = dev(torch.randn(N, P, generator=g)).requires_grad_(True)
    xs = [cl(dev(torch.randn(N, c, H, W, generator=g))) for c in chans]
    gys = [cl(dev(torch.randn(N, c, H, W, generator=g))) for c in chans]
    made = []
    orig = ops.ParamGrad.buffer
    monkeypatch.setattr(ops.ParamGrad, "buffer", lambda self, like: (made.append(self), orig(self, like))[1])

    def run(on):
        monkeypatch.setattr(ops, "ADAIN_FORK", on)
        made.clear()
        w.grad = None
        params, pg = ops.adain_param_fork(w * 1.5)
        assert (pg is not None) == on
        off, outs = 3, []
        for c, x in zip(chans, xs):
            outs.append(ops.adain(x, params, off + c, off, act='relu', pgrad=pg))      # gamma at off + c, beta at off
            off += 2 * c
        roots, ups = list(outs), list(gys)
        if outside:
            roots.append((params[:, 1:5] * 0.25).sum().view(1))
            ups.append(dev(torch.ones(1)))
        torch.autograd.backward(roots, ups)
        torch.cuda.synchronize()
        return w.grad.clone(), len(made), (pg.buf if pg is not None else None)

    g1, n1, left = run(True)
    g0, n0, _ = run(False)
    assert n1 == len(chans) and n0 == 0 and left is None        # every layer wrote into the shared buffer; the fork took it
    assert torch.equal(g1, g0), float((g1 - g0).abs().max())
    tail = g1[:, 3 + 2 * sum(chans):]
    assert float(tail.abs().max()) == 0.0
    if outside:
        assert float(g1[:, 1:3].abs().min()) == 0.375              # 1.5 * 0.25 on the columns only the outside consumer reads


@pytest.mark.parametrize("case", [("c64_relu", 2, 64, 16, 16, "relu"), ("c256_none", 3, 256, 16, 8, "none"),
                                  ("c128_tails", 2, 128, 64, 64, "relu")], ids=lambda c: c[0])
@pytest.mark.parametrize("affine", [False, True], ids=["in", "adain"])
def test_instance_norm_backward_in_split_form(cga, case, affine):
    """cg_instnorm_bwd_split: dx written directly as the {hi, lo} planes of scale * dx, the scale chosen from an upper bound
    of max |dx| before dx exists.  hi + lo over the scale must be dx to 22 bits, the scaled peak must sit within fp16's
    range and not more than two binades below the [4096, 8192) window of the exact-maximum scale; dgamma / dbeta unchanged.
    `c128_tails`: a heavy-tailed upstream gradient (a few spikes 1e4 x the bulk)."""
    from council_gan_amd import hip, ops
    _, N, C, H, W, act = case
    lib = hip.load()
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64) * 2 + 0.7
    P = 2 * C + 5
    params = torch.randn(N, P, generator=g, dtype=torch.float64)
    boff, goff = 3, 3 + C
    gy = torch.randn(N, C, H, W, generator=g, dtype=torch.float64) * 1e-3
    if "tails" in case[0]:
        gy.view(-1)[torch.randint(0, gy.numel(), (20,), generator=g)] *= 1e4
    xr, pr = x.clone().requires_grad_(True), params.clone().requires_grad_(True)
    if affine:
        y = F.batch_norm(xr.view(1, N * C, H, W), None, None, pr[:, goff:goff + C].reshape(-1),
                         pr[:, boff:boff + C].reshape(-1), True, 0.1, 1e-5).view(N, C, H, W)
    else:
        y = F.instance_norm(xr, eps=1e-5)
    ref_act(y, act).backward(gy)
    xd, gyd, pd = cl(dev(x)), cl(dev(gy)), dev(params)
    HW = H * W
    mean = torch.empty(N * C, device="cuda")
    rstd = torch.empty_like(mean)
    ws = hip.workspace(lib.cg_instnorm_workspace(N, HW, C))
    hip.check(lib.cg_instnorm_stats(hip.ptr(xd), N, HW, C, 1e-5, hip.ptr(mean), hip.ptr(rstd), hip.ptr(ws), ws.numel(),
                                    hip.stream()), "stats")
    need = lib.cg_instnorm_bwd_split_workspace(N, HW, C)
    assert need > 0
    ws = hip.workspace(need)
    buf = torch.zeros(2 * xd.numel(), dtype=torch.float16, device="cuda")
    state = torch.zeros(hip.SPLIT_STATE_FLOATS, device="cuda")
    dx32 = torch.empty_like(xd)
    dpar = torch.zeros_like(pd)
    gp = ops._off(pd, goff) if affine else None
    bp = ops._off(pd, boff) if affine else None
    dgp = ops._off(dpar, goff) if affine else None
    dbp = ops._off(dpar, boff) if affine else None
    hip.check(lib.cg_instnorm_bwd_split(hip.ptr(gyd), hip.ptr(xd), hip.ptr(mean), hip.ptr(rstd), gp, bp, P if affine else C,
                                        hip.ptr(buf), ops.x3_lo(xd.numel()), hip.ptr(state), hip.ptr(dx32), dgp, dbp, N, HW, C,
                                        ops.ACT[act], hip.ptr(ws), ws.numel(), hip.stream()), "cg_instnorm_bwd_split")
    torch.cuda.synchronize()
    scale = float(state[1])
    got = ops.SplitTensor(buf, xd.shape, state=state).to_float().double().cpu() / scale      # physical NHWC order
    want = xr.grad.permute(0, 2, 3, 1).reshape(-1)
    peak = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2.0 ** -21 * peak + 1e-12, (float((got - want).abs().max()), peak)
    assert rel(dx32, xr.grad) < 5e-5
    assert scale > 0 and np.log2(scale) == round(np.log2(scale))
    assert 1024.0 <= peak * scale < 8192.0 * 1.0001, (peak, scale, float(state[0]))
    if affine:
        assert rel(dpar[:, goff:goff + C], pr.grad[:, goff:goff + C]) < 5e-5
        assert rel(dpar[:, boff:boff + C], pr.grad[:, boff:boff + C]) < 5e-5


def test_layer_norm(cga):
    g = torch.Generator().manual_seed(9)
    N, C, H, W = 3, 16, 9, 7
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64) * 1.5 + 0.3
    gamma = torch.rand(C, generator=g, dtype=torch.float64)
    beta = torch.randn(C, generator=g, dtype=torch.float64)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    mean = xr.reshape(N, -1).mean(1).view(N, 1, 1, 1)
    std = xr.reshape(N, -1).std(1).view(N, 1, 1, 1)
    yr = (xr - mean) / (std + 1e-5) * gr.view(1, -1, 1, 1) + br.view(1, -1, 1, 1)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xd = cl(dev(x)).requires_grad_(True)
    gd, bd = dev(gamma).requires_grad_(True), dev(beta).requires_grad_(True)
    yd = cga.ops.layer_norm(xd, gd, bd)
    yd.backward(cl(dev(gy)))
    errs = dict(fwd=rel(yd, yr), dx=rel(xd.grad, xr.grad), dg=rel(gd.grad, gr.grad), db=rel(bd.grad, br.grad))
    assert max(errs.values()) < 5e-5, errs


@pytest.mark.parametrize("shape", [(2, 3, 32, 32), (2, 6, 9, 7), (1, 3, 5, 5), (2, 3, 256, 256)])
def test_avgpool_and_upsample(cga, shape):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(*shape, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr = F.avg_pool2d(xr, 3, stride=2, padding=1, count_include_pad=False)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(gy)
    xd = cl(dev(x)).requires_grad_(True)
    yd = cga.ops.avgpool3s2(xd)
    yd.backward(cl(dev(gy)))
    assert rel(yd, yr) < TOL and rel(xd.grad, xr.grad) < TOL
    xr2 = x.clone().requires_grad_(True)
    ur = F.interpolate(xr2, scale_factor=2, mode="nearest")
    gu = torch.randn(ur.shape, generator=g, dtype=torch.float64)
    ur.backward(gu)
    xd2 = cl(dev(x)).requires_grad_(True)
    ud = cga.ops.upsample2x(xd2)
    ud.backward(cl(dev(gu)))
    assert rel(ud, ur) < TOL and rel(xd2.grad, xr2.grad) < TOL


def test_global_avgpool(cga):
    x = torch.randn(3, 40, 9, 7, dtype=torch.float64)
    yd = cga.ops.global_avgpool(cl(dev(x)))
    assert rel(yd, F.adaptive_avg_pool2d(x, 1)) < TOL


@pytest.mark.parametrize("od,k", [(3, 3), (3, 1), (1, 2)])
def test_mask_blend(cga, od, k):
    g = torch.Generator().manual_seed(13)
    N, H, W = 2, 17, 9
    new_x = torch.tanh(torch.randn(N, od * k + k, H, W, generator=g, dtype=torch.float64) * 0.3)
    im = torch.rand(N, od, H, W, generator=g, dtype=torch.float64) * 2 - 1
    nr = new_x.clone().requires_grad_(True)
    mask = (torch.tanh(10 * nr[:, -k:]) + 1) / 2
    out = im
    for j in range(k):
        m = mask[:, j:j + 1]
        out = (1 - m) * out + m * nr[:, od * j:od * (j + 1)]
    g_im = torch.randn(out.shape, generator=g, dtype=torch.float64)
    g_m = torch.randn(mask.shape, generator=g, dtype=torch.float64)
    (out * g_im).sum().backward(retain_graph=True)
    grad_im_only = nr.grad.clone()
    nr.grad = None
    ((out * g_im).sum() + (mask * g_m).sum()).backward()
    nd = cl(dev(new_x)).requires_grad_(True)
    od_, md = cga.ops.mask_blend(nd, cl(dev(im)), od, k)
    assert rel(od_, out) < TOL and rel(md, mask) < TOL
    ((od_ * cl(dev(g_im))).sum() + (md * cl(dev(g_m))).sum()).backward()
    assert rel(nd.grad, nr.grad) < 5e-5
    nd2 = cl(dev(new_x)).requires_grad_(True)
    o2, _ = cga.ops.mask_blend(nd2, cl(dev(im)), od, k)
    (o2 * cl(dev(g_im))).sum().backward()          # mask output unused -> d_mask is None
    assert rel(nd2.grad, grad_im_only) < 5e-5


def test_lsgan(cga):
    g = torch.Generator().manual_seed(15)
    B = 2
    outs = [torch.randn(3 * B, 1, 8, 8, generator=g, dtype=torch.float64),
            torch.randn(3 * B, 1, 4, 4, generator=g, dtype=torch.float64)]
    tgt = [0.0] * B + [1.0] * (2 * B)
    wt = [3.0] * B + [1.0] * B + [2.0] * B
    refs = [o.clone().requires_grad_(True) for o in outs]
    loss = 0
    for o in refs:
        loss = loss + 3 * torch.mean(o[:B] ** 2) + torch.mean((o[B:2 * B] - 1) ** 2) + 2 * torch.mean((o[2 * B:] - 1) ** 2)
    (loss * 0.7).backward()
    ds = [dev(o).requires_grad_(True) for o in outs]
    ld = cga.ops.lsgan_loss(ds, dev(torch.tensor(tgt)), dev(torch.tensor(wt)), B)
    (ld * 0.7).backward()
    assert rel(ld, loss) < TOL
    for d, r in zip(ds, refs):
        assert rel(d.grad, r.grad) < TOL


@pytest.mark.parametrize("use_abs,use_square", [(False, True), (True, False), (True, True)])
def test_focus_loss(cga, use_abs, use_square):
    from oracle import council_oracle as O
    g = torch.Generator().manual_seed(17)
    m = torch.rand(2, 3, 19, 11, generator=g, dtype=torch.float64)
    mr = m.clone().requires_grad_(True)
    zo = O.mask_zero_one(mr, 0.5, 0.01)
    small = O.mask_small(mr, use_abs, use_square)
    tv = O.mask_tv(mr)
    total = 0.5 * zo + 57 * small + 2.2 * tv
    (total * 1.3).backward()
    md = cl(dev(m)).requires_grad_(True)
    td, parts = cga.ops.focus_loss(md, 0.5, 0.01, 0.5, 57, 2.2, use_abs, use_square)
    (td * 1.3).backward()
    assert rel(td, total) < TOL
    assert rel(parts, torch.stack([zo, small, tv])) < TOL
    assert rel(md.grad, mr.grad) < 5e-5


def test_l1_mean(cga):
    a = torch.randn(2, 3, 9, 7, dtype=torch.float64)
    b = torch.randn(2, 3, 9, 7, dtype=torch.float64)
    ar = a.clone().requires_grad_(True)
    l = torch.mean(torch.abs(ar - b))
    l.backward()
    ad = cl(dev(a)).requires_grad_(True)
    ld = cga.ops.l1_mean(ad, cl(dev(b)))
    ld.backward()
    assert rel(ld, l) < TOL and rel(ad.grad, ar.grad) < TOL


def test_adam_matches_torch(cga):
    """FlatAdam == torch.optim.Adam over 3 steps, including a parameter that never gets a gradient."""
    g = torch.Generator().manual_seed(19)
    shapes = [(8, 4, 3, 3), (8,), (16, 8), (5,)]
    ps = [torch.randn(*s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [torch.nn.Parameter(p.clone()) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=1e-2, betas=(0.5, 0.999), weight_decay=1e-4)
    o = cga.FlatAdam(mine, lr=1e-2, betas=(0.5, 0.999), weight_decay=1e-4)
    o.materialize("cuda")
    for step in range(3):
        o.zero_grad()
        o_ref.zero_grad()
        for k, (r, m) in enumerate(zip(ref, mine)):
            if k == 3:
                continue          # untouched parameter: torch skips it (grad is None)
            gr = torch.randn(*shapes[k], generator=g)
            r.grad = gr.clone()
            m._cg_grad.copy_(gr)
            m._cg_grad._cg_touched = True
        o.step()
        o_ref.step()
    for r, m in zip(ref, mine):
        assert float((r.detach() - m.detach().cpu()).abs().max()) < 1e-6
    sd = o.state_dict()
    assert set(sd["state"]) == {0, 1, 2} and float(sd["state"][0]["step"]) == 3.0
    assert tuple(sd["state"][0]["exp_avg"].shape) == shapes[0]
    ref_sd = o_ref.state_dict()
    assert rel(sd["state"][0]["exp_avg_sq"], ref_sd["state"][0]["exp_avg_sq"]) < 1e-4


def test_loss_match_ring(cga):
    from ctypes import c_void_p
    hip = cga.hip
    lib = hip.load()
    n = 100
    rg, rc, w = torch.ones(n, device="cuda"), torch.ones(n, device="cuda"), torch.zeros(1, device="cuda")
    hg, hc = np.ones(n), np.ones(n)
    for it in range(5):
        a, c = 1.9 + 0.01 * it, 2.5 - 0.1 * it
        hip.check(lib.cg_ring_push(hip.ptr(rg), n, it, hip.ptr(torch.tensor([a], device="cuda")), hip.stream()), "push")
        hip.check(lib.cg_loss_match(hip.ptr(rg), hip.ptr(rc), n, it, hip.ptr(torch.tensor([c], device="cuda")),
                                    hip.ptr(w), hip.stream()), "match")
        hg[it % n], hc[it % n] = np.float32(a), np.float32(c)
        assert abs(float(w) - np.mean(hg) / np.mean(hc)) < 1e-6


def test_no_cpu_fallback(cga):
    """A CPU tensor must be rejected loudly, never computed by some fallback."""
    with pytest.raises(cga.hip.HipError):
        cga.ops.conv2d(torch.randn(1, 4, 4, 4), torch.randn(4, 4, 1, 1).cuda(), None)


# ------------------------------------------------------------------------------------------
# thin-input layers on the spatial-tile kernel (conv_fwd_thin_kernel, tile configuration 40 / cg_conv2d_fwd_thin)
# ------------------------------------------------------------------------------------------
THIN_CASES = [
    # name, N, H, W, C1, C2, K, stride, pad, act  (sizes that are multiples of no tile edge)
    ("gen_7x7_3to64", 2, 40, 24, 3, 0, 7, 1, 3, "none"),
    ("dis_4x4s2_3to64", 3, 36, 20, 3, 0, 4, 2, 1, "lrelu"),
    ("council_dis_3x3_3+3to64", 2, 20, 33, 3, 3, 3, 1, 1, "lrelu"),
    ("3x3_3to64", 1, 16, 16, 3, 0, 3, 1, 1, "relu"),
    ("1x1_12to64", 2, 17, 16, 12, 0, 1, 1, 0, "none"),
    ("gen_7x7_3to64_full_tiles", 4, 32, 32, 3, 0, 7, 1, 3, "none"),
]


@pytest.mark.parametrize("case", THIN_CASES, ids=[c[0] for c in THIN_CASES])
def test_thin_input_convolution(cga, case):
    """Forced (configuration 40) against fp64 and against the generic kernel; then through ops.conv2d with the switch on:
    a member-batched launch equals the members' own launches bit for bit, the reported block maxima bound the output
    exactly, and layers that do not match a variant keep their kernel."""
    from ctypes import byref
    from council_gan_amd import hip, ops
    _, N, H, W, C1, C2, K, stride, pad, act = case
    lib = hip.load()
    g = torch.Generator().manual_seed(sum(map(ord, case[0])))
    Ct = C1 + C2
    x = torch.randn(N, Ct, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(64, Ct, K, K, generator=g, dtype=torch.float64) / np.sqrt(Ct * K * K)
    b = torch.randn(64, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.pad(x, (pad,) * 4), w, b, stride=stride)
    ref = {"none": ref, "relu": F.relu(ref), "lrelu": F.leaky_relu(ref, 0.2)}[act]
    x1d = cl(dev(x[:, :C1]))
    x2d = cl(dev(x[:, C1:])) if C2 else None
    wd, bd = cl(dev(w)), dev(b)
    geom = ops.fwd_geom(N, H, W, C1, C2, 0, K, K, stride, pad, 64, ops.ACT[act])

    def forced(cfg):
        y = torch.full((N, 64, geom.Ho, geom.Wo), float("nan"), device="cuda").contiguous(memory_format=torch.channels_last)
        hip.check(lib.cg_conv2d_fwd_tile(byref(geom), hip.ptr(x1d), hip.ptr(x2d), hip.ptr(wd), hip.ptr(bd), hip.ptr(y), cfg,
                                         hip.stream()), "cg_conv2d_fwd_tile")
        return y
    thin, generic = forced(40), forced(1)
    assert rel(thin, ref) < 3e-6, rel(thin, ref)
    assert rel(thin, generic) < 2e-6
    # the same layer on the fp16 x 3 MFMA (conv_fwd_thin_x3_kernel, configuration 41): what the split-precision datapath runs
    takes_x3 = bool(lib.cg_conv2d_fwd_thin_x3_ok(byref(geom)))
    assert takes_x3 == (K * K * Ct != 12)          # every thin layer but the 12 -> 64 1x1 (faster on the fp32 kernel)
    thin_x3 = forced(41)
    assert rel(thin_x3, ref) < 2e-5, rel(thin_x3, ref)

    prev = lib.cg_conv2d_fwd_thin(1)
    try:
        with torch.no_grad():
            flags = (ops.X3_FORWARD, ops.X3_DYNAMIC_INPUT)
            ops.X3_FORWARD = ops.X3_DYNAMIC_INPUT = True       # the split-precision consumers ask producers for block maxima
            try:
                y = ops.conv2d(x1d, wd, bd, stride, pad, act, x2=x2d)
                thin_flag, ops.THIN_X3 = ops.THIN_X3, False       # CG_THIN_X3=0: the exact-fp32 kernel under the split datapath too
                try:
                    y32path = ops.conv2d(x1d, wd, bd, stride, pad, act, x2=x2d)
                finally:
                    ops.THIN_X3 = thin_flag
            finally:
                ops.X3_FORWARD, ops.X3_DYNAMIC_INPUT = flags
            assert torch.equal(y, thin_x3 if takes_x3 else thin) and torch.equal(y32path, thin)
            for out in (y, y32path):
                state, nslots = out._cg_amax
                assert float(state[2:2 + nslots].max()) == float(out.abs().max())
            ops.X3_FORWARD = False                                  # the exact-fp32 datapath keeps the fp32 kernel
            try:
                yx = ops.conv2d(x1d, wd, bd, stride, pad, act, x2=x2d)
            finally:
                ops.X3_FORWARD = flags[0]
            assert torch.equal(yx, thin)
            # a layer no variant covers (32 output channels) keeps the generic kernel
            w32 = cl(dev(w[:32]))
            y32 = ops.conv2d(x1d, w32, bd[:32], stride, pad, act, x2=x2d)
            assert rel(y32, ref[:, :32]) < 3e-6
        # member-batched: two members' weights one pool stride apart, their samples stacked along the batch
        if N % 2 == 0:
            w2 = torch.randn(64, Ct, K, K, generator=g, dtype=torch.float64) / np.sqrt(Ct * K * K)
            b2 = torch.randn(64, generator=g, dtype=torch.float64)
            nw = wd.numel()
            stride_el = nw + 64 + 32
            pool = torch.zeros(2 * stride_el, device="cuda")
            for m, (wm_, bm_) in enumerate(((w, b), (w2, b2))):
                pool[m * stride_el:m * stride_el + nw] = dev(wm_).permute(0, 2, 3, 1).reshape(-1)
                pool[m * stride_el + nw:m * stride_el + nw + 64] = dev(bm_)
            grp = hip.Group(2, 0, stride_el)
            yg = torch.full_like(thin, float("nan"))
            hip.check(lib.cg_conv2d_fwd_g(byref(geom), byref(grp), hip.ptr(x1d), hip.ptr(x2d), hip.ptr(pool[:nw]),
                                          hip.ptr(pool[nw:nw + 64]), hip.ptr(yg), None, 0, None, None, None, hip.stream()),
                      "cg_conv2d_fwd_g")
            ygx = torch.full_like(thin, float("nan"))
            if takes_x3:
              hip.check(lib.cg_conv2d_fwd_thin_x3_g(byref(geom), byref(grp), hip.ptr(x1d), hip.ptr(x2d), hip.ptr(pool[:nw]),
                                                    hip.ptr(pool[nw:nw + 64]), hip.ptr(ygx), None, None, hip.stream()),
                        "cg_conv2d_fwd_thin_x3_g")
            h = N // 2
            gh = ops.fwd_geom(h, H, W, C1, C2, 0, K, K, stride, pad, 64, ops.ACT[act])
            for m, (wm_, bm_) in enumerate(((w, b), (w2, b2))):
                ym = torch.full((h, 64, geom.Ho, geom.Wo), float("nan"), device="cuda").contiguous(memory_format=torch.channels_last)
                xa = x1d[m * h:(m + 1) * h]
                xb = x2d[m * h:(m + 1) * h] if C2 else None
                wmd, bmd = cl(dev(wm_)), dev(bm_)                # (named: a temporary's block may be handed out again at once)
                hip.check(lib.cg_conv2d_fwd_tile(byref(gh), hip.ptr(xa), hip.ptr(xb), hip.ptr(wmd), hip.ptr(bmd),
                                                 hip.ptr(ym), 40, hip.stream()), "cg_conv2d_fwd_tile")
                assert torch.equal(yg[m * h:(m + 1) * h], ym)
                ymx = torch.full_like(ym, float("nan"))
                hip.check(lib.cg_conv2d_fwd_tile(byref(gh), hip.ptr(xa), hip.ptr(xb), hip.ptr(wmd), hip.ptr(bmd),
                                                 hip.ptr(ymx), 41, hip.stream()), "cg_conv2d_fwd_tile")
                assert not takes_x3 or torch.equal(ygx[m * h:(m + 1) * h], ymx)
                refm = F.conv2d(F.pad(x[m * h:(m + 1) * h], (pad,) * 4), wm_, bm_, stride=stride)
                refm = {"none": refm, "relu": F.relu(refm), "lrelu": F.leaky_relu(refm, 0.2)}[act]
                assert rel(ym, refm) < 3e-6
    finally:
        lib.cg_conv2d_fwd_thin(prev)


@pytest.mark.parametrize("case", THIN_CASES, ids=[c[0] for c in THIN_CASES])
def test_thin_input_weight_gradient(cga, case):
    """conv_wgrad_thin_kernel (cg_conv2d_wgrad_thin) against fp64 and the generic kernel: weight and bias gradient, for one
    member and for a member-batched launch, overwrite and accumulate."""
    from ctypes import byref
    from council_gan_amd import hip, ops
    _, N, H, W, C1, C2, K, stride, pad, _act = case
    lib = hip.load()
    g = torch.Generator().manual_seed(1 + sum(map(ord, case[0])))
    Ct = C1 + C2
    x = torch.randn(N, Ct, H, W, generator=g, dtype=torch.float64)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    gy = torch.randn(N, 64, Ho, Wo, generator=g, dtype=torch.float64) * 1e-2
    x1d = cl(dev(x[:, :C1]))
    x2d = cl(dev(x[:, C1:])) if C2 else None
    gyd = cl(dev(gy))
    geom = ops.fwd_geom(N, H, W, C1, C2, 0, K, K, stride, pad, 64, 0)
    nw = 64 * Ct * K * K

    def ref(n0, n1):
        wz = torch.zeros(64, Ct, K, K, dtype=torch.float64, requires_grad=True)
        bz = torch.zeros(64, dtype=torch.float64, requires_grad=True)
        F.conv2d(F.pad(x[n0:n1], (pad,) * 4), wz, bz, stride=stride).backward(gy[n0:n1])
        return wz.grad, bz.grad

    def run(nmember, accumulate):
        stride_el = nw + 64 + 32
        flat = torch.full((nmember * stride_el,), 0.5 if accumulate else float("nan"), device="cuda")
        grp = hip.Group(nmember, 0, stride_el)
        wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace_g(byref(geom), byref(grp)))
        hip.check(lib.cg_conv2d_wgrad_g(byref(geom), byref(grp), hip.ptr(x1d), hip.ptr(x2d), hip.ptr(gyd), hip.ptr(flat[:nw]),
                                        hip.ptr(flat[nw:]), accumulate, hip.ptr(wsb), wsb.numel(), hip.stream()), "wgrad_g")
        torch.cuda.synchronize()
        out = []
        for m in range(nmember):
            o = flat[m * stride_el:(m + 1) * stride_el] - (0.5 if accumulate else 0.0)
            out.append((o[:nw].view(64, K, K, Ct).permute(0, 3, 1, 2).cpu().double(), o[nw:nw + 64].cpu().double()))
        return out

    members = [1] + ([2] if N % 2 == 0 else [3] if N % 3 == 0 else [])
    prev = lib.cg_conv2d_wgrad_thin(0)
    try:
        base = {n: run(n, 0) for n in members}
        lib.cg_conv2d_wgrad_thin(1)
        thin = {n: run(n, 0) for n in members}
        thin_acc = run(1, 1)
    finally:
        lib.cg_conv2d_wgrad_thin(prev)
    for n in members:
        per = N // n
        for m in range(n):
            rw, rb = ref(m * per, (m + 1) * per)
            for got in (base[n][m], thin[n][m]):
                assert rel(got[0], rw) < 5e-6 and rel(got[1], rb) < 5e-6, (n, m, rel(got[0], rw), rel(got[1], rb))
    rw, rb = ref(0, N)
    assert rel(thin_acc[0][0], rw) < 2e-5 and rel(thin_acc[0][1], rb) < 2e-5     # (0.5 + g) - 0.5 in fp32
    # the activation backward folded into the kernel's dz load (cg_conv2d_wgrad_act_g: the discriminators' first layers in their
    # own updates need no data gradient) against cg_act_bwd + cg_conv2d_wgrad_g: the same bits, for lrelu and relu
    assert lib.cg_conv2d_wgrad_act_ok(byref(geom)) == 1
    yd = cl(dev(torch.randn(N, 64, Ho, Wo, generator=g, dtype=torch.float64)))
    for act in (ops.ACT["lrelu"], ops.ACT["relu"]):
        for nmember in members:
            stride_el = nw + 64 + 32
            grp = hip.Group(nmember, 0, stride_el)
            wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace_g(byref(geom), byref(grp)))
            dz = torch.empty_like(gyd)
            hip.check(lib.cg_act_bwd(hip.ptr(gyd), hip.ptr(yd), hip.ptr(dz), dz.numel(), act, hip.stream()), "act_bwd")
            two = torch.full((nmember * stride_el,), float("nan"), device="cuda")
            hip.check(lib.cg_conv2d_wgrad_g(byref(geom), byref(grp), hip.ptr(x1d), hip.ptr(x2d), hip.ptr(dz), hip.ptr(two[:nw]),
                                            hip.ptr(two[nw:]), 0, hip.ptr(wsb), wsb.numel(), hip.stream()), "wgrad_g")
            one = torch.full((nmember * stride_el,), float("nan"), device="cuda")
            hip.check(lib.cg_conv2d_wgrad_act_g(byref(geom), byref(grp), hip.ptr(x1d), hip.ptr(x2d), hip.ptr(gyd), hip.ptr(yd), act,
                                                hip.ptr(one[:nw]), hip.ptr(one[nw:]), 0, hip.ptr(wsb), wsb.numel(), hip.stream()),
                      "wgrad_act_g")
            torch.cuda.synchronize()
            for m in range(nmember):
                a, b = one[m * stride_el:m * stride_el + nw + 64], two[m * stride_el:m * stride_el + nw + 64]
                assert torch.equal(a, b), (act, nmember, m, float((a - b).abs().max()))
    wide = ops.fwd_geom(2, 16, 16, 64, 0, 0, 3, 3, 1, 1, 64, 0)          # not a thin-input layer: refused, not mis-computed
    assert lib.cg_conv2d_wgrad_act_ok(byref(wide)) == 0
    xw, gw = cl(torch.randn(2, 64, 16, 16).cuda()), cl(torch.randn(2, 64, 16, 16).cuda())
    out = torch.empty(64 * 64 * 9 + 64, device="cuda")
    wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace_g(byref(wide), None))
    assert lib.cg_conv2d_wgrad_act_g(byref(wide), None, hip.ptr(xw), None, hip.ptr(gw), hip.ptr(gw), ops.ACT["relu"], hip.ptr(out),
                                     None, 0, hip.ptr(wsb), wsb.numel(), hip.stream()) != 0


# ------------------------------------------------------------------------------------------
# split-precision (fp16 x 3) kernels: forward, data gradient, weight gradient -- against fp64, with operands whose
# magnitude is far outside fp16's comfortable range (the device-side power-of-two scale must absorb it)
# ------------------------------------------------------------------------------------------
X3_CASES = [
    # name, N, H, W, Cin, Cout, K, stride, pad, up, x magnitude, dz magnitude
    ("3x3_64to128", 2, 16, 16, 64, 128, 3, 1, 1, 0, 1.0, 1.0),
    ("3x3_256to256_tiny_grad", 2, 32, 32, 256, 256, 3, 1, 1, 0, 1.0, 1e-7),
    ("4x4s2_64to128_tiny_x", 2, 32, 32, 64, 128, 4, 2, 1, 0, 3e-5, 1e-3),
    ("4x4s2_128to256_huge", 2, 32, 32, 128, 256, 4, 2, 1, 0, 4e3, 2e2),
    ("up_3x3_128to64", 2, 16, 16, 128, 64, 3, 1, 1, 1, 1.0, 1e-4),
    ("1x1_64to64", 2, 32, 32, 64, 64, 1, 1, 0, 0, 1.0, 1e-2),
    ("1x1_512to512_few_rows", 3, 8, 8, 512, 512, 1, 1, 0, 0, 10.0, 1e-5),
    # ragged: rows not a multiple of any tile height, channel counts not a multiple of any tile width
    ("odd_3x3_32to96", 1, 9, 7, 32, 96, 3, 1, 1, 0, 1.0, 1.0),
    ("odd_4x4s2_64to160", 3, 9, 7, 64, 160, 4, 2, 1, 0, 1.0, 1e-3),
    ("odd_up_3x3_96to64", 1, 5, 3, 96, 64, 3, 1, 1, 1, 2.0, 1.0),
    ("single_pixel_1x1_64to32", 1, 1, 1, 64, 32, 1, 1, 0, 0, 1.0, 1.0),
]


@pytest.mark.parametrize("case", X3_CASES, ids=[c[0] for c in X3_CASES])
def test_split_precision_conv_kernels(cga, case):
    from ctypes import byref
    from council_gan_amd import hip, ops
    _, N, H, W, Cin, Cout, K, stride, pad, up, xmag, gmag = case
    lib = hip.load()
    g = torch.Generator().manual_seed(sum(map(ord, case[0])))
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64) * xmag
    w = torch.randn(Cout, Cin, K, K, generator=g, dtype=torch.float64) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g, dtype=torch.float64) * xmag
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    inp = F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr
    yr = F.conv2d(F.pad(inp, (pad,) * 4), wr, br, stride=stride)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64) * gmag
    yr.backward(gy)

    xd, wd, bd, gyd = cl(dev(x)), cl(dev(w)), dev(b), cl(dev(gy))
    geom = ops.fwd_geom(N, H, W, Cin, 0, up, K, K, stride, pad, Cout, 0)
    with torch.no_grad():
        xs = ops.split_f16_dynamic(xd)                      # device-side scale for x
        ws = ops.split_f16(wd, hip.X3_WSCALE)
        scale = float(xs.state[1])
        peak = float(xd.abs().max()) * scale
        assert 4096.0 <= peak < 8192.0 and np.log2(scale) == np.round(np.log2(scale)), (scale, peak)
        y = ops.conv2d_x3(xs, ws, Cout, K, K, bd, stride, pad, "none", upsample=bool(up))
        dzs = ops.split_f16_dynamic(gyd)
        dx = ops.conv_dgrad_x3(geom, dzs, wd, 0, Cin)
        dw = torch.zeros_like(wd)
        db = torch.zeros(Cout, device="cuda")
        wgrad_ok = bool(lib.cg_conv2d_wgrad_x3_ok(byref(geom)))
        assert wgrad_ok or case[0].startswith(("odd_", "single_")), "every regular layer shape must qualify"
        if wgrad_ok:
            wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace(byref(geom)))
            hip.check(lib.cg_conv2d_wgrad_x3(byref(geom), xs.hi_ptr(), xs.lo, xs.scale_ptr(), dzs.hi_ptr(), dzs.lo, dzs.scale_ptr(),
                                             hip.ptr(dw), hip.ptr(db), 0, hip.ptr(wsb), wsb.numel(), hip.stream()), "wgrad_x3")
    errs = {"fwd": rel(y, yr), "dx": rel(dx, xr.grad)}
    if wgrad_ok:
        errs.update(dw=rel(dw, wr.grad), db=rel(db, br.grad))
    assert max(errs.values()) < 2e-5, errs          # fp32-class: the fp32-MFMA kernels sit at ~1e-6 on these shapes


@pytest.mark.parametrize("shape", [(4, 16, 16, 256, 256, 3, 1, 1), (6, 9, 16, 128, 512, 4, 2, 1), (3, 8, 8, 512, 256, 1, 1, 0)],
                         ids=["3x3_256to256", "4x4s2_128to512_ragged_rows", "1x1_512to256"])
def test_split_precision_weight_gradient_256x128_tile(cga, shape):
    """The 256 x 128 / 16-wave tile of conv_wgrad_x3t_kernel (cg_conv2d_wgrad_x3_bm256, CG_WGRAD_X3_BM256): against
    fp64 and against the 128 x 128 tile, with the bias gradient, for one member and for a member-batched launch."""
    _wgrad_tile_case(cga, shape, lambda lib, on: lib.cg_conv2d_wgrad_x3_bm256(1 if on else 0),
                     lambda lib, prev: lib.cg_conv2d_wgrad_x3_bm256(prev))


@pytest.mark.parametrize("shape", [(4, 16, 16, 256, 256, 3, 1, 1), (6, 9, 16, 256, 512, 4, 2, 1), (3, 8, 8, 512, 256, 1, 1, 0)],
                         ids=["3x3_256to256", "4x4s2_256to512_ragged_rows", "1x1_512to256"])
def test_split_precision_weight_gradient_wide_tile(cga, shape):
    """conv_wgrad_x3tw_kernel (256 x 256 LDS-DMA tile, cg_conv2d_wgrad_x3_wide): written after the round's last GPU call."""
    _wgrad_tile_case(cga, shape, lambda lib, on: lib.cg_conv2d_wgrad_x3_wide(1 if on else 0),
                     lambda lib, prev: lib.cg_conv2d_wgrad_x3_wide(prev))


@pytest.mark.parametrize("shape", [(4, 16, 16, 256, 256, 3, 1, 1), (6, 8, 16, 128, 128, 3, 1, 1), (6, 9, 16, 64, 128, 4, 2, 1),
                                   (3, 16, 16, 64, 64, 3, 1, 1)],
                         ids=["wide_3x3_256to256", "128x128_ragged_rows", "128x128_multitap_4x4s2", "64x128_3x3_64to64"])
def test_weight_gradient_xcd_grouped_block_order_is_bit_identical(cga, shape):
    """cg_tuning.wgrad_xcd_group (round 6): the split-precision weight-gradient grids walk their (tile, member, split) cells in an
    order that keeps the tap tiles of one position range on ONE XCD.  Every block computes the same cell either way: weight and
    bias gradients must be bit-identical with the grouping on and off, for one member and for a member-batched launch, on the
    LDS-DMA tile and on the transposing-read tiles, with row counts that are multiples of no tile edge."""
    from council_gan_amd import hip

    def switch(lib, on):
        prev = hip.tuning().wgrad_xcd_group
        t = hip.tuning()
        t.wgrad_xcd_group = 1 if on else 0
        import ctypes
        hip.check(lib.cg_tuning_set(ctypes.byref(t)), "cg_tuning_set")
        return prev

    def restore(lib, prev):
        switch(lib, bool(prev))
    _wgrad_tile_case(cga, shape, switch, restore, exact=True)


def _wgrad_tile_case(cga, shape, switch, restore, exact=False):
    from ctypes import byref
    from council_gan_amd import hip, ops
    N, H, W, Cin, Cout, K, stride, pad = shape
    lib = hip.load()
    g = torch.Generator().manual_seed(7 + Cout + K)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    gy = torch.randn(N, Cout, Ho, Wo, generator=g, dtype=torch.float64) * 1e-3
    xd, gyd = cl(dev(x)), cl(dev(gy))
    geom = ops.fwd_geom(N, H, W, Cin, 0, 0, K, K, stride, pad, Cout, 0)

    def ref(n0, n1):
        wz = torch.zeros(Cout, Cin, K, K, dtype=torch.float64, requires_grad=True)
        bz = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
        F.conv2d(F.pad(x[n0:n1], (pad,) * 4), wz, bz, stride=stride).backward(gy[n0:n1])
        return wz.grad, bz.grad

    def run(nmember):
        stride_el = Cout * Cin * K * K + Cout + 64          # members' gradients `stride_el` floats apart, as in a pool
        flat = torch.zeros(nmember * stride_el, device="cuda")
        grp = hip.Group(nmember, 0, stride_el)
        with torch.no_grad():
            xs, dzs = ops.split_f16_dynamic(xd), ops.split_f16_dynamic(gyd)
            assert lib.cg_conv2d_wgrad_x3_ok_g(byref(geom), byref(grp))
            wsb = hip.workspace(lib.cg_conv2d_wgrad_workspace_g(byref(geom), byref(grp)))
            dw, db = flat[:Cout * Cin * K * K], flat[Cout * Cin * K * K:]
            hip.check(lib.cg_conv2d_wgrad_x3_g(byref(geom), byref(grp), xs.hi_ptr(), xs.lo, xs.scale_ptr(), dzs.hi_ptr(), dzs.lo,
                                               dzs.scale_ptr(), hip.ptr(dw), hip.ptr(db), 0, hip.ptr(wsb), wsb.numel(),
                                               hip.stream()), "wgrad_x3_g")
        torch.cuda.synchronize()
        out = []
        for m in range(nmember):
            o = flat[m * stride_el:(m + 1) * stride_el]
            out.append((o[:Cout * Cin * K * K].view(Cout, K, K, Cin).permute(0, 3, 1, 2).cpu().double(),
                        o[Cout * Cin * K * K:Cout * Cin * K * K + Cout].cpu().double()))
        return out

    members = [1] + ([3] if N % 3 == 0 else [2])
    prev = switch(lib, False)
    try:
        # (a layer may qualify for the split-precision weight gradient only WITH the switch on: then there is no baseline)
        has_base = bool(lib.cg_conv2d_wgrad_x3_ok_g(byref(geom), byref(hip.Group(1, 0, 0))))
        base = {n: run(n) for n in members} if has_base else None
        switch(lib, True)
        wide = {n: run(n) for n in members}
    finally:
        restore(lib, prev)
    for n in members:
        per = N // n
        for m in range(n):
            rw, rb = ref(m * per, (m + 1) * per)
            for got in ((base[n][m], wide[n][m]) if has_base else (wide[n][m],)):
                assert rel(got[0], rw) < 2e-5 and rel(got[1], rb) < 2e-5, (n, m, rel(got[0], rw), rel(got[1], rb))
            if has_base:
                assert rel(wide[n][m][0], base[n][m][0]) < 2e-6
                if exact:
                    assert torch.equal(wide[n][m][0], base[n][m][0]) and torch.equal(wide[n][m][1], base[n][m][1])


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 5, 6, 7, 12, 13, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 50, 51, 52, 60, 64, 67, 68])
def test_split_precision_forward_every_tile_configuration(cga, cfg):
    """Every tile configuration of conv_fwd_x3_kernel the library ships, forced explicitly, on a shape whose row count
    (960) and channel count (160) are multiples of no tile edge, with bias + LeakyReLU in the epilogue."""
    from ctypes import byref
    from council_gan_amd import hip, ops
    lib = hip.load()
    N, H, W, Cin, Cout = 2, 24, 20, 64, 160
    g = torch.Generator().manual_seed(40 + cfg)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) / np.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    xd, wd, bd = cl(dev(x)), cl(dev(w)), dev(b)
    geom = ops.fwd_geom(N, H, W, Cin, 0, 0, 3, 3, 1, 1, Cout, ops.ACT["lrelu"])
    y = torch.full((N, Cout, H, W), float("nan"), device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        xs, ws = ops.split_f16(xd), ops.split_f16(wd, hip.X3_WSCALE)
        hip.check(lib.cg_conv2d_fwd_x3(byref(geom), xs.hi_ptr(), xs.lo, ws.hi_ptr(), ws.lo, float(ws.scale), None, hip.ptr(bd),
                                       hip.ptr(y), None, 0, None, 0, None, cfg, None, None, hip.stream()), "cg_conv2d_fwd_x3")
    assert rel(y, ref) < 2e-5, rel(y, ref)


def test_instnorm_apply_split_and_act_bwd_split(cga):
    """The split-emitting producers: AdaIN apply -> {hi, lo} planes, fused activation-backward -> scaled planes."""
    from council_gan_amd import hip, ops
    g = torch.Generator().manual_seed(11)
    x = cl(dev(torch.randn(2, 64, 16, 16, generator=g) * 3 + 1))
    res = cl(dev(torch.randn(2, 64, 16, 16, generator=g)))
    with torch.no_grad():
        y_ref = cga.ops.instance_norm(x, act="relu", residual=res)
        y, ys = ops.instnorm_split(x, None, 0, 0, act="relu", residual=res, want_f32=True)
        assert torch.equal(y, y_ref)
        n = y.numel()
        phys = y.permute(0, 2, 3, 1).reshape(-1)
        recon = ys.to_float()
        assert float((recon - phys).abs().max()) <= 2.0 ** -21 * float(phys.abs().max())
        dy = cl(dev(torch.randn(2, 64, 16, 16, generator=g) * 1e-6))
        yact = cl(dev(torch.randn(2, 64, 16, 16, generator=g)))
        dz, dzs = ops.act_bwd_split(dy, yact, hip.ACT["lrelu"], True)
        ref = dy * torch.where(yact > 0, torch.ones_like(yact), torch.full_like(yact, 0.2))
        assert torch.allclose(dz, ref, rtol=1e-6, atol=0)
        scale = float(dzs.state[1])
        recon = dzs.to_float() / scale
        assert float((recon - ref.permute(0, 2, 3, 1).reshape(-1)).abs().max()) <= 2.0 ** -20 * float(ref.abs().max())


def test_device_input_pipeline_matches_reference_transforms(cga):
    """crop window + (deferred) horizontal flip + ToTensor + Normalize on the device, bit-exact against
    the restated torchvision definitions (oracle/input_oracle.py); the trainer takes the result without a copy."""
    from oracle import input_oracle as IO
    rng = np.random.RandomState(5)
    N, Hs, Ws, H, W = 5, 71, 83, 64, 64
    imgs = rng.randint(0, 256, size=(N, Hs, Ws, 3)).astype(np.uint8)
    imgs[0, :2, :2] = [[[0, 255, 1], [254, 127, 128]], [[3, 85, 170], [17, 34, 51]]]
    tops, lefts = rng.randint(0, Hs - H + 1, size=N), rng.randint(0, Ws - W + 1, size=N)
    flips = np.array([0, 1, 1, 0, 1], dtype=bool)
    pipe = cga.DeviceInput('cuda:0', H, W)
    # the reference flips the whole image first and then crops at (top, left): same pixels as cropping the original
    # at the mirrored window and flipping the crop
    crop = np.stack([tops, [IO.window_after_flip(l, Ws, W) if f else l for l, f in zip(lefts, flips)]], 1)
    got = pipe(torch.from_numpy(imgs), crop_tl=crop, flip=flips)
    assert got.shape == (N, 3, H, W) and got.is_contiguous(memory_format=torch.channels_last)
    want = torch.stack([IO.sample(imgs[n], int(tops[n]), int(lefts[n]), H, W, flip_first=bool(flips[n])) for n in range(N)])
    assert torch.equal(got.cpu(), want)
    # Council_Trainer._img: `.to(device, float32).contiguous(channels_last)` is the identity on this tensor
    assert got.to(got.device, dtype=torch.float32).contiguous(memory_format=torch.channels_last) is got
    # no crop / no flip, full-size window; a second call reuses the staging ring
    full = cga.DeviceInput('cuda:0', Hs, Ws)
    for _ in range(6):
        g2 = full(imgs)
    assert torch.equal(g2.cpu(), torch.stack([IO.to_tensor_normalize(imgs[n]) for n in range(N)]))
    with pytest.raises(ValueError):
        pipe(imgs, crop_tl=np.array([[Hs - H + 1, 0]] * N))
    with pytest.raises(ValueError):
        cga.DeviceInput('cuda:0', 100, 64)(imgs)


def test_device_input_pipeline_matches_the_pil_fixture(cga):
    """The device input tail against ground truth from an independent implementation: tests/golden/input_pil.npz (flip and
    crop by PIL, ToTensor / Normalize by NumPy; oracle/make_input_golden.py), bit for bit."""
    from oracle import input_oracle as IO
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "input_pil.npz"))
    H, W = int(z["height"]), int(z["width"])
    Ws = z["images"].shape[2]
    pipe = cga.DeviceInput('cuda:0', H, W)
    # RandomCrop draws its window in the FLIPPED image; the kernel crops the original at the mirrored window, then flips
    crop = np.stack([z["tops"], [IO.window_after_flip(int(l), Ws, W) if f else int(l) for l, f in zip(z["lefts"], z["flips"])]], 1)
    got = pipe(torch.from_numpy(z["images"]), crop_tl=crop, flip=z["flips"])
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert np.array_equal(got.cpu().numpy(), z["out"])


@pytest.mark.parametrize("kind", ["fp32_first_layer", "x3", "x3_too_many_blocks"])
def test_conv_epilogue_reports_output_maximum(cga, kind):
    """The per-block max|y| a forward convolution can leave behind for the consumer's dynamic split
    (cg_conv2d_fwd_amax / the amax arguments of cg_conv2d_fwd_x3): their maximum is exactly max|y|, and a split that
    uses them produces the same scale and halves as one that measures the tensor itself."""
    import ctypes
    from ctypes import byref
    from council_gan_amd import hip, ops
    lib = hip.load()
    torch.manual_seed(9)
    if kind == "fp32_first_layer":
        N, H, W, Cin, Cout, K, stride, pad = 2, 64, 64, 3, 64, 4, 2, 1
    elif kind == "x3":
        N, H, W, Cin, Cout, K, stride, pad = 2, 32, 32, 64, 128, 4, 2, 1
    else:
        N, H, W, Cin, Cout, K, stride, pad = 8, 256, 256, 32, 32, 3, 1, 1       # 8192 blocks of 64 rows: shared slots
    x = cl(torch.randn(N, Cin, H, W).cuda() * 3)
    w = cl((torch.randn(Cout, Cin, K, K) / np.sqrt(Cin * K * K)).cuda())
    b = torch.randn(Cout).cuda()
    g = ops.fwd_geom(N, H, W, Cin, 0, 0, K, K, stride, pad, Cout, ops.ACT["lrelu"])
    y = torch.empty((N, Cout, g.Ho, g.Wo), device="cuda").contiguous(memory_format=torch.channels_last)
    state = torch.full((hip.SPLIT_STATE_FLOATS,), -1.0, device="cuda")
    nslots = ctypes.c_int(-1)
    with torch.no_grad():
        if kind == "fp32_first_layer":
            hip.check(lib.cg_conv2d_fwd_amax(byref(g), hip.ptr(x), None, hip.ptr(w), hip.ptr(b), hip.ptr(y), hip.ptr(state),
                                             byref(nslots), hip.stream()), "cg_conv2d_fwd_amax")
        else:
            xs, ws = ops.split_f16_dynamic(x), ops.split_f16(w, hip.X3_WSCALE)
            hip.check(lib.cg_conv2d_fwd_x3(byref(g), xs.hi_ptr(), xs.lo, ws.hi_ptr(), ws.lo, float(ws.scale), xs.scale_ptr(),
                                           hip.ptr(b), hip.ptr(y), None, 0, None, 0, None, -1, hip.ptr(state), byref(nslots),
                                           hip.stream()), "cg_conv2d_fwd_x3")
        assert nslots.value == (1024 if kind == "x3_too_many_blocks" else nslots.value) and 0 < nslots.value <= 1024
        assert float(state[2:2 + nslots.value].max()) == float(y.abs().max())
        assert float(state[2:2 + nslots.value].min()) >= 0.0
        a = ops.split_f16_dynamic(y, (state, nslots.value))
        r = ops.split_f16_dynamic(y)
        assert float(a.state[1]) == float(r.state[1]) and torch.equal(a.buf, r.buf)


GROUP_CASES = [
    # name, Cin, C2, Cout, K, stride, pad, up, H, act, split datapath, exact (bit-identical expected)
    ("first_dis_4x4s2_3to64", 3, 0, 64, 4, 2, 1, 0, 32, "lrelu", True, True),
    ("first_disc_3x3_3+3to64", 3, 3, 64, 3, 1, 1, 0, 32, "lrelu", True, True),
    ("fp32_pipe_3x3_64to128", 64, 0, 128, 3, 1, 1, 0, 16, "relu", False, True),
    ("fp32_pipe_4x4s2_64to128", 64, 0, 128, 4, 2, 1, 0, 16, "lrelu", False, True),
    ("x3_3x3_64to128", 64, 0, 128, 3, 1, 1, 0, 16, "relu", True, False),
    ("x3_4x4s2_128to256", 128, 0, 256, 4, 2, 1, 0, 16, "lrelu", True, False),
    ("x3_up_3x3_128to64", 128, 0, 64, 3, 1, 1, 1, 8, "none", True, False),
    ("x3_1x1_256to256", 256, 0, 256, 1, 1, 0, 0, 8, "none", True, False),
    ("head_1x1_256to1", 256, 0, 1, 1, 1, 0, 0, 8, "none", True, False),
    ("head_1x1_64to12_tanh", 64, 0, 12, 1, 1, 0, 0, 16, "tanh", True, False),
]


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("case", GROUP_CASES, ids=[c[0] for c in GROUP_CASES])
def test_member_batched_conv_matches_single_member_launches(cga, case, n):
    """The same layer of n council members as ONE launch (ops.members(n): batched activations, parameters at a uniform
    stride in an optim.ParamPool) against n single-member launches: forward, data gradient, weight and bias gradient.
    fp32 datapath and first layers: bit for bit.  Split-precision layers: the batched tensor gets ONE power-of-two scale
    instead of one per member, which may move a lo half in or out of fp16's subnormal range -- 2e-6."""
    from council_gan_amd import ops
    from council_gan_amd.optim import ParamPool
    _, Cin, C2, Cout, K, stride, pad, up, H, act, split, exact = case
    B = 2

    def build():
        torch.manual_seed(sum(map(ord, case[0])))
        convs = [torch.nn.Conv2d(Cin + C2, Cout, K, stride, bias=True) for _ in range(n)]
        for k, c in enumerate(convs):
            with torch.no_grad():
                c.bias.normal_()
                c.weight.mul_(1.0 + 0.25 * k)
        opts = [cga.FlatAdam(list(c.parameters()), lr=1e-4) for c in convs]
        pool = ParamPool(opts)
        pool.materialize('cuda')
        return convs, pool, (ops.SplitWeights(pool) if split else None)

    torch.manual_seed(7)
    x = cl(torch.randn(n * B, Cin, H, H).cuda())
    x2 = cl(torch.randn(n * B, C2, H, H).cuda()) if C2 else None
    saved = (ops.X3_FORWARD, ops.X3_BACKWARD, ops.X3_DYNAMIC_INPUT)
    ops.X3_FORWARD = ops.X3_BACKWARD = ops.X3_DYNAMIC_INPUT = split
    try:
        def run(convs, pool, mgr, scope, rows):
            xi = x[rows].clone().requires_grad_(True)
            x2i = x2[rows].clone().requires_grad_(True) if C2 else None
            with ops.members(scope):
                y = ops.conv2d(xi, convs[0].weight, convs[0].bias, stride, pad, act, x2=x2i, upsample=bool(up), wmgr=mgr)
            return xi, x2i, y

        convs_g, pool_g, mgr_g = build()
        xi, x2i, y_g = run(convs_g, pool_g, mgr_g, n, slice(0, n * B))
        torch.manual_seed(8)
        gy = cl(torch.randn(y_g.shape).cuda())
        y_g.backward(gy)
        dx_g = xi.grad
        dx2_g = x2i.grad if C2 else None

        convs_s, pool_s, mgr_s = build()
        ys, dxs, dx2s = [], [], []
        for m in range(n):
            rows = slice(m * B, (m + 1) * B)
            xi, x2i, y = run(convs_s[m:], pool_s, mgr_s, 1, rows)
            y.backward(gy[rows])
            ys.append(y.detach()); dxs.append(xi.grad)
            if C2:
                dx2s.append(x2i.grad)
    finally:
        ops.X3_FORWARD, ops.X3_BACKWARD, ops.X3_DYNAMIC_INPUT = saved

    def close(a, b, what):
        if exact:
            assert torch.equal(a, b), what
        else:
            e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            assert e < 2e-6, (what, e)
    ops.wgrad_join()              # weight gradients run on a companion stream of the backward's stream
    close(y_g.detach(), torch.cat(ys), "forward")
    close(dx_g, torch.cat(dxs), "data gradient")
    if C2:
        close(dx2_g, torch.cat(dx2s), "data gradient of the second source")
    close(pool_g.grad, pool_s.grad, "weight / bias gradients (whole pool)")
    assert float(pool_g.grad.abs().max()) > 0


def test_member_batched_losses_and_adam(cga):
    """Per-member LSGAN / focus criteria, loss matching and the pooled Adam step under ops.members(n) against the
    single-member calls, bit for bit."""
    from council_gan_amd import hip, ops
    from council_gan_amd.optim import ParamPool
    lib = hip.load()
    n, B = 3, 2
    torch.manual_seed(3)
    outs = [torch.randn(n * 2 * B, 1, 8, 8).cuda().requires_grad_(True), torch.randn(n * 2 * B, 1, 4, 4).cuda().requires_grad_(True)]
    tgt = torch.tensor(([0.0] * B + [1.0] * B) * n).cuda()
    wt = torch.tensor([0.5 + 0.1 * i for i in range(n * 2 * B)]).cuda()
    up = torch.tensor([1.0, 2.0, 0.5]).cuda()
    with ops.members(n):
        l = ops.lsgan_loss(outs, tgt, wt, B)
    assert tuple(l.shape) == (n,)
    torch.autograd.backward([l], [up])
    for m in range(n):
        rows = slice(m * 2 * B, (m + 1) * 2 * B)
        o1 = [o.detach()[rows].clone().requires_grad_(True) for o in outs]
        l1 = ops.lsgan_loss(o1, tgt[rows].clone(), wt[rows].clone(), B)
        assert float(l1) == float(l[m]), m
        (l1 * up[m]).backward()
        for a, b in zip(o1, outs):
            assert torch.equal(a.grad, b.grad[rows]), m
    mask = cl(torch.rand(n * B, 3, 16, 16).cuda()).requires_grad_(True)
    with ops.members(n):
        ft, parts = ops.focus_loss(mask, 0.5, 0.01, 1.0, 3.0, 0.5, False, True)
    torch.autograd.backward([ft], [up])
    for m in range(n):
        rows = slice(m * B, (m + 1) * B)
        mk = mask.detach()[rows].clone().requires_grad_(True)
        f1, p1 = ops.focus_loss(mk, 0.5, 0.01, 1.0, 3.0, 0.5, False, True)
        assert float(f1) == float(ft[m]) and torch.equal(p1, parts[m]), m
        (f1 * up[m]).backward()
        assert torch.equal(mk.grad, mask.grad[rows]), m
    # pooled Adam: one launch for every member == per-member steps
    def build():
        torch.manual_seed(5)
        ps = [[torch.nn.Parameter(torch.randn(40, 8, 3, 3)), torch.nn.Parameter(torch.randn(40))] for _ in range(n)]
        pool = ParamPool([cga.FlatAdam(p, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4) for p in ps])
        pool.materialize('cuda')
        torch.manual_seed(6)
        pool.grad.copy_(torch.randn(pool.grad.shape).cuda())
        return ps, pool
    ps_a, pool_a = build()
    ps_b, pool_b = build()
    for p in ps_a[0]:
        p._cg_grad._cg_touched = True           # member-batched launches flag the lead member only
    for mem in ps_b:
        for p in mem:
            p._cg_grad._cg_touched = True
    for _ in range(2):
        pool_a.step(0, n, lockstep=True)
        for o in pool_b.opts:
            o.step()
    assert torch.equal(pool_a.data, pool_b.data) and torch.equal(pool_a.m, pool_b.m) and torch.equal(pool_a.v, pool_b.v)
    assert [o._steps for o in pool_a.opts] == [o._steps for o in pool_b.opts] == [[2, 2]] * n
    # rings / loss matching / objective assembly
    ring_g, ring_c = torch.ones(n * 5).cuda(), torch.ones(n * 5).cuda()
    adv, lc, w = torch.tensor([1.0, 2.0, 3.0]).cuda(), torch.tensor([4.0, 5.0, 6.0]).cuda(), torch.zeros(n).cuda()
    hip.check(lib.cg_ring_push_g(hip.ptr(ring_g), 5, 7, hip.ptr(adv), n, hip.stream()), "ring")
    hip.check(lib.cg_loss_match_g(hip.ptr(ring_g), hip.ptr(ring_c), 5, 7, hip.ptr(lc), hip.ptr(w), n, hip.stream()), "match")
    want_w = [((4 + a) / 5) / ((4 + c) / 5) for a, c in zip((1.0, 2.0, 3.0), (4.0, 5.0, 6.0))]
    assert np.allclose(w.cpu().numpy(), want_w, rtol=1e-6)
    assert float(ring_g[5 + 7 % 5]) == 2.0 and float(ring_c[10 + 7 % 5]) == 6.0
    total, council, gc = ops.gen_total(ft.detach(), adv, lc, w, 1.0, 6.0, n)
    assert np.allclose(council.cpu().numpy(), [6.0 * a * b for a, b in zip(want_w, (4.0, 5.0, 6.0))], rtol=1e-6)
    assert np.allclose(total.cpu().numpy(), (ft.detach() + adv + council).cpu().numpy(), rtol=1e-6)
    assert np.allclose(gc.cpu().numpy(), [6.0 * a for a in want_w], rtol=1e-6)
    t = ops.take_rows(mask.detach(), outs[0].detach().view(n * 2 * B, 64)[:, :0] if False else None, [3, 0, 5])
    assert torch.equal(t, mask.detach()[[3, 0, 5]])


@pytest.mark.parametrize("members", [1, 2])
def test_activation_backward_takes_its_scale_from_the_data_gradient_kernel(cga, members, monkeypatch):
    """A stack of stride-2 LeakyReLU convolutions (the discriminators' shape, networks.py:38-46): the data-gradient launch of
    layer L+1 -- four output-parity classes in one launch -- leaves the per-block maxima of dx behind, and layer L's fused
    activation-backward + split takes its power-of-two scale from them instead of measuring dz in a pass of its own.
    Gradients against fp64, with the hand-off on and off, single member and member-batched."""
    from council_gan_amd import ops, optim
    torch.manual_seed(3)
    B = 2
    convs = [[torch.nn.Conv2d(ci, co, 4, 2) for ci, co in ((32, 64), (64, 128), (128, 128))] for _ in range(members)]
    opts = [cga.FlatAdam([p for c in cs for p in c.parameters()], lr=1e-4) for cs in convs]
    pool = optim.ParamPool(opts)
    pool.materialize('cuda')
    mgr = ops.SplitWeights(pool)
    pool.split = mgr
    x = torch.randn(members * B, 32, 32, 32, dtype=torch.float64)
    gy = torch.randn(members * B, 128, 4, 4, dtype=torch.float64)

    def ref(m):
        xr = x[m * B:(m + 1) * B].clone().requires_grad_(True)
        h = xr
        ws = []
        for c in convs[m]:
            w, b = c.weight.detach().double().cpu().requires_grad_(True), c.bias.detach().double().cpu().requires_grad_(True)
            ws += [w, b]
            h = F.leaky_relu(F.conv2d(F.pad(h, (1, 1, 1, 1)), w, b, stride=2), 0.2)
        h.backward(gy[m * B:(m + 1) * B])
        return xr.grad, [t.grad for t in ws]

    def run():
        pool.zero_grad()
        xd = cl(dev(x)).requires_grad_(True)
        with ops.members(members):
            h = xd
            for c in convs[0]:
                h = ops.conv2d(h, c.weight, c.bias, 2, 1, 'lrelu', wmgr=mgr)
            h.backward(cl(dev(gy)))
        torch.cuda.synchronize()
        return xd.grad.double().cpu(), [[p._cg_grad.double().cpu().clone() for c in convs[m] for p in c.parameters()]
                                         for m in range(members)]

    used = []
    orig = ops.act_bwd_split
    monkeypatch.setattr(ops, "act_bwd_split", lambda dy, y, act, want, amax=None: (used.append(amax is not None), orig(dy, y, act, want, amax))[1])
    got = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "ACT_BWD_AMAX", on)
        used.clear()
        got[on] = run()
        if on:
            assert used.count(True) >= 2, used      # the two lower layers received the maxima of their dy
    for m in range(members):
        rx, rw = ref(m)
        for on in (True, False):
            gx, gw = got[on]
            assert rel(gx[m * B:(m + 1) * B], rx) < 2e-5, (on, m, rel(gx[m * B:(m + 1) * B], rx))
            for a, b in zip(gw[m], rw):
                assert rel(a, b) < 2e-5, (on, m)


@pytest.mark.parametrize("case", [(2, 64, 16, 16, 'in', False), (4, 256, 32, 32, 'in', False), (16, 256, 64, 64, 'in', False),
                                  (2, 96, 12, 20, 'in', False), (2, 64, 16, 16, 'in', True), (16, 256, 64, 64, 'in', True),
                                  (2, 64, 16, 16, 'none', False)])
def test_resblock_skip_gradient_joins_the_data_gradient_epilogue(cga, case, monkeypatch):
    """networks.py:448-461 (`out += residual`): the block input's two gradients -- through the first convolution and through the
    skip edge -- are added in the data-gradient kernel's epilogue (ops.SkipLink, cg_x3_epilogue.addend) instead of by the autograd
    engine.  Two stacked blocks, so the upper block's fused sum is the lower block's incoming gradient.  The sum is formed after
    the value is rounded to fp32: every gradient BIT-IDENTICAL to the engine's addition (CG_SKIP_FUSE=0) -- on the wide LDS-DMA
    tile (the benchmark's member-batched launch: 256 channels at 64x64, batch 16), the register-staged tiles, ragged rows, the
    generic epilogue copies, and (norm 'none': no hand-over point) with the link declined.  Small cases also against fp64."""
    from council_gan_amd import ops, optim, networks, hip
    N, C, H, W, norm, generic = case
    torch.manual_seed(5)
    # the benchmark-size cases keep the model's ReLU; the cases that are also compared with fp64 take a smooth activation (a ReLU
    # whose pre-activation sits within round-off of zero flips, and the comparison then measures that lottery, not the kernels)
    activation = 'relu' if N * H * W >= 65536 else 'tanh'
    blocks = [networks.ResBlock(C, norm=norm, activation=activation, pad_type='zero') for _ in range(2)]
    opt = cga.FlatAdam([p for b in blocks for p in b.parameters()], lr=1e-4)
    pool = optim.ParamPool([opt])
    pool.materialize('cuda')
    mgr = ops.SplitWeights(pool)
    pool.split = mgr
    for b in blocks:
        for m in b.model:
            m._cg_wmgr = mgr
    x = torch.randn(N, C, H, W, dtype=torch.float64)
    gy = torch.randn(N, C, H, W, dtype=torch.float64)
    links = []
    orig = ops.skip_link
    monkeypatch.setattr(ops, "skip_link", lambda t: (links.append(orig(t)), links[-1])[1])
    fused = []
    orig_dgrad = ops.conv_dgrad_x3
    monkeypatch.setattr(ops, "conv_dgrad_x3", lambda *a, **k: (fused.append(k.get("addend") is not None), orig_dgrad(*a, **k))[1])

    def run(on):
        monkeypatch.setattr(ops, "SKIP_FUSE", on)
        links.clear()
        fused.clear()
        pool.zero_grad()
        xd = cl(dev(x)).requires_grad_(True)
        h = xd
        for b in blocks:
            h = b(h)
        h.backward(cl(dev(gy)))
        torch.cuda.synchronize()
        return xd.grad.clone(), [p._cg_grad.clone() for b in blocks for p in b.parameters()], h.detach().clone()

    with hip.tuned(x3_generic_epilogue=1 if generic else 0):
        gx1, gw1, y1 = run(True)
        n_links = sum(l is not None for l in links)
        n_fused = fused.count(True)
        gx0, gw0, y0 = run(False)
    if norm == 'none':
        assert n_links == 0 and n_fused == 0      # no norm backward to hand the gradient over: the engine adds it
    else:
        # the lower block's input is a leaf that wants its gradient, the upper block's is the lower block's output
        assert n_links == 2 and all(l.grad is None for l in links if l is not None)
        assert n_fused == 2 or C == 96, fused       # (96 channels: whichever kernel the width gets -- the result is what counts)
    assert torch.equal(y1, y0)
    assert torch.equal(gx1, gx0), float((gx1 - gx0).abs().max())
    for a, b in zip(gw1, gw0):
        assert torch.equal(a, b)
    if N * H * W >= 65536:
        return
    # fp64 reference of the same two blocks
    xr = x.clone().requires_grad_(True)
    h = xr
    for b in blocks:
        r = h
        for i, m in enumerate(b.model):
            w, bb = m.conv.weight.detach().double().cpu(), m.conv.bias.detach().double().cpu()
            h = F.conv2d(F.pad(h, (1, 1, 1, 1)), w, bb)
            if norm == 'in':
                h = F.instance_norm(h, eps=1e-5)
            if i == 0:
                h = torch.tanh(h)
        h = h + r
    h.backward(gy)
    assert rel(y1.double().cpu(), h.detach()) < 2e-5
    assert rel(gx1.double().cpu(), xr.grad) < 5e-5, rel(gx1.double().cpu(), xr.grad)


@pytest.mark.parametrize("members", [1, 2])
def test_fused_decoder_head_matches_layer_by_layer(cga, members):
    """cg_decoder_head_fwd_x3: the decoder's three 1x1 convolutions + mask / blend head (networks.py:393-407) as one kernel,
    against fp64 (image and mask <= 2e-5 of their range) and against the layer-by-layer split-precision path it replaces in
    the tape-free passes; a ragged pixel count (not a multiple of the 32-pixel tile), one member and two."""
    from council_gan_amd import ops, optim
    torch.manual_seed(5)
    B, H, W = 2, 17, 13
    nets = []
    for _ in range(members):
        dec = cga.Decoder_V2_atten(2, 1, 256, 3, res_norm='adain', activ='relu', pad_type='zero', num_of_mask_dim_to_add=3)
        for k in (7, 8, 9):
            torch.nn.init.kaiming_normal_(dec.model[k].conv.weight)
            torch.nn.init.normal_(dec.model[k].conv.bias, 0, 0.1)
        nets.append(dec)
    opts = [cga.FlatAdam(list(d.parameters()), lr=1e-4) for d in nets]
    pool = optim.ParamPool(opts)
    pool.materialize('cuda')
    mgr = ops.SplitWeights(pool)
    x = torch.randn(members * B, 64, H, W, dtype=torch.float64)
    im_in = torch.rand(members * B, 3, H, W, dtype=torch.float64) * 2 - 1
    xd, imd = cl(dev(x)), cl(dev(im_in))
    convs = [nets[0].model[k].conv for k in (7, 8, 9)]
    with torch.no_grad(), ops.members(members):
        xs = ops.split_f16(xd)
        got_im, got_mask = ops.decoder_head_x3(xs, convs, mgr, imd, 3, 3)
        # the path it replaces: three split-precision 1x1 convolutions, then cg_mask_blend_fwd
        y = xd
        for k, act in ((7, 'relu'), (8, 'relu'), (9, 'tanh')):
            c = nets[0].model[k].conv
            y = ops.conv2d(y, c.weight, c.bias, 1, 0, act, wmgr=mgr)
        ref_im, ref_mask = ops.mask_blend(y, imd, 3, 3)
    torch.cuda.synchronize()
    for m in range(members):
        h = x[m * B:(m + 1) * B]
        for k, act in ((7, 'relu'), (8, 'relu'), (9, 'tanh')):
            c = nets[m].model[k].conv
            h = F.conv2d(h, c.weight.detach().double().cpu(), c.bias.detach().double().cpu())
            h = torch.relu(h) if act == 'relu' else torch.tanh(h)
        mask = (torch.tanh(10 * h[:, 9:12]) + 1) / 2
        im = im_in[m * B:(m + 1) * B]
        for j in range(3):
            im = (1 - mask[:, j:j + 1]) * im + mask[:, j:j + 1] * h[:, 3 * j:3 * j + 3]
        sl = slice(m * B, (m + 1) * B)
        assert rel(got_im[sl], im) < 2e-5 and rel(got_mask[sl], mask) < 2e-5, (m, rel(got_im[sl], im), rel(got_mask[sl], mask))
    assert rel(got_im, ref_im.double().cpu()) < 2e-5 and rel(got_mask, ref_mask.double().cpu()) < 2e-5


@pytest.mark.parametrize("shape", [(4, 32, 32, 64, 128, 4, 2, 1), (4, 16, 8, 64, 64, 3, 1, 1), (2, 16, 16, 32, 128, 3, 1, 1),
                                   (4, 16, 16, 64, 128, 1, 1, 0)],
                         ids=["4x4s2_64to128", "3x3_64to64_K576_tail", "3x3_32to128", "1x1_64to128_K64_not_taken"])
def test_split_precision_weight_gradient_multi_tap_tiles(cga, shape):
    """conv_wgrad_x3t_kernel with K-tiles that span several taps (cg_tuning.wgrad_x3_multitap): layers with 32 / 64 input
    channels on 128-wide tiles -- every 32-channel group of a tile carries its own tap, columns past K = T * Cin read zeros.
    Against fp64 and against the one-tap-per-tile plan, weight + bias gradient, one member and member-batched."""
    from council_gan_amd import hip

    def switch(lib, on):
        t = hip.tuning()
        prev = t.wgrad_x3_multitap
        t.wgrad_x3_multitap = 1 if on else 0
        hip.check(lib.cg_tuning_set(t), "cg_tuning_set")
        return prev

    def restore(lib, prev):
        t = hip.tuning()
        t.wgrad_x3_multitap = prev
        hip.check(lib.cg_tuning_set(t), "cg_tuning_set")
    _wgrad_tile_case(cga, shape, switch, restore)


@pytest.mark.parametrize("n", [1, 2])
@pytest.mark.parametrize("shape", [(128, 64, 16), (256, 128, 16), (64, 32, 32)], ids=["128to64", "256to128", "64to32"])
def test_upsample_conv_as_transposed_conv(cga, shape, n):
    """nn.Upsample(2x nearest) + ZeroPad2d(1) + Conv2d(3x3) (networks.py:385-386, 513-516) on the summed-tap form
    (cg_upconv_*: four output-parity classes of a 4x4 stride-2 transposed convolution; data gradient = a 4x4 stride-2
    convolution over dz; weight gradient folded back onto the nine taps) against fp64 torch on the upsampled tensor:
    forward, instance-norm partial sums from the epilogue, data / weight / bias gradients, n members per launch."""
    from council_gan_amd import ops
    from council_gan_amd.optim import ParamPool
    Cin, Cout, H = shape
    B = 2
    torch.manual_seed(11 + Cin + n)
    convs = [torch.nn.Conv2d(Cin, Cout, 3, 1, bias=True) for _ in range(n)]
    for k, c in enumerate(convs):
        with torch.no_grad():
            c.bias.normal_()
            c.weight.mul_(1.0 + 0.5 * k)
    w64 = [c.weight.detach().double().clone() for c in convs]
    b64 = [c.bias.detach().double().clone() for c in convs]
    pool = ParamPool([cga.FlatAdam(list(c.parameters()), lr=1e-4) for c in convs])
    pool.materialize('cuda')
    mgr = ops.SplitWeights(pool)
    x = cl(torch.randn(n * B, Cin, H, H).cuda())
    gy = cl(torch.randn(n * B, Cout, 2 * H, 2 * H).cuda())
    saved = (ops.X3_FORWARD, ops.X3_BACKWARD, ops.X3_DYNAMIC_INPUT)
    ops.X3_FORWARD = ops.X3_BACKWARD = ops.X3_DYNAMIC_INPUT = True
    try:
        xi = x.clone().requires_grad_(True)
        stats = []
        with ops.members(n):
            taken = ops._upconv_ok(tuple(x.shape), convs[0].weight, 1, 1, "none", None, mgr)
            assert taken == (Cout >= 64), "the shipped widths (256 -> 128, 128 -> 64) must take the summed-tap path"
            y = ops.conv2d(xi, convs[0].weight, convs[0].bias, 1, 1, "none", upsample=True, stats=stats, wmgr=mgr)
            assert stats, "the epilogue should have produced the instance-norm partial sums"
            yn = ops.instance_norm(y, act="relu", stats=stats)
        y.backward(gy, retain_graph=True)
        torch.cuda.synchronize()
    finally:
        ops.X3_FORWARD, ops.X3_BACKWARD, ops.X3_DYNAMIC_INPUT = saved
    for m in range(n):
        rows = slice(m * B, (m + 1) * B)
        xr = x[rows].detach().double().cpu().requires_grad_(True)
        wr, br = w64[m].clone().requires_grad_(True), b64[m].clone().requires_grad_(True)
        yr = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wr, br, padding=1)
        yr.backward(gy[rows].double().cpu())
        assert rel(y[rows], yr) < TOL, ("forward", m)
        ynr = F.relu(F.instance_norm(yr.detach()))
        assert rel(yn[rows], ynr) < 1e-4, ("instance norm on the epilogue's partial sums", m)
        assert rel(xi.grad[rows], xr.grad) < TOL, ("data gradient", m)
        assert rel(convs[m].weight._cg_grad, wr.grad) < TOL, ("weight gradient", m)
        assert rel(convs[m].bias._cg_grad, br.grad) < TOL, ("bias gradient", m)


@pytest.mark.parametrize("n", [1, 2, 4])
@pytest.mark.parametrize("C", [64, 512, 96])
def test_composed_1x1_tail(cga, C, n):
    """MsImageDisCouncil's last two layers, Conv2d(dim, dim, 1) -> Conv2d(dim, 1, 1) with nothing in between
    (networks.py:142-143), evaluated as ONE dim -> 1 convolution on w_eff = W2 W1, b_eff = W2 b1 + b2 (cg_compose1x1_fwd / _bwd)
    against fp64 torch running the two layers: output, data gradient, and the gradients of all four parameters; n members per
    launch (pool-strided parameters)."""
    from council_gan_amd import ops
    from council_gan_amd.optim import ParamPool
    B, H = 2, 8
    torch.manual_seed(3 + C + n)
    nets = [(torch.nn.Conv2d(C, C, 1), torch.nn.Conv2d(C, 1, 1)) for _ in range(n)]
    for m, (c1, c2) in enumerate(nets):
        with torch.no_grad():
            c1.weight.mul_(1.0 + 0.4 * m)
            c1.bias.normal_(0, 0.3)
            c2.bias.normal_(0, 0.3)
    ref = [[p.detach().double().clone() for p in (c1.weight, c1.bias, c2.weight, c2.bias)] for c1, c2 in nets]
    pool = ParamPool([cga.FlatAdam([p for c in net for p in c.parameters()], lr=1e-4) for net in nets])
    pool.materialize('cuda')
    assert ops.composed_tail_ok(*nets[0])
    y = cl(torch.randn(n * B, C, H, H).cuda())
    g = cl(torch.randn(n * B, 1, H, H).cuda())
    yi = y.clone().requires_grad_(True)
    with ops.members(n):
        out = ops.composed_tail(yi, *nets[0])
    out.backward(g)
    torch.cuda.synchronize()
    for m in range(n):
        rows = slice(m * B, (m + 1) * B)
        yr = y[rows].detach().double().cpu().requires_grad_(True)
        W1, b1, W2, b2 = [t.clone().requires_grad_(True) for t in ref[m]]
        o = F.conv2d(F.conv2d(yr, W1, b1), W2, b2)
        o.backward(g[rows].double().cpu())
        assert rel(out[rows], o) < TOL, ("forward", m)
        assert rel(yi.grad[rows], yr.grad) < TOL, ("data gradient", m)
        c1, c2 = nets[m]
        for name, got, want in (("dW1", c1.weight._cg_grad, W1.grad), ("db1", c1.bias._cg_grad, b1.grad),
                                ("dW2", c2.weight._cg_grad, W2.grad), ("db2", c2.bias._cg_grad, b2.grad)):
            assert rel(got, want) < TOL, (name, m)


@pytest.mark.parametrize("n", [1, 2])
@pytest.mark.parametrize("first", ["thin3", "x3"])
def test_bounded_split_chain_and_fused_activation_backward(cga, n, first, monkeypatch):
    """A discriminator-like chain conv+LeakyReLU x 3 -> 1x1 (networks.py:44-52, no norm) on the split-precision path with the
    round-4 epilogue extras (cg_x3_epilogue): every un-normalised output leaves its kernel as {hi, lo} planes on the a-priori
    scale L1(W) * max|x| + max|b| (no measuring / splitting pass, no fp32 copy), and every activation backward is folded into
    the data-gradient epilogue of the layer above.  Against fp64 torch: outputs, data gradient, every weight / bias gradient;
    and against the same chain with the extras off (CG_BOUNDED_SPLIT=0-style switches): same numbers to 2e-6."""
    from council_gan_amd import ops
    from council_gan_amd.optim import ParamPool
    B, H = 2, 32
    cin = 3 if first == "thin3" else 64
    chans = [(cin, 64, 4, 2, 1, "lrelu"), (64, 128, 4, 2, 1, "lrelu"), (128, 256, 4, 2, 1, "lrelu"), (256, 256, 1, 1, 0, "none"),
             (256, 1, 1, 1, 0, "none")]

    def build():
        torch.manual_seed(5)
        nets = [[torch.nn.Conv2d(ci, co, k, s, bias=True) for ci, co, k, s, p, a in chans] for _ in range(n)]
        for m, net in enumerate(nets):
            for c in net:
                with torch.no_grad():
                    c.weight.mul_(1.0 + 0.3 * m)
                    c.bias.normal_(0, 0.1)
        pool = ParamPool([cga.FlatAdam([p for c in net for p in c.parameters()], lr=1e-4) for net in nets])
        pool.materialize('cuda')
        return nets, pool, ops.SplitWeights(pool)

    torch.manual_seed(9)
    x = cl((torch.rand(n * B, cin, H, H) * 2 - 1).cuda())
    gy = None
    saved = (ops.X3_FORWARD, ops.X3_BACKWARD, ops.X3_DYNAMIC_INPUT)
    ops.X3_FORWARD = ops.X3_BACKWARD = ops.X3_DYNAMIC_INPUT = True
    results = {}
    try:
        for mode in ("on", "off"):
            monkeypatch.setattr(ops, "BOUNDED_SPLIT", mode == "on")
            monkeypatch.setattr(ops, "FUSED_ACT_BWD", mode == "on")
            nets, pool, mgr = build()
            xi = x.clone().requires_grad_(True)
            h = xi
            used = []
            with ops.members(n):
                for li, (ci, co, k, s, p, a) in enumerate(chans):
                    c = nets[0][li]
                    h = ops.conv2d(h, c.weight, c.bias, s, p, a, wmgr=mgr, want_split=(co % 32 == 0), want_f32=(li == len(chans) - 1))
                    used.append((getattr(h, "_cg_no_f32", False), getattr(h, "_cg_split", None) is not None))
            if gy is None:
                torch.manual_seed(10)
                gy = torch.randn(h.shape).cuda()
            h.backward(gy)
            torch.cuda.synchronize()
            results[mode] = (h.detach().clone(), xi.grad.clone(), pool.grad.clone(), used, nets)
    finally:
        ops.X3_FORWARD, ops.X3_BACKWARD, ops.X3_DYNAMIC_INPUT = saved
    y_on, dx_on, g_on, used_on, nets = results["on"]
    y_off, dx_off, g_off, used_off, _ = results["off"]
    # the extras were in effect: the second and third LeakyReLU layers exist as planes only, nothing does with the switches off
    assert used_on[1] == (True, True) and used_on[2] == (True, True), used_on
    assert not any(u[0] for u in used_off), used_off
    assert rel(y_on, y_off) < 2e-6 and rel(dx_on, dx_off) < 2e-6 and rel(g_on, g_off) < 2e-6
    for m in range(n):
        rows = slice(m * B, (m + 1) * B)
        xr = x[rows].detach().double().cpu().requires_grad_(True)
        ws = [(c.weight.detach().double().cpu().requires_grad_(True), c.bias.detach().double().cpu().requires_grad_(True)) for c in nets[m]]
        hr = xr
        for (w, b), (ci, co, k, s, p, a) in zip(ws, chans):
            hr = ref_act(F.conv2d(hr, w, b, stride=s, padding=p), a)
        hr.backward(gy[rows].double().cpu())
        assert rel(y_on[rows], hr) < TOL, ("forward", m)
        assert rel(dx_on[rows], xr.grad) < TOL, ("data gradient", m)
        for li, (w, b) in enumerate(ws):
            assert rel(nets[m][li].weight._cg_grad, w.grad) < TOL, ("weight gradient", m, li)
            assert rel(nets[m][li].bias._cg_grad, b.grad) < 5 * TOL, ("bias gradient", m, li)


def test_value_keyed_device_constants_are_valid_on_every_stream(cga):
    """ops.take_rows caches its row-index vector on the device BY VALUE, and the cache is read from whatever stream asks next
    (the two discriminator-side updates run on two side streams).  The entry must be complete when it is published: a second
    stream that hits the cache while the first stream's queue is still long must read the indices, not the fresh allocation's
    previous contents (round 4: an out-of-bounds gather -> `Memory access fault` in the driver's run).  Same for the LSGAN
    target / weight vectors (networks._LossVectors)."""
    ops = cga.ops
    x = torch.randn(64, 8, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    filler = torch.randn(1 << 24, device='cuda')
    vec = cga.networks._LossVectors()
    torch.cuda.synchronize()
    for trial in range(6):
        idx = [(7 * trial + 3 * i + 1) % 64 for i in range(37 + trial)]          # a tuple no earlier test has used
        tgt = [float((trial + i) % 2) for i in range(19 + trial)]
        with torch.cuda.stream(a):
            for _ in range(40):
                ops.fill_(filler, float(trial))                                  # a few ms of queued work ahead of the miss
            ya = ops.take_rows(x, None, idx)
            ta, _ = vec.get(tgt, tgt, x.device)
        with torch.cuda.stream(b):
            yb = ops.take_rows(x, None, idx)                                     # cache hit, no ordering against stream a
            tb, _ = vec.get(tgt, tgt, x.device)
            tb = tb.clone()
        torch.cuda.synchronize()
        assert torch.equal(ya, x[idx]) and torch.equal(yb, x[idx])
        assert tb.tolist() == tgt and ta.tolist() == tgt
