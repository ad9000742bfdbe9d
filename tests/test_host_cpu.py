"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the
host logic (conv geometry, flat optimizer views, trainer construction / state_dict / RNG contract,
schedules) is right, and the product path refuses to run without a GPU instead of falling back."""
import copy
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

import council_gan_amd as cga
from council_gan_amd import ops
from golden_util import Golden, case_names
from oracle import council_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "council_gan_hip.h")).read()
    declared = set(re.findall(r"\b(cg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cg_stream_t"}
    assert len(declared) >= 35
    lib = ctypes.CDLL(cga.hip.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    # and the ctypes binding covers the same set
    assert declared == set(cga.hip.EXPORTS), declared ^ set(cga.hip.EXPORTS)
    assert cga.hip.load().cg_version() >= 100


def test_collective_entry_points_resolve_rccl_lazily():
    """cg_comm_*: RCCL is looked up on first use, not at load time; without a GPU a communicator cannot be created and
    the call says so instead of crashing (the single- and two-rank runs are -m gpu, tests/test_gpu_world.py)."""
    hdr = open(os.path.join(ROOT, "include", "council_gan_hip.h")).read()
    assert int(re.search(r"#define CG_COMM_ID_BYTES (\d+)", hdr).group(1)) == cga.hip.COMM_ID_BYTES
    try:
        uid = cga.hip.comm_unique_id()
    except cga.hip.HipError as e:               # a host without librccl: the message must name the cause
        assert "RCCL" in str(e)
        return
    assert len(uid) == cga.hip.COMM_ID_BYTES
    with pytest.raises(ValueError):
        cga.hip.Comm(uid[:5], 0, 1)
    with pytest.raises(cga.hip.HipError):
        cga.hip.Comm(uid, 3, 2)                  # rank out of range: argument check, before RCCL is asked
    if not torch.cuda.is_available():
        with pytest.raises(cga.hip.HipError):
            cga.hip.Comm(uid, 0, 1)


def test_geometry_struct_matches_header():
    assert ctypes.sizeof(cga.hip.ConvGeom) == 18 * 4 + 2 * 64


def test_no_cpu_fallback_on_hot_path():
    x = torch.randn(1, 4, 8, 8)
    w = torch.randn(4, 4, 3, 3)
    with pytest.raises(cga.hip.HipError):
        ops.conv2d(x, w, None, 1, 1)
    cfg = Golden("glasses_c1").cfg
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            tr.dis_update(torch.randn(1, 3, 32, 32), torch.randn(1, 3, 32, 32), cfg)
    with pytest.raises(cga.hip.HipError):
        tr.cuda('cpu')


def _brute_dgrad(dz, w, stride, pad, Hl, Wl):
    """dx of y = conv(x, w, stride, pad) by definition."""
    N, Cout, Ho, Wo = dz.shape
    _, Cin, KH, KW = w.shape
    dx = np.zeros((N, Cin, Hl, Wl))
    for oh in range(Ho):
        for ow in range(Wo):
            for kh in range(KH):
                for kw in range(KW):
                    ih, iw = oh * stride + kh - pad, ow * stride + kw - pad
                    if 0 <= ih < Hl and 0 <= iw < Wl:
                        dx[:, :, ih, iw] += np.einsum('no,oc->nc', dz[:, :, oh, ow], w[:, :, kh, kw])
    return dx


@pytest.mark.parametrize("K,stride,pad,Hl,Wl", [(3, 1, 1, 5, 4), (4, 2, 1, 8, 6), (4, 2, 1, 9, 7), (7, 1, 3, 6, 6),
                                                  (1, 1, 0, 3, 3), (3, 2, 1, 7, 8)])
def test_dgrad_parity_classes(K, stride, pad, Hl, Wl):
    """The tap tables handed to the kernel for the data-gradient reproduce the definition."""
    rng = np.random.RandomState(0)
    Ho, Wo = (Hl + 2 * pad - K) // stride + 1, (Wl + 2 * pad - K) // stride + 1
    dz = rng.randn(2, 3, Ho, Wo)
    w = rng.randn(3, 2, K, K)
    want = _brute_dgrad(dz, w, stride, pad, Hl, Wl)
    got = np.full((2, 2, Hl, Wl), np.nan)
    for ph, pw, Hc, Wc, taps in ops.dgrad_classes(Hl, Wl, K, K, stride, pad):
        for oy in range(Hc):
            for ox in range(Wc):
                acc = np.zeros((2, 2))
                for t, dy, dx in taps:
                    iy, ix = oy + dy, ox + dx
                    if 0 <= iy < Ho and 0 <= ix < Wo:
                        acc += np.einsum('no,oc->nc', dz[:, :, iy, ix], w[:, :, t // K, t % K])
                got[:, :, oy * stride + ph, ox * stride + pw] = acc
    assert not np.isnan(got).any(), "some input position is covered by no class"
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_fwd_geom():
    g = ops.fwd_geom(2, 8, 8, 3, 3, 1, 3, 3, 1, 1, 16, 2)
    assert (g.Ho, g.Wo, g.T, g.C1, g.C2, g.up) == (16, 16, 9, 3, 3, 1)
    assert [g.dy[t] for t in range(9)] == [-1, -1, -1, 0, 0, 0, 1, 1, 1]
    assert [g.dx[t] for t in range(9)] == [-1, 0, 1] * 3
    g = ops.fwd_geom(1, 9, 7, 8, 0, 0, 4, 4, 2, 1, 16, 0)
    assert (g.Ho, g.Wo) == (4, 3)


@pytest.mark.parametrize("name", case_names())
def test_trainer_init_matches_reference(name):
    """Same seeds as train.py:55-62 -> bit-identical initial weights and state_dict keys as the reference
    (fixtures were produced by the real reference constructor)."""
    g = Golden(name)
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    tr = cga.Council_Trainer(copy.deepcopy(g.cfg), 'cuda:0')
    st = g.init_state()
    for d in g.dirs:
        for net, attr in (('gen', 'gen_%s_s'), ('dis', 'dis_%s_s'), ('dis_council', 'dis_council_%s_s')):
            if net not in g.nets:
                continue
            for i in range(g.C):
                sd = getattr(tr, attr % d)[i].state_dict()
                assert set(sd) == set(st[d][net][i])
                for k, v in sd.items():
                    assert np.array_equal(v.numpy(), st[d][net][i][k]), (d, net, i, k)


def test_adain_offsets_follow_reference_order():
    g = Golden("m2f_c3")
    gen = cga.AdaINGen(3, g.cfg['gen'])
    offs = [(m.boff, m.goff, m.num_features) for m in gen.dec.modules() if isinstance(m, cga.AdaptiveInstanceNorm2d)]
    widths = O.OracleGen({}, g.cfg['gen']).adain_layout()
    assert [o[2] for o in offs] == widths
    start = 0
    for b, gm, c in offs:
        assert (b, gm) == (start, start + c)
        start += 2 * c
    assert start == gen.get_num_adain_params(gen.dec)


def test_colleague_draws_match_reference():
    g = Golden("m2f_c3")
    random.seed(1)
    n_rel = g.cfg['council']['numberOfCouncil_dis_relative_iteration']
    flat = []
    for i in range(g.C):
        flat += cga.Council_Trainer.draw_colleagues(i, g.C, n_rel)
    assert flat == list(g["it0/disc/choice"])


def test_flat_adam_views_and_state_dict_on_host():
    """Flat buffer bookkeeping (no kernels involved): views alias the buffer, conv weights keep OIHW
    shape with channels_last strides, torch.optim.Adam-compatible state_dict round-trips."""
    conv = torch.nn.Conv2d(4, 8, 3)
    lin = torch.nn.Linear(5, 3)
    params = list(conv.parameters()) + list(lin.parameters())
    before = [p.detach().clone() for p in params]
    opt = cga.FlatAdam(params, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    opt.materialize('cpu')
    f = opt.flat
    # every parameter starts on a 32-element boundary of the flat buffer (the interleaved fp16 mirror); the padding is zero
    assert all(o % 32 == 0 for o in f["offs"]) and f["offs"] == [0, 288, 320, 352]
    assert f["data"].numel() == 384 and float(f["data"][352 + 3:].abs().sum()) == 0.0
    assert float(f["data"][296:320].abs().sum()) == 0.0 and float(f["data"][335:352].abs().sum()) == 0.0
    for p, b in zip(params, before):
        assert torch.equal(p.detach(), b)
        assert p.data.untyped_storage().data_ptr() == f["data"].untyped_storage().data_ptr()
        assert p.grad is p._cg_grad
    assert tuple(conv.weight.shape) == (8, 4, 3, 3)
    assert conv.weight.is_contiguous(memory_format=torch.channels_last)
    assert conv.weight.stride() == (36, 1, 12, 4)
    conv.weight._cg_grad._cg_touched = True
    conv.bias._cg_grad._cg_touched = True
    assert opt.touched_runs() == [(0, 2)]
    sd = opt.state_dict()
    assert sd["param_groups"][0]["params"] == [0, 1, 2, 3] and sd["param_groups"][0]["betas"] == (0.5, 0.999)
    ref = torch.optim.Adam([torch.nn.Parameter(b.clone()) for b in before], lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    for p in ref.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    ref.step()
    opt.load_state_dict(ref.state_dict())
    assert opt._steps == [1, 1, 1, 1]
    back = opt.state_dict()
    for i in range(4):
        assert torch.allclose(back["state"][i]["exp_avg"], ref.state_dict()["state"][i]["exp_avg"])


def test_adam_step_scalars_on_host():
    """cg_adam_hyper (host arithmetic, no kernel): the two per-step scalars hipGraph mode stages into device memory are
    torch.optim.Adam's step_size and sqrt(bias_correction2) (trainer_council.py:170-179 builds torch.optim.Adam), rounded
    once from double; ParamPool.plan_hyper / advance keep the host-side step counts a replayed step would."""
    lib = cga.hip.load()
    buf = (ctypes.c_float * 2)()
    for lr, b1, b2, step in ((1e-4, 0.5, 0.999, 1), (1e-4, 0.5, 0.999, 2), (5e-5, 0.5, 0.999, 1000), (1e-3, 0.9, 0.99, 37)):
        assert lib.cg_adam_hyper(lr, b1, b2, step, buf) == 0
        assert buf[0] == np.float32(np.float64(np.float32(lr)) / (1.0 - np.float64(np.float32(b1)) ** step))
        assert buf[1] == np.float32(np.sqrt(1.0 - np.float64(np.float32(b2)) ** step))
    assert lib.cg_adam_hyper(1e-4, 0.5, 0.999, 0, buf) != 0 and b"cg_adam_hyper" in lib.cg_last_error()

    nets = [torch.nn.Conv2d(4, 8, 3) for _ in range(2)]
    opts = [cga.FlatAdam(list(n.parameters()), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-4) for n in nets]
    pool = cga.optim.ParamPool(opts)
    pool.materialize('cpu')
    assert pool.plan_hyper(0, 2) is None                  # the runs of a step are unknown before its first execution
    runs = [(0, 2)]
    pool._runs = {(0, 2): runs}
    v0 = pool.version
    h1 = pool.plan_hyper(0, 2)
    assert tuple(h1.shape) == (1, 2) and float(h1[0, 0]) == np.float32(1e-4 / 0.5)
    pool.advance(0, 2, runs)
    assert [o._steps for o in opts] == [[1, 1], [1, 1]] and pool.version == v0 + 2
    h2 = pool.plan_hyper(0, 2)
    assert float(h2[0, 0]) == np.float32(np.float64(np.float32(1e-4)) / (1.0 - 0.25))
    assert float(h2[0, 1]) == np.float32(np.sqrt(1.0 - np.float64(np.float32(0.999)) ** 2))


def test_council_discriminator_batch_plan():
    """MsImageDisCouncil.plan_members (host half of the member-batched council-discriminator objective) against the
    reference's loop (trainer_council.py:861-874): for every pick k of member m -- repeats included -- one
    mean(D(own)^2) + mean((D(colleague_k) - 1)^2) term, all scaled by council_w / n_rel."""
    b, scale = 2, 0.25
    draws = [[1, 3, 1], [0, 2, 2]]                          # member 0 drew colleague 1 twice, member 1 colleague 2 twice
    pk = [[(j, float(d.count(j))) for j in sorted(set(d))] for d in draws]
    idx, idx_in, tgt, wt = cga.networks.MsImageDisCouncil.plan_members(pk, 2, b, float(len(draws[0])), scale)
    rows = 2 * (1 + 2) * b                                  # per member: own + two distinct colleagues, b samples each
    assert len(idx) == len(tgt) == len(wt) == len(idx_in) == rows
    for m, d in enumerate(draws):
        base = m * 3 * b
        # own translations: rows m*b .. of x_full (non-negative indices), target 0, weight = one fake term per pick
        assert idx[base:base + b] == [m * b + r for r in range(b)]
        assert tgt[base:base + b] == [0.0] * b and wt[base:base + b] == [scale * len(d)] * b
        for u, j in enumerate(sorted(set(d))):
            lo = base + (1 + u) * b
            # colleague j: rows of x_cmp as negative indices -(row) - 1 (ops.take_rows), target 1, weight = its multiplicity
            assert idx[lo:lo + b] == [-(j * b + r) - 1 for r in range(b)]
            assert tgt[lo:lo + b] == [1.0] * b and wt[lo:lo + b] == [scale * d.count(j)] * b
        assert idx_in[base:base + 3 * b] == list(range(b)) * 3       # every block is conditioned on the same input batch
    with pytest.raises(ValueError):
        cga.networks.MsImageDisCouncil.plan_members([pk[0], pk[1][:1]], 2, b, 3.0, scale)


def test_member_groups_respect_the_31_bit_offsets():
    """Council_Trainer._plan_groups: as many members per launch as divide the local members, fit CG_GROUP and keep the
    largest batched activation (the council discriminator's first full-resolution map) below 2 GiB."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "male2female_council_folder.yaml")))

    def groups(council, batch, size):
        c = copy.deepcopy(cfg)
        c['council']['council_size'] = council
        tr = cga.Council_Trainer(c, 'cuda:0')                 # host-side construction only
        tr._group_max, tr._groups, tr._hp_last = 4, None, c
        return tr._plan_groups(torch.empty((batch, 3, size, size), device='meta'))
    assert groups(4, 4, 256) == [[0, 1, 2, 3]]                 # 4 x 336 MB
    assert groups(8, 4, 256) == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert groups(4, 8, 256) == [[0, 1], [2, 3]]               # 4 x 671 MB would cross 2 GiB
    assert groups(2, 16, 256) == [[0, 1]]                      # council 2: one colleague -> 2 x 16 samples per member, 2 x 0.5 GiB
    assert groups(2, 32, 256) == [[0], [1]]                    # 2 x 1 GiB would not fit
    with pytest.raises(cga.hip.HipError):
        groups(2, 64, 256)                                     # one member alone: 2 GiB


def test_bench_work_model_and_traffic_record():
    """bench.py's roofline inputs: W_min per iteration is SURVEY.md section 8d's table (0.130 / 1.040 / 14.68 / 31.0 TFLOP for
    configurations 1 / 2 / 3 / 5), the split-precision peak is the fp16 dense peak over three passes, and the committed PMC
    record the `traffic` field comes from names the dominant kernel, its bytes per launch and the build it was taken on."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert abs(bench.w_min_tflop(1, 128, 1, 4) - 0.130) < 5e-4
    assert abs(bench.w_min_tflop(8, 128, 1, 4) - 1.040) < 5e-4
    assert abs(bench.w_min_tflop(4, 256, 4, 4) - 14.68) < 5e-3
    assert abs(bench.w_min_tflop(4, 256, 8, 4) - 31.0) < 5e-2
    assert abs(bench.F16X3_PEAK_TFLOPS - 2500.0 / 3) < 1e-9 and bench.FP32_MFMA_PEAK_TFLOPS == 157.3
    assert bench.kernel_peak("conv_fwd_x3w_kernel<256,256,fast>") == bench.F16X3_PEAK_TFLOPS
    assert bench.kernel_peak("conv_fwd_thin_kernel<256,64,fast>") == bench.FP32_MFMA_PEAK_TFLOPS
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    name = "conv_fwd_x3w_kernel<256,256,fast>"
    assert name in rec
    nbytes, r = bench.pmc_traffic(name)
    # gfx950: FETCH_SIZE counts a 128-byte request as 64 bytes -> x 2 (MI355X_MICROARCH.md); WRITE_SIZE as is; both in KiB
    assert nbytes == rec[name]["bytes_per_launch"] == 1024 * (2 * rec[name]["fetch_size_kib"] + rec[name]["write_size_kib"])
    assert nbytes > rec[name]["algorithmic_bytes_per_launch"] > 0 and len(rec[name]["build_stamp"]) == 16
    assert set(("same_build", "build_stamp_now")) <= set(r)
    assert bench.pmc_traffic("no_such_kernel") == (None, None)
    for cfg_id, p in bench.PRESETS.items():
        assert os.path.exists(os.path.join(ROOT, "configs", p["config"])), cfg_id


def test_schedules_match_oracle():
    g = Golden("m2f_c3")
    for it in (0, 9999, 10000, 10001, 60000):
        for flip in (False, True):
            hp = copy.deepcopy(g.cfg)
            hp['iteration'] = it
            hp['council']['flipOnOff'] = flip
            assert cga.Council_Trainer._flip_state(hp) == O.council_flip_state(hp)


def test_input_oracle_definitions():
    """oracle/input_oracle.py against the arithmetic it restates (ToTensor = /255 in fp32, Normalize = (t - m) / s)."""
    from oracle import input_oracle as IO
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(9, 11, 3)).astype(np.uint8)
    t = IO.to_tensor_normalize(img)
    want = ((img.astype(np.float32) / np.float32(255)) - np.float32(0.5)) / np.float32(0.5)
    assert t.dtype == torch.float32 and tuple(t.shape) == (3, 9, 11)
    assert np.array_equal(t.numpy(), want.transpose(2, 0, 1))
    assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    s = IO.sample(img, 2, 3, 4, 5, flip_first=True)
    assert np.array_equal(s.numpy(), want[:, ::-1][2:6, 3:8].transpose(2, 0, 1))
    # flip-then-crop == crop-the-mirrored-window-then-flip
    l2 = IO.window_after_flip(3, 11, 5)
    assert np.array_equal(s.numpy(), IO.sample(img, 2, l2, 4, 5).numpy()[:, :, ::-1])


@pytest.mark.parametrize("interleaved", [True, False])
def test_split_tensor_layout_bookkeeping(interleaved, monkeypatch):
    """Host-side arithmetic of the {hi, lo} layouts (include/council_gan_hip.h, CG_X3_LO_ELEMS): lo offset, byte offset
    of a sub-tensor, reconstruction -- on host tensors, no kernel involved."""
    from council_gan_amd import ops
    monkeypatch.setattr(ops, "_X3_IL", interleaved)
    n, off = 96, 64
    vals = torch.arange(off + n, dtype=torch.float32) / 7.0
    hi = vals.half()
    lo = (vals - hi.float()).half()
    buf = torch.zeros(2 * (off + n), dtype=torch.float16)
    total = off + n
    if interleaved:
        g = buf.view(-1, 64)
        g[:, :32] = hi.view(-1, 32)
        g[:, 32:] = lo.view(-1, 32)
        assert ops.x3_lo(total) == 32
    else:
        buf[:total] = hi
        buf[total:] = lo
        assert ops.x3_lo(total) == total
    st = ops.SplitTensor(buf, (n,), off=off, lo=ops.x3_lo(total))
    assert st.hi_ptr().value - buf.data_ptr() == (4 if interleaved else 2) * off
    rec = st.to_float()
    want = hi[off:].float() + lo[off:].float()
    assert torch.equal(rec, want) and float((rec - vals[off:]).abs().max()) <= 2.0 ** -20 * float(vals.max())
    if interleaved:
        with pytest.raises(Exception):
            ops.SplitTensor(buf, (n,), off=8).hi_ptr()          # sub-tensors start on 32-element boundaries


def test_dgrad_mirror_prefetch_bookkeeping(monkeypatch):
    """ops.SplitWeights.prefetch_dgrad (gen_update's start): the layers that asked for data-gradient weights under the previous
    weight version are prepared again for the new one -- each once, on the given stream, only those without a current entry --
    and a lazy request afterwards finds them.  The library is replaced by a recorder: host bookkeeping only."""
    import types
    from council_gan_amd import hip, ops

    calls = []

    class Lib:
        def cg_split_f16_dynamic_capped(self, *a):
            calls.append(("split", a[-1].value))
            return 0

        def cg_conv2d_dgrad_x3_wt_elems(self, g, nci):
            return 64 * nci

        def cg_conv2d_dgrad_x3_prep(self, *a):
            calls.append(("prep", a[-1].value))
            return 0

    monkeypatch.setattr(ops, "_lib", lambda: Lib())
    monkeypatch.setattr(ops, "_X3_IL", True)
    monkeypatch.setattr(ops, "stream", lambda: ops.c_void_p(111))
    monkeypatch.setattr(ops, "ptr", lambda t: None if t is None else ops.c_void_p(t.data_ptr()))     # (the real one refuses host tensors)
    w1, w2 = torch.nn.Parameter(torch.zeros(64, 32, 3, 3)), torch.nn.Parameter(torch.zeros(32, 32, 1, 1))
    owner = types.SimpleNamespace(version=0, _flat=True, _params=[w1, w2],
                                  flat={'data': torch.zeros(w1.numel() + w2.numel()), 'offs': [0, w1.numel()]})
    mgr = ops.SplitWeights(owner)
    side = types.SimpleNamespace(cuda_stream=222)
    g = hip.ConvGeom()
    assert mgr.prefetch_dgrad(side) == 0 and not calls                      # nothing was ever asked for: nothing to prepare
    a = mgr.dgrad_weights(w1, w1, g, 0, 32, None, 1)                        # lazy request: built on the current stream, planned
    assert [c for c in calls if c[0] == "prep"] == [("prep", 111)] and a is mgr.dgrad_weights(w1, w1, g, 0, 32, None, 1)
    b = mgr.dgrad_weights(w2, w2, g, 0, 32, None, 1)
    assert mgr.prefetch_dgrad(side) == 0                                    # both entries are current
    owner.version = 1                                                       # an optimizer step
    calls.clear()
    assert mgr.prefetch_dgrad(side) == 2
    assert [c for c in calls if c[0] == "prep"] == [("prep", 222)] * 2 and calls[0] == ("split", 111)      # mirror first, own stream
    calls.clear()
    a2 = mgr.dgrad_weights(w1, w1, g, 0, 32, None, 1)                       # the backward finds them: no launch
    assert not calls and a2 is not a and a2.numel() == a.numel() and b.numel() == 2 * 64 * 32
    assert mgr.prefetch_dgrad(side) == 0


def test_tool_helpers_on_host(tmp_path):
    """The host-only helpers of tools/translate_folder.py and tools/train_synthetic.py (image loading geometry, strip layout)."""
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import train_synthetic as TS
    import translate_folder as TF
    rng = np.random.RandomState(1)
    src = rng.randint(0, 256, size=(100, 150, 3)).astype(np.uint8)
    Image.fromarray(src).save(tmp_path / "wide.png")
    Image.fromarray(src.transpose(1, 0, 2).copy()).save(tmp_path / "tall.png")
    for name in ("wide.png", "tall.png"):
        a = TF.load_image(str(tmp_path / name), 72, 64, 64)          # shorter side -> 72, centre crop 64x64
        assert a.shape == (64, 64, 3) and a.dtype == np.uint8 and a.flags.writeable
    same = TF.load_image(str(tmp_path / "wide.png"), None, 100, 150)  # no resize, crop = whole image
    assert np.array_equal(same, src)
    with pytest.raises(ValueError):
        TF.load_image(str(tmp_path / "wide.png"), 32, 64, 64)
    u8_a, u8_b = TS.synthetic_u8(rng, 3, 16)
    assert u8_a.shape == u8_b.shape == (3, 16, 16, 3) and u8_a.dtype == np.uint8
    rows = (torch.full((2, 3, 8, 8), -1.0), None, torch.full((2, 3, 8, 8), 1.0))
    TS.save_strip(rows, str(tmp_path / "strip.png"))
    strip = np.asarray(Image.open(tmp_path / "strip.png"))
    assert strip.shape == (16, 16, 3) and strip[:8].max() == 0 and strip[8:].min() == 255
    TF.save_normalised(torch.linspace(-3, 5, 3 * 4 * 4).view(1, 3, 4, 4), str(tmp_path / "n.png"))
    n = np.asarray(Image.open(tmp_path / "n.png"))
    assert n.shape == (4, 4, 3) and n.min() == 0 and n.max() == 255


def test_exclusive_time_report_on_a_synthetic_trace(tmp_path, capsys):
    """tools/rocpd_report.py alone: time with exactly one kernel running goes to that kernel, overlapped time is shared 1/n, an idle
    gap goes to the kernel that ended before it -- and the three columns add up to the window."""
    import sqlite3
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import rocpd_report as R
    db = str(tmp_path / "t.db")
    c = sqlite3.connect(db)
    c.execute("create table kernels(name text, start int, end int)")
    # ns:  a [0, 1000)   b [500, 1500)   (idle 1500-2000)   c [2000, 2500)   a [2500, 3000)
    c.executemany("insert into kernels values (?, ?, ?)",
                  [("void (anonymous namespace)::a<1, 2>(float*)", 0, 1000), ("b(int)", 500, 1500), ("c", 2000, 2500),
                   ("void (anonymous namespace)::a<1, 2>(float*)", 2500, 3000)])
    c.commit()
    c.close()
    R.alone(db, 1.0)
    out = capsys.readouterr().out
    rows = {}
    for line in out.splitlines():
        f = line.split()
        if len(f) >= 5 and f[0] in ("a<1,", "b", "c", "TOTAL"):
            rows[f[0]] = [float(v) for v in f[-3:]] + [int(f[-4])]
    assert rows["a<1,"][:3] == [0.001, 0.0, 0.0] and rows["a<1,"][3] == 2          # 500 + 500 ns alone, 250 ns shared (rounds to 0.000)
    assert rows["b"][0] == 0.001 and rows["b"][2] == 0.001                          # 500 ns alone, the 500 ns gap follows it
    assert rows["c"][0] == 0.001
    assert abs(sum(rows["TOTAL"][:3]) - 0.003) < 1.1e-3                            # = the 3000 ns window (three-decimal ms columns)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` with no launcher around it must start N ranks itself (torch.distributed.run, rendezvous on
    127.0.0.1) -- the way the driver runs it.  CG_BENCH_DRY=1: everything but the training step, on CPU over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    env = dict(os.environ, CG_BENCH_DRY="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["ranks"] == n and out["steps"] == 2 and out["warmup"] == 1
    if n == 8:      # 8 GPUs default to BASELINE.json configs[4]: council 8, one member per GPU
        assert out["config"]["council"] == 8 and out["config"]["members_per_rank"] == 1


def test_bench_refuses_a_world_size_mismatch():
    """Under a launcher that started a different number of ranks than --gpus says, bench.py fails instead of measuring
    the wrong job."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    env = dict(os.environ, CG_BENCH_DRY="1", WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1", CG_BENCH_SPAWNED="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0


def test_input_oracle_matches_the_pil_fixture():
    """SURVEY.md 8f.3 pin: oracle/input_oracle.py against tests/golden/input_pil.npz -- flip and crop done by PIL itself,
    ToTensor / Normalize by NumPy (oracle/make_input_golden.py), bit for bit."""
    from oracle import input_oracle as IO
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "input_pil.npz"))
    H, W = int(z["height"]), int(z["width"])
    for n in range(len(z["images"])):
        got = IO.sample(z["images"][n], int(z["tops"][n]), int(z["lefts"][n]), H, W, flip_first=bool(z["flips"][n]))
        assert got.dtype == torch.float32 and np.array_equal(got.numpy(), z["out"][n]), n


def test_upsample_conv_equals_the_summed_tap_transposed_conv():
    """The algebra behind cg_upconv_* (include/council_gan_hip.h): nn.Upsample(2x nearest) -> ZeroPad2d(1) -> Conv2d(3x3)
    (/root/reference/networks.py:385-386, 513-516) equals conv_transpose2d(x, W_F, stride 2, padding 1) with the 4x4 kernel
    W_F[u][v] = sum_{kh in S(u), kw in S(v)} W[kh][kw], S(0) = {2}, S(1) = {1, 2}, S(2) = {0, 1}, S(3) = {0}; its input gradient is
    conv2d(dz, W_F, stride 2, padding 1) and the 3x3 weight gradient is the fold of the 4x4 one.  fp64 torch on the CPU: this pins
    the tap sets the device code hard-wires (csrc/conv_gemm.hip: upc_first / upc_count)."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    S = {0: (2,), 1: (1, 2), 2: (0, 1), 3: (0,)}
    x = torch.randn(2, 5, 6, 7, dtype=torch.float64, requires_grad=True)
    w = torch.randn(4, 5, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, dtype=torch.float64)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    wf = torch.zeros(5, 4, 4, 4, dtype=torch.float64)              # conv_transpose2d weight: [Cin][Cout][4][4]
    for u in range(4):
        for v in range(4):
            wf[:, :, u, v] = sum(w.detach()[:, :, kh, kw] for kh in S[u] for kw in S[v]).t()
    wf.requires_grad_(True)
    xt = x.detach().clone().requires_grad_(True)
    yt = F.conv_transpose2d(xt, wf, b, stride=2, padding=1)
    assert yt.shape == y.shape and float((yt - y).abs().max()) < 1e-12
    dz = torch.randn_like(y)
    y.backward(dz)
    yt.backward(dz)
    assert float((xt.grad - x.grad).abs().max()) < 1e-12
    # data gradient = the 4x4 stride-2 pad-1 convolution over dz with W_F as [Cin][Cout][4][4] -> conv2d weight [Cin][Cout][4][4]
    dx = F.conv2d(dz, wf.detach(), stride=2, padding=1)
    assert float((dx - x.grad).abs().max()) < 1e-12
    # fold of the 4x4 weight gradient back onto the nine taps
    dw = torch.zeros_like(w)
    for u in range(4):
        for v in range(4):
            for kh in S[u]:
                for kw in S[v]:
                    dw[:, :, kh, kw] += wf.grad[:, :, u, v].t()
    assert float((dw - w.grad).abs().max()) < 1e-11
    # the device code's closed form of S(u): first = {2, 1, 0, 0}[u], count = {1, 2, 2, 1}[u]
    for u in range(4):
        first, count = (2, 1, 0, 0)[u], (1, 2, 2, 1)[u]
        assert tuple(range(first, first + count)) == S[u]


def test_bounded_split_scale_rule():
    """cg_x3_epilogue's a-priori scale (csrc/conv_x3.inc: x3_bound_scale): 2^(14 - floor(log2 bound)) puts scale * bound into
    [2^14, 2^15) -- below fp16's largest finite value with a binade to spare -- for any bound a float can hold between the clamps."""
    import math
    for bound in (1e-20, 3e-7, 0.02, 0.99999, 1.0, 1.5, 65.0, 4096.0, 3e9, 1e20):
        e = math.floor(math.log2(bound))
        se = min(max(14 - e, 27 - 127), 227 - 127)
        s = 2.0 ** se
        if 27 - 127 < 14 - e < 227 - 127:
            assert 2 ** 14 <= s * bound < 2 ** 15
        assert s * bound < 65504 or 14 - e < 27 - 127


def _header_prototypes():
    """{name: (return class, [parameter classes])} parsed from include/council_gan_hip.h; classes: 'p' pointer / cg_stream_t,
    'i' int / unsigned, 'f' float, 'd' double, 'q' size_t and the other 64-bit integers."""
    hdr = open(os.path.join(ROOT, "include", "council_gan_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", " ", hdr)

    def cls(decl):
        decl = decl.strip()
        if "*" in decl or re.search(r"\bcg_stream_t\b", decl) or "[" in decl:
            return "p"
        if re.search(r"\bsize_t\b", decl):           # ctypes.c_size_t IS c_ulong here: one class for every 64-bit integer
            return "q"
        if re.search(r"\b(u?int64_t|long long|unsigned long long|long|unsigned long)\b", decl):
            return "q"
        if re.search(r"\bdouble\b", decl):
            return "d"
        if re.search(r"\bfloat\b", decl):
            return "f"
        if re.search(r"\b(int|unsigned|uint32_t|int32_t)\b", decl):
            return "i"
        raise AssertionError("unclassified parameter: %r" % decl)

    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(cg_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", hdr):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        params = [] if args in ("", "void") else [cls(a) for a in args.split(",")]
        protos[name] = (cls(ret) if ret != "void" else "v", params)
    return protos


def test_ctypes_signatures_match_the_header():
    """Every entry point's ctypes declaration (council_gan_amd/hip.py) against its prototype in include/council_gan_hip.h:
    parameter count and the class of every parameter -- an int bound where the header says size_t or float would pass the
    symbol test and corrupt the call on the GPU box."""
    import ctypes as C
    protos = _header_prototypes()
    assert set(protos) == set(cga.hip.EXPORTS), set(protos) ^ set(cga.hip.EXPORTS)

    def cls(t):
        if t is None:
            return "v"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "_type_") and not isinstance(t._type_, str):
            return "p"               # POINTER(struct / int / ...)
        return {C.c_int: "i", C.c_uint: "i", C.c_float: "f", C.c_double: "d", C.c_size_t: "q", C.c_int64: "q",
                C.c_uint64: "q", C.c_longlong: "q", C.c_ulonglong: "q", C.c_long: "q", C.c_ulong: "q"}[t]

    bad = []
    for name, (res, args) in cga.hip._SIGS.items():
        want_ret, want = protos[name]
        got = [cls(a) for a in args]
        if got != want or cls(res) != want_ret:
            bad.append((name, want_ret + ":" + "".join(want), cls(res) + ":" + "".join(got)))
    assert not bad, bad


def test_gpu_suite_order_and_perf_marker():
    """with `-x`, one fault must not hide the cheap, wide parity tests -- the GPU suite collects operators first
    and the heavy full-width graph runs last (tests/conftest.py: _ORDER), and wall-clock assertions are not part of `-m gpu` at all."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", "tests", "-q", "-m", "gpu", "--collect-only", "-p", "no:cacheprovider"],
                         cwd=root, capture_output=True, text=True, timeout=300).stdout
    files = []
    for line in out.splitlines():
        if "::" in line:
            f = line.split("::")[0]
            if not files or files[-1] != f:
                files.append(f)
    assert files[0].endswith("test_gpu_ops.py") and files[-1].endswith("test_gpu_graph.py"), files
    assert len(files) == len(set(files)), files                      # every file's tests stay together
    assert not any(f.endswith("test_gpu_perf.py") for f in files)    # marker `perf`, not `gpu`
    perf = subprocess.run([sys.executable, "-m", "pytest", "tests", "-q", "-m", "perf", "--collect-only", "-p", "no:cacheprovider"],
                          cwd=root, capture_output=True, text=True, timeout=300).stdout
    assert "test_gpu_perf.py::test_graph_mode_host_cost" in perf


def test_value_keyed_constant_caches_are_bounded_until_a_graph_is_captured():
    """ops._idx_cache / networks._LossVectors grow with every distinct value (varying colleague picks, a ramped loss
    weight).  hip.const_cache_put bounds them at CONST_CACHE_MAX entries -- until the first hipGraph capture pins them (graphs
    hold the addresses)."""
    import council_gan_amd as cga
    hip = cga.hip
    pinned = hip._const_pinned[0]
    try:
        hip._const_pinned[0] = False
        cache = {}
        for i in range(hip.CONST_CACHE_MAX + 5):
            hip.const_cache_put(cache, i, i)
        assert len(cache) == 5 and (hip.CONST_CACHE_MAX + 4) in cache          # cleared once, at the bound
        hip.pin_const_caches()
        for i in range(hip.CONST_CACHE_MAX + 5):
            hip.const_cache_put(cache, ("p", i), i)
        assert len(cache) == hip.CONST_CACHE_MAX + 10                            # pinned: nothing is ever dropped
    finally:
        hip._const_pinned[0] = pinned


def test_hot_kernels_use_no_scratch_and_keep_their_occupancy():
    """Compile-time guard (no GPU): registers / scratch of the kernels the step spends its time in, read from the built library's
    code-object metadata (tools/kernel_resources.py).  A loop left rolled by hipcc moved the 128x128 wave tile's 256 accumulators to
    scratch in round 6 until it became a compile-time loop -- silently, but for this table."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("ROCm LLVM tools not installed")
    root = os.path.join(os.path.dirname(__file__), "..")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), "conv_fwd_x3w_kernel", "conv_wgrad_x3tw_kernel",
                          "conv_fwd_x3_kernel", "conv_fwd_pipe_kernel", "conv_fwd_thin_x3_kernel"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines()[1:]:
        f = line.split()
        name, (agpr, vgpr, spill, sgpr, lds, scratch, waves) = " ".join(f[:-7]), map(int, f[-7:])
        rows[name] = dict(agpr=agpr, vgpr=vgpr, spill=spill, lds=lds, scratch=scratch, waves=waves)
    assert len(rows) >= 40, len(rows)
    dom = [r for n, r in rows.items() if "conv_fwd_x3w_kernel" in n and "256" in n]
    assert dom and all(r["scratch"] == 0 for r in dom)                       # every variant of the wide tile, the 4-wave ones included
    spilled = {n: r for n, r in rows.items() if r["scratch"] > 16}           # (the chunked 128x128 fp32 tile: 2 dwords by design)
    assert not spilled, spilled
    wide = [r for n, r in rows.items() if "conv_fwd_x3w_kernel" in n and "ILi256ELi256ELi128ELi64ELi1ELi0ELi0E" in n]
    assert wide and wide[0]["waves"] >= 2 and wide[0]["lds"] <= 160 * 1024    # two waves per SIMD, one block per CU
