"""Whole-iteration parity of the HIP trainer against the oracle (fp32 run + fp64 twin) -- shared by the GPU tests.

Criteria (north star: within 1e-3 rel-fp32):
  * every loss <= 1e-3 relative to the fp32 oracle, except the mask_zero_one criterion (mean 1/(|m-c|+eps): it amplifies
    a mask perturbation by up to 1/eps^2), which is judged like the generator gradients;
  * discriminator / council-discriminator gradients: l2-rel error against the fp64 oracle <= 1e-3;
  * generator gradients: the reference's own fp32-vs-fp64 gradient gap is 2-4e-3 and chaotic in the
    forward round-off; judged by level (within the measured chaos band of the reference arithmetic itself) AND per-tensor
    uniformity of the error -- see GEN_GRAD_FACTOR / check_gen_grad below;
  * post-Adam weights: mean |w_ours - w_fp64| <= max(2 x mean |w_fp32 - w_fp64|, 2e-6) per network (one Adam step moves
    every weight by ~lr = 1e-4 in the direction of its gradient's sign, so round-off-sized gradients flip steps in the
    reference too)."""
import copy
import random

import numpy as np
import torch

from oracle import council_oracle as O

ACT_TOL = 1e-3
# Generator gradients.  The error of ANY fp32 evaluation against fp64 is a single global perturbation born at the loss
# head (the mask criteria 1/(|m - c| + eps), the steep tanh(10 x) mask, ReLU / LeakyReLU sign decisions): every tensor of a
# generator carries the SAME relative error (tools/diag_gengrad.py, profiles/r02_gengrad_diag.txt: e.g. 2.9e-3 +- 3 % over
# all 60 tensors), and its size is CHAOTIC in the forward round-off, not a property of the backward arithmetic: re-running
# the reference arithmetic itself (the fp32 oracle, same ATen kernels) with every Conv2dBlock output perturbed by a relative
# 6e-7 -- one extra rounding -- moves that error by 0.8x ... 18x, member by member (tools/diag_gengrad_lottery.py,
# profiles/r02_gengrad_lottery.txt: median 2.4x, 90th percentile 13x on anime2face 128^2).  The HIP path's forward round-off
# is ~2x the CPU's (sequential MFMA accumulation along K; profiles/r02_forward_error.txt), i.e. such a perturbation.  Hence:
#   (1) level:      err(ours, fp64) <= max(GEN_GRAD_FACTOR x err(fp32 oracle, fp64), 1e-3), factor 20 = the chaos band;
#   (2) uniformity: once the error is in the chaotic regime (> 1e-3), no tensor that carries >= 2 % of the gradient norm
#                   has a relative error above 3 x the overall one -- a wrong backward kernel shows up in ITS tensors,
#                   not as a global scale (this is the check that discriminates; the discriminator gradients pin the same
#                   kernels at 1e-7).  Tensors next to the loss head sit BELOW the common level, which is fine.
GEN_GRAD_FACTOR = 20.0
GEN_GRAD_UNIFORM = 3.0


def check_gen_grad(gs, r32, r64, what):
    """Asserts criteria (1) and (2) for one generator; returns (err ours, err fp32 oracle)."""
    keys = list(r64)
    e_ours, e_ref = l2rel(gs, r64, keys), l2rel(r32, r64, keys)
    assert e_ours <= max(GEN_GRAD_FACTOR * e_ref, ACT_TOL), ("generator gradient level", what, e_ours, e_ref)
    tot = np.sqrt(sum(float((r64[k].astype(np.float64) ** 2).sum()) for k in keys))
    if e_ours > ACT_TOL:                  # below that the level criterion alone already is the 1e-3 tolerance
        for k in keys:
            n = np.sqrt(float((r64[k].astype(np.float64) ** 2).sum()))
            if n < 0.02 * tot:
                continue
            e_k = np.sqrt(float(((gs[k].astype(np.float64) - r64[k].astype(np.float64)) ** 2).sum())) / n
            assert e_k <= e_ours * GEN_GRAD_UNIFORM, ("generator gradient: tensor above the common error level", what, k, e_k, e_ours)
    return e_ours, e_ref
NETS = (("dis", "dis", "dis_%s_s"), ("disc", "dis_council", "dis_council_%s_s"), ("gen", "gen", "gen_%s_s"))


def np_(t):
    return t.detach().float().cpu().numpy()


def grads_of(net):
    out = {}
    for k, p in net.named_parameters():
        gbuf = getattr(p, '_cg_grad', None)
        if gbuf is not None and gbuf._cg_touched:
            out[k] = np_(gbuf)
    return out


def weights_of(net):
    return {k: np_(v) for k, v in net.state_dict().items() if 'running_' not in k}


def l2rel(a, b, keys=None):
    keys = list(b if keys is None else keys)
    num = np.sqrt(sum(float(((a[k].astype(np.float64) - b[k].astype(np.float64)) ** 2).sum()) for k in keys))
    den = np.sqrt(sum(float((b[k].astype(np.float64) ** 2).sum()) for k in keys))
    return num / max(den, 1e-30)


def mean_abs_diff(a, b, keys):
    n = sum(a[k].size for k in keys)
    return sum(float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).sum()) for k in keys) / max(n, 1)


def lossvec(v):
    return np.array([float(t.detach()) if torch.is_tensor(t) else float(t) for t in v], dtype=np.float64)


def host_state(tr):
    """The trainer's (host-resident) weights in the oracle's input format."""
    st = {}
    for d in tr._dirs:
        st[d] = {'gen': [O.to_numpy_state(m.state_dict()) for m in tr._nets('gen', d)],
                 'dis': [O.to_numpy_state(m.state_dict()) for m in tr._nets('dis', d)]}
        if tr.do_dis_council:
            st[d]['dis_council'] = [O.to_numpy_state(m.state_dict()) for m in tr._nets('disc', d)]
    return st


def run_oracle(cfg, state, x_a, x_b, rng, dtype):
    """One iteration of the oracle from the given host-RNG state; gradients snapshotted right after each update
    (gen_update's backward also deposits gradients in the discriminators, in the oracle as in the reference)."""
    random.setstate(rng[0]); torch.set_rng_state(rng[1])
    cfg = copy.deepcopy(cfg)
    otr = O.OracleTrainer(cfg, state, dtype=dtype)
    grads, post = {}, {}

    def snap(kind, onet):
        for d in otr.dirs:
            for i in range(otr.C):
                sd = otr.sd[d][onet][i]
                grads[(kind, d, i)] = {k: t.grad.float().numpy().copy() if dtype == torch.float32 else t.grad.numpy().copy()
                                       for k, t in sd.items() if t.requires_grad and t.grad is not None}
                post[(kind, d, i)] = {k: t.detach().numpy().copy() for k, t in sd.items() if 'running_' not in k}
    otr.dis_update(x_a, x_b, cfg); snap("dis", "dis")
    ran = otr.dis_council_update(x_a, x_b, cfg)
    if ran:
        snap("disc", "dis_council")
    otr.gen_update(x_a, x_b, cfg, cfg['iteration']); snap("gen", "gen")
    return otr, grads, post, bool(ran)


def iteration_vs_oracle(cga, cfg, size, batch, seed=1, report=None):
    """Builds the HIP trainer from `seed`, runs ONE train.py:237-250 iteration on it and on the oracle (fp32 and fp64)
    from identical weights, inputs and host-RNG state, and asserts the criteria in the module docstring.
    Returns the measured errors."""
    cfg = copy.deepcopy(cfg)
    cfg['batch_size'] = batch
    cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = size
    O.seed_all(seed)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    state = host_state(tr)
    tr.cuda('cuda:0')
    x_a, x_b = O.synthetic_batch(batch, size)
    rng = (random.getstate(), torch.get_rng_state())
    got_g, got_w = {}, {}

    def snap(kind, attr):
        for d in tr._dirs:
            for i in range(tr.council_size):
                m = getattr(tr, attr % d)[i]
                got_g[(kind, d, i)] = grads_of(m)
                got_w[(kind, d, i)] = weights_of(m)
    tr.dis_update(x_a, x_b, cfg); snap("dis", "dis_%s_s")
    tr.dis_council_update(x_a, x_b, cfg)
    ran_disc = tr.do_dis_council and tr.council_size > 1 and O.dis_council_active(cfg, tr.council_size)
    if ran_disc:
        snap("disc", "dis_council_%s_s")
    tr.gen_update(x_a, x_b, cfg, cfg['iteration']); snap("gen", "gen_%s_s")
    torch.cuda.synchronize()

    o32, g32, w32, ran32 = run_oracle(cfg, state, x_a, x_b, rng, torch.float32)
    o64, g64, w64, _ = run_oracle(cfg, state, x_a, x_b, rng, torch.float64)
    assert ran32 == bool(ran_disc)

    errs = {}
    # ---- losses ---------------------------------------------------------------------------------------------------
    def close(name, mine, ref):
        mine, ref = lossvec(mine), lossvec(ref)
        e = float(np.max(np.abs(mine - ref) / np.maximum(np.abs(ref), 1e-7))) if len(ref) else 0.0
        errs["loss/" + name] = e
        assert e <= ACT_TOL, (name, mine, ref)
    close("dis_total", tr.loss_dis_total_s, o32.loss_dis_total)
    if ran32:
        close("disc_total", tr.loss_dis_council_total_s, o32.loss_disc_total)
    close("gen_total", tr.loss_gen_total_s, o32.loss_gen_total)
    for d in tr._dirs:
        ab = 'ab' if d == 'a2b' else 'ba'
        close("gen_adv_" + d, getattr(tr, 'loss_gen_adv_%s_s' % d), o32.loss_gen_adv[d])
        if ran32:
            close("council_" + d, getattr(tr, 'council_loss_%s_s' % ab), o32.council_loss[d])
        if len(o32.loss_mask_zero_one[d]):
            mine = lossvec(getattr(tr, 'loss_gen_mask_zero_one_%s_s' % ab))
            r32, r64 = lossvec(o32.loss_mask_zero_one[d]), lossvec(o64.loss_mask_zero_one[d])
            tol = np.maximum(ACT_TOL * np.abs(r64), 2 * np.abs(r32 - r64)) + 1e-7
            errs["loss/mask_zero_one_" + d] = float(np.max(np.abs(mine - r64) / np.abs(r64)))
            assert np.all(np.abs(mine - r64) <= tol), ("mask_zero_one", d, mine, r32, r64)
        if cfg['mask_total_w'] != 0 and cfg['iteration'] > cfg['focus_loss']['focus_loss_start_at_iter']:
            close("mask_total_" + d, getattr(tr, 'loss_gen_mask_total_%s_s' % ab), o32.loss_mask_total[d])
        if cfg['mask_tv_w'] != 0 and cfg['iteration'] > cfg['focus_loss']['focus_loss_start_at_iter']:
            close("mask_tv_" + d, getattr(tr, 'loss_gen_mask_TV_%s_s' % ab), o32.loss_mask_tv[d])
    # ---- gradients and post-step weights --------------------------------------------------------------------------------
    for key, gs in got_g.items():
        kind = key[0]
        r64, r32 = g64[key], g32[key]
        assert set(gs) == set(r64), (key, set(gs) ^ set(r64))
        if kind == "gen":
            e_ours, e_ref = check_gen_grad(gs, r32, r64, key)
        else:
            e_ours, e_ref = l2rel(gs, r64), l2rel(r32, r64)
            assert e_ours <= ACT_TOL, ("discriminator gradient", key, e_ours, e_ref)
        errs[("grad",) + key] = (e_ours, e_ref)
        keys = list(r64)
        w_ours, w_ref = mean_abs_diff(got_w[key], w64[key], keys), mean_abs_diff(w32[key], w64[key], keys)
        errs[("post",) + key] = (w_ours, w_ref)
        assert w_ours <= max(2 * w_ref, 2e-6), ("post-step weights", key, w_ours, w_ref)
    if report is not None:
        print("\n[%s] losses (rel vs fp32 oracle): %s" % (report, {k[5:]: "%.1e" % v for k, v in errs.items()
                                                                 if isinstance(k, str)}))
        print("[%s] grad l2-rel vs fp64 (ours, fp32 oracle): %s" %
              (report, {"%s/%s/%d" % k[1:]: ("%.1e" % v[0], "%.1e" % v[1]) for k, v in errs.items()
                        if not isinstance(k, str) and k[0] == "grad"}))
        print("[%s] post-step mean|dw| vs fp64 (ours, fp32 oracle): %s" %
              (report, {"%s/%s/%d" % k[1:]: ("%.1e" % v[0], "%.1e" % v[1]) for k, v in errs.items()
                        if not isinstance(k, str) and k[0] == "post"}))
    del tr
    return errs
