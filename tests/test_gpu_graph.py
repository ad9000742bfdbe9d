"""hipGraph mode (CG_GRAPH=1 / cfg['cg_graph'], council-gan_amd/graphs.py): every update's device work is captured once and
replayed; only the host-derived inputs (style noise, colleague picks, Adam's per-step scalars, the loss-matching ring
positions) are refreshed per iteration.  The replayed iterations must be BIT-IDENTICAL to the eager ones -- same kernels,
same launch parameters, same inputs -- and the host must spend a small fraction of the eager enqueue time."""
import copy
import os
import time

import numpy as np
import pytest
import torch
import yaml

import parity_util as P
from oracle import council_oracle as O

pytestmark = pytest.mark.gpu
CONFIGS = os.path.join(os.path.dirname(__file__), "..", "configs")


@pytest.fixture(scope="module")
def cga():
    import council_gan_amd
    council_gan_amd.hip.load()
    return council_gan_amd


def _tiny(name, council, batch):
    cfg = yaml.safe_load(open(os.path.join(CONFIGS, name)))
    cfg['gen'].update(dim=16, mlp_dim=32, n_res=2)
    cfg['dis'].update(dim=16)
    cfg['council']['council_size'] = council
    cfg['batch_size'] = batch
    cfg['iteration'] = 60000
    return cfg


def _run(cga, cfg, graph, iters, size, group_max=None, overlap=None, expect_graph_mode_after=None, dis_twice=False):
    c = copy.deepcopy(cfg)
    c['cg_graph'] = '1' if graph else '0'
    O.seed_all(11)
    tr = cga.Council_Trainer(c, 'cuda:0')
    tr.cuda('cuda:0')
    if group_max is not None:
        tr._group_max = group_max
    assert tr._graph_mode == bool(graph)
    names = ['loss_dis_total_s', 'loss_gen_total_s'] + (['loss_dis_council_total_s', 'council_loss_ab_s', 'council_loss_ba_s']
                                                        if tr.council_size > 1 else [])
    rows = []
    for it in range(iters):
        O.seed_all(100 + it)                                   # host RNG: style noise + colleague picks of this iteration
        if it == 0 or not os.environ.get('CG_DIAG_SAME_X'):
            x_a, x_b = O.synthetic_batch(c['batch_size'], size, seed=7 + it)
            x_a, x_b = x_a.cuda(), x_b.cuda()
        c['iteration'] = 60000 + it
        if dis_twice:          # train.py:241-246 with dis.numberOf_dis_relative_iteration = 2: an extra discriminator update on
            y_a, y_b = O.synthetic_batch(c['batch_size'], size, seed=900 + it)      # ANOTHER batch before this iteration's three
            tr.dis_update(y_a.cuda(), y_b.cuda(), c)
        tr.dis_update(x_a, x_b, c)
        tr.dis_council_update(x_a, x_b, c)
        tr.gen_update(x_a, x_b, c, c['iteration'])
        torch.cuda.synchronize()
        row = {n: [float(v) for v in getattr(tr, n, [])] for n in names}
        if os.environ.get('CG_DIAG_POOLS'):      # per-pool checksums of weights / gradients / moments (tools/probes/diag_graph.py)
            for kind, pool in tr._pools.items():
                for nm in ('data', 'grad', 'm', 'v'):
                    row['%s.%s' % (kind, nm)] = [float(getattr(pool, nm).double().abs().sum())]
            if it == int(os.environ.get('CG_DIAG_DUMP_IT', '-1')):
                row['_dump'] = {'%s/%s/%d/%s' % (d, kind, i, k): p._cg_grad.detach().cpu().clone()
                                for d in tr._dirs for kind in ('gen',) for i, net in enumerate(tr._nets(kind, d))
                                for k, p in net.named_parameters() if getattr(p, '_cg_grad', None) is not None}
        rows.append(row)
    w = {}
    for d in tr._dirs:
        for kind in ('gen', 'dis') + (('disc',) if tr.do_dis_council else ()):
            for i, net in enumerate(tr._nets(kind, d)):
                for k, v in net.state_dict().items():
                    w[(d, kind, i, k)] = v.detach().cpu().clone()
    steps = [list(o._steps) for o in tr.gen_opt_s]
    ring = {d: list(tr._ring_pos[d]) for d in tr._dirs}
    captured = tr.__dict__.get('_n_captures', 0)      # captures over the run (a new key of a kind evicts the old graph)
    assert sum(1 for s in tr._segs.values() if s.graph is not None) <= 8       # at most CG_GRAPH_KEEP = 2 resident captures per kind
    if expect_graph_mode_after is not None:
        assert tr._graph_mode == expect_graph_mode_after
    del tr
    return rows, w, steps, ring, captured


@pytest.mark.parametrize("case", ["m2f_c4_two_groups", "m2f_c2_one_by_one", "anime_c2_b2a", "bidir_c2"])
def test_graph_replay_is_bit_identical_to_eager(cga, case):
    if case == "m2f_c4_two_groups":        # council 4 as two launches of two members: member streams inside the capture
        cfg, gm = _tiny("male2female_council_folder.yaml", 4, 2), 2
    elif case == "m2f_c2_one_by_one":      # one member per launch (what a rank with ONE member runs): the single-member Adam path
        cfg, gm = _tiny("male2female_council_folder.yaml", 2, 2), 1
    elif case == "anime_c2_b2a":
        cfg, gm = _tiny("anime2face_council_folder.yaml", 2, 2), None
    else:
        cfg, gm = _tiny("male2female_council_folder.yaml", 2, 1), None
        cfg['do_b2a'] = True
    try:
        e_rows, e_w, e_steps, e_ring, e_cap = _run(cga, cfg, False, 5, 64, gm)
        g_rows, g_w, g_steps, g_ring, g_cap = _run(cga, cfg, True, 5, 64, gm)
    finally:
        cga.ops.X3_FORWARD = cga.ops.X3_BACKWARD = cga.ops.X3_DYNAMIC_INPUT = True
    assert e_cap == 0 and g_cap >= 4, (e_cap, g_cap)          # dis, disc1, disc2, gen
    assert e_rows == g_rows, [(a, b) for a, b in zip(e_rows, g_rows) if a != b][:2]
    assert e_steps == g_steps and e_ring == g_ring
    for k in e_w:
        assert torch.equal(e_w[k], g_w[k]), k


@pytest.mark.parametrize("fail_at", [1, 2, 4])
def test_a_failed_capture_falls_back_to_eager(cga, fail_at, monkeypatch):
    """If a segment cannot be captured (here: torch.cuda.graph made to fail on its fail_at-th use -- the first segment, one in
    the middle of an iteration, the last one), the trainer warns, leaves graph mode and runs that update and everything
    after it eagerly: the numbers are those of the eager run, nothing is applied twice or skipped."""
    cfg = _tiny("male2female_council_folder.yaml", 2, 2)
    real, uses = torch.cuda.graph, [0]

    class Flaky(real):
        def __enter__(self):
            uses[0] += 1
            if uses[0] == fail_at:
                raise RuntimeError("injected: operation not permitted when stream is capturing")
            return super().__enter__()
    try:
        e_rows, e_w, e_steps, e_ring, _ = _run(cga, cfg, False, 4, 64)
        monkeypatch.setattr(torch.cuda, "graph", Flaky)
        with pytest.warns(UserWarning, match="hipGraph capture of segment"):
            g_rows, g_w, g_steps, g_ring, _ = _run(cga, cfg, True, 4, 64, expect_graph_mode_after=False)
    finally:
        cga.ops.X3_FORWARD = cga.ops.X3_BACKWARD = cga.ops.X3_DYNAMIC_INPUT = True
    assert uses[0] == fail_at
    assert e_rows == g_rows
    assert e_steps == g_steps and e_ring == g_ring
    for k in e_w:
        assert torch.equal(e_w[k], g_w[k]), k


def test_graph_mode_two_discriminator_updates_on_different_batches(cga):
    """dis.numberOf_dis_relative_iteration > 1 (train.py:241-246): two discriminator updates in a row, each on its own batch.
    In graph mode both batches land in the same static buffer; the content-code cache is keyed on the buffer's version, so
    the second update encodes ITS batch (it used to reuse the first batch's codes).  One member per launch -- the
    configuration of a rank that holds one member -- and bit-identical to the eager run."""
    cfg = _tiny("male2female_council_folder.yaml", 2, 2)
    try:
        e_rows, e_w, e_steps, _, _ = _run(cga, cfg, False, 4, 64, 1, dis_twice=True)
        g_rows, g_w, g_steps, _, g_cap = _run(cga, cfg, True, 4, 64, 1, dis_twice=True)
    finally:
        cga.ops.X3_FORWARD = cga.ops.X3_BACKWARD = cga.ops.X3_DYNAMIC_INPUT = True
    assert g_cap >= 4
    assert e_rows == g_rows, [(a, b) for a, b in zip(e_rows, g_rows) if a != b][:2]
    assert e_steps == g_steps
    for k in e_w:
        assert torch.equal(e_w[k], g_w[k]), k


def test_graph_mode_recaptures_when_the_schedule_changes(cga):
    """Crossing focus_loss_start_at_iter / council_start_at_iter changes which kernels an update launches: a new key, a
    new warm-up + capture; the numbers stay those of the eager run."""
    cfg = _tiny("male2female_council_folder.yaml", 2, 2)
    cfg['focus_loss']['focus_loss_start_at_iter'] = 60002
    try:
        e_rows, e_w, _, _, _ = _run(cga, cfg, False, 6, 64)
        g_rows, g_w, _, _, g_cap = _run(cga, cfg, True, 6, 64)
    finally:
        cga.ops.X3_FORWARD = cga.ops.X3_BACKWARD = cga.ops.X3_DYNAMIC_INPUT = True
    assert g_cap >= 5                                        # the generator update was captured under both schedules
    assert e_rows == g_rows
    for k in e_w:
        assert torch.equal(e_w[k], g_w[k]), k


def full_width_eager_vs_graph(cga, iters_warm=4, iters_timed=5):
    """male2female 256x256, council 4, batch 1 at the shipped widths, eager then replayed: returns
    {graph: (median host enqueue ms, median total ms, generator losses)}.  Shared with tests/test_gpu_perf.py."""
    cfg = yaml.safe_load(open(os.path.join(CONFIGS, "male2female_council_folder.yaml")))
    cfg['council']['council_size'] = 4
    cfg['batch_size'] = 1
    cfg['iteration'] = 60000
    x_a, x_b = O.synthetic_batch(1, 256)
    x_a, x_b = x_a.cuda(), x_b.cuda()
    res = {}
    try:
        for graph in (False, True):
            c = copy.deepcopy(cfg)
            c['cg_graph'] = '1' if graph else '0'
            O.seed_all(3)
            tr = cga.Council_Trainer(c, 'cuda:0')
            tr.cuda('cuda:0')

            def step():
                tr.dis_update(x_a, x_b, c); tr.dis_council_update(x_a, x_b, c); tr.gen_update(x_a, x_b, c, 60000)
            for _ in range(iters_warm):
                step()
            torch.cuda.synchronize()
            host, total = [], []
            for _ in range(iters_timed):
                t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
                host.append(1e3 * (t1 - t0)); total.append(1e3 * (t2 - t0))
            res[graph] = (float(np.median(host)), float(np.median(total)), [float(v) for v in tr.loss_gen_total_s])
            del tr
    finally:
        cga.ops.X3_FORWARD = cga.ops.X3_BACKWARD = cga.ops.X3_DYNAMIC_INPUT = True
    return res


def test_graph_mode_full_width(cga):
    """At full width (male2female 256x256, council 4, batch 1) nine replayed iterations end on the same generator losses, bit for
    bit, as nine eager ones.  The wall-clock side of this run (host enqueue cost, total time) is asserted in
    tests/test_gpu_perf.py under the `perf` marker -- a timing assertion must not be able to stop the parity suite."""
    res = full_width_eager_vs_graph(cga)
    print("\n[graph mode] host enqueue per iteration: eager %.1f ms (GPU done after %.1f ms), graph %.1f ms (GPU done after %.1f ms)"
          % (res[False][0], res[False][1], res[True][0], res[True][1]))
    assert res[True][2] == res[False][2]


def test_captured_launches_own_their_workspaces(cga):
    """Round 4's driver abort (`Memory access fault ... Write access to a read-only page` after four full-width iterations): the
    scratch cache is keyed by the raw stream handle, torch recycles 32 handles, so a capture could bake in a buffer some EARLIER
    trainer had allocated eagerly on the same handle; the cache later replaced it, and the next capture's torch.cuda.empty_cache()
    unmapped it under the graph.  Inside hip.capture_workspaces(table) a capture takes every buffer from its own table (private
    pool memory that lives as long as the graph's owner keeps the table), never from the process-wide cache."""
    hip = cga.hip
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eager = hip.workspace(1 << 20, slot=7)                        # what an earlier trainer left behind on this handle
        assert hip.workspace(1 << 19, slot=7) is eager
    torch.cuda.synchronize()
    table, g = {}, torch.cuda.CUDAGraph()
    x = torch.zeros(1 << 18, device='cuda')
    with hip.capture_workspaces(table), torch.cuda.graph(g, stream=st, capture_error_mode='thread_local'):
        a = hip.workspace(1 << 19, slot=7)
        cga.ops.fill_(a.view(torch.float32)[:1 << 17], 3.0)
        b = hip.workspace(3 << 20, slot=7)                            # outgrown inside the same capture: `a` must stay alive
        cga.ops.fill_(b.view(torch.float32)[:1 << 18], 5.0)
        x.copy_(b.view(torch.float32)[:1 << 18])
    assert a.data_ptr() != eager.data_ptr() and b.data_ptr() != eager.data_ptr()
    assert any(t is a for t in table.get('retired', []))
    with torch.cuda.stream(st):
        assert hip.workspace(1 << 19, slot=7) is eager                # outside the capture: the process-wide cache, untouched
        hip.workspace(8 << 20, slot=7)                                # ... which may now grow and drop its old buffer
    del eager
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                                          # what the NEXT capture's __enter__ does
    g.replay()
    torch.cuda.synchronize()
    assert float(x.min()) == 5.0 and float(a.view(torch.float32)[0]) == 3.0
    # and without a table an allocation inside a capture is refused instead of baked in
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(cga.hip.HipError, match="capture_workspaces"):
        with torch.cuda.graph(g2, stream=st, capture_error_mode='thread_local'):
            hip.workspace(64 << 20, slot=7)
