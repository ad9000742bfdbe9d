"""What ONE rank of the sharded council does per iteration, measured on one GPU.

bench.py --gpus 8 runs council 8 with one member per rank; the only multi-rank step is the all-gather of the members'
comparison images (trainer_council.py:853-856).  This tool builds rank 0's trainer of a world of `--world` ranks, replaces
that ONE collective by tiling the local images (same shapes, same downstream kernels; the numbers are meaningless, the
timing is not) and times the iteration: GPU time (HIP events), host enqueue time, eager vs hipGraph.  The prediction for
N ranks is  council*batch / (GPU time + exchange)  against bench.py --cfg 5 at N = 1.

  python tools/one_member_rank.py [--world 8] [--council 8] [--steps 20] [--graph 0|1] [--kernels]
"""
import argparse
import os
import sys
import time

import torch
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--council", type=int, default=8)
    ap.add_argument("--config", default="anime2face_council_folder.yaml")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--graph", default="both", choices=["0", "1", "auto", "both"], help="both = eager, graph replay, then the sharded default 'auto'")
    ap.add_argument("--kernels", action="store_true", help="per-kernel totals of one iteration (torch profiler)")
    ap.add_argument("--shapes", default="", help="write the per-layer-shape conv timing table of one eager iteration here")
    args = ap.parse_args()

    import council_gan_amd as cga
    from council_gan_amd.parallel import CouncilShard
    cga.hip.load()
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", args.config)))
    cfg['council']['council_size'] = args.council
    cfg['batch_size'] = args.batch
    cfg['new_size'] = cfg['crop_image_height'] = cfg['crop_image_width'] = args.size
    cfg['iteration'] = 60000
    x_a, x_b = cga.synthetic_batch(args.batch, args.size)
    x_a, x_b = x_a.cuda(), x_b.cuda()

    for graph in (["0", "1", "auto"] if args.graph == "both" else [args.graph]):
        c = dict(cfg, cg_graph=graph)
        cga.seed_everything(cfg['random_seed'])
        shard = CouncilShard(args.council, rank=0, world_size=args.world)
        per, ranks = shard.per_rank, shard.slice_ranks
        # the one collective of the step, replaced by "every rank sent what I sent"
        shard.exchange_flat = lambda local: local.repeat(ranks, 1, 1, 1).contiguous(memory_format=torch.channels_last)
        tr = cga.Council_Trainer(c, 'cuda:0', shard=shard)
        tr.cuda('cuda:0')
        assert (graph == "auto" or tr._graph_mode == (graph == "1")) and len(shard.local) == per

        def step(it):
            c['iteration'] = 60000 + it
            tr.dis_update(x_a, x_b, c)
            tr.dis_council_update(x_a, x_b, c)
            tr.gen_update(x_a, x_b, c, c['iteration'])

        for it in range(8 if graph == "auto" else 4):      # 'auto' decides on its iterations 3 and 4, then (maybe) captures
            step(it)
        torch.cuda.synchronize()
        if graph == "auto":
            print("cg_graph auto decided:", tr.graph_auto, flush=True)
            graph = "auto->%s" % ("graph" if tr._graph_mode else "eager")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        host = 0.0
        for it in range(args.steps):
            h0 = time.perf_counter()
            step(4 + it)
            host += time.perf_counter() - h0
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / args.steps
        gpu = e0.elapsed_time(e1) / args.steps
        # the host's OWN cost of an iteration: step() against an empty queue (synchronize first), so that nothing in it waits for
        # the GPU -- in the free-running loop above the host is simply held back by the staging ring once it is ~3 iterations ahead
        own = 0.0
        for it in range(5):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            step(4 + args.steps + it)
            own += time.perf_counter() - h0
        torch.cuda.synchronize()
        print("rank 0 of %d (council %d, %d member(s) per rank, batch %d, %dx%d)  graph=%s:  %.2f ms per iteration (events), "
              "wall %.2f ms; host: %.2f ms per iteration of its own work (step() against an empty queue), %.2f ms inside step() when "
              "free-running (the rest is back-pressure from the staging ring)  ->  %d ranks: %.1f images/s before the exchange"
              % (args.world, args.council, per, args.batch, args.size, args.size, graph, gpu, wall, own * 1e3 / 5,
                 host * 1e3 / args.steps, args.world, 1e3 * args.batch / max(gpu, wall)), flush=True)
        if args.shapes and graph == "0":
            side, tr._overlap = tr._overlap, False          # serialised: a launch's events see only that launch
            try:
                cga.hip.prof_enable(True)
                step(4 + args.steps)
                torch.cuda.synchronize()
                cga.hip.prof_collect()
                open(args.shapes, "w").write(cga.hip.prof_report())
            finally:
                cga.hip.prof_enable(False)
                tr._overlap = side
        if args.kernels:
            from torch.profiler import profile, ProfilerActivity
            was = tr._graph_mode
            tr2 = None
            if was:          # graph replays show up as one node: profile an eager twin instead
                c2 = dict(cfg, cg_graph="0")
                tr2 = cga.Council_Trainer(c2, 'cuda:0', shard=shard)
                tr2.cuda('cuda:0')
            t = tr2 or tr
            cc = c2 if tr2 is not None else c

            def step2(it):
                cc['iteration'] = 60000 + it
                t.dis_update(x_a, x_b, cc); t.dis_council_update(x_a, x_b, cc); t.gen_update(x_a, x_b, cc, cc['iteration'])
            for it in range(3):
                step2(it)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for it in range(3):
                    step2(3 + it)
                torch.cuda.synchronize()
            rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)
            tot = sum(r.device_time_total for r in rows)
            print("kernel time per iteration: %.2f ms over %d launches" % (tot / 3e3, sum(r.count for r in rows) // 3))
            for r in rows[:28]:
                print("  %8.3f ms  %5d x  %s" % (r.device_time_total / 3e3, r.count // 3, r.key[:110]))
        del tr


if __name__ == "__main__":
    main()
