"""Effective HBM bandwidth of the bandwidth-bound passes of the step on two tensor sizes (bytes moved = what the pass must read and
write; time = best of 5 x 10 launches between HIP events).  usage: python tools/bw_passes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import council_gan_amd as cga  # noqa: E402,F401
from council_gan_amd import hip, ops  # noqa: E402

CL = torch.channels_last


def timeit(fn, reps=10, rounds=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e-3


def main():
    hip.load()
    for shape in ((16, 256, 64, 64), (16, 64, 256, 256), (32, 64, 256, 256)):
        N, C, H, W = shape
        x = torch.randn(shape, device="cuda").contiguous(memory_format=CL)
        dy = torch.randn(shape, device="cuda").contiguous(memory_format=CL)
        n = x.numel()
        mb = n * 4 / 1e6
        rows = []
        with torch.no_grad():
            t = timeit(lambda: ops.split_f16_dynamic(x))
            rows.append(("split_f16_dynamic (amax pass + split: 2 reads, 1 write)", 3 * mb, t))
            st = ops.split_f16_dynamic(x)
            t = timeit(lambda: ops.split_f16_dynamic(x, (st.state, 0)) if False else ops.split_f16(x))
            rows.append(("split_f16 static (1 read, 1 write)", 2 * mb, t))
            t = timeit(lambda: ops.act_bwd_split(dy, x, 2, False))
            rows.append(("act_bwd_split no amax given (amax pass 2 reads + 2 reads, 1 write)", 5 * mb, t))
            t = timeit(lambda: ops.instnorm_split(x, None, 0, 0, want_f32=False))
            rows.append(("instnorm stats + apply_split (2 reads, 1 write)", 3 * mb, t))
            t = timeit(lambda: ops.instnorm_split(x, None, 0, 0, residual=dy, want_f32=True))
            rows.append(("instnorm stats + apply_split + residual + f32 (3 reads, 2 writes)", 5 * mb, t))
            y = torch.empty_like(x)
            t = timeit(lambda: y.copy_(x))
            rows.append(("torch copy_ (1 read, 1 write) -- reference", 2 * mb, t))
        print("shape %s: %.0f MB per fp32 tensor" % (shape, mb))
        for name, mbytes, t in rows:
            print("   %-75s %8.1f us  %6.2f TB/s" % (name, t * 1e6, mbytes / 1e6 / t))


if __name__ == "__main__":
    main()
