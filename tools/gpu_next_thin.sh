#!/bin/bash
# Prepared for the next GPU session (not yet run): the whole-iteration parity tests with the thin FORWARD kernel on
# (CG_FWD_THIN=1 -- it changes first-layer activations in the last bit, so the chaotic generator-gradient comparison of
# DESIGN.md section 3 has to be re-run before it becomes the default), then the same-box bench A/B.
set -u
O=gpurun_out/thin_fwd
mkdir -p $O
export TMPDIR=/tmp
CG_FWD_THIN=1 timeout -k 5 330 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_golden.py -m gpu -q -p no:cacheprovider --durations=5 < /dev/null > $O/parity.log 2>&1
echo "rc=$?" >> $O/parity.log; tail -4 $O/parity.log
J='"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*'
timeout -k 5 100 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile < /dev/null > $O/bench_default.json 2> $O/bench_default.err
echo "default:  $(grep -o "$J" $O/bench_default.json)"
CG_FWD_THIN=1 timeout -k 5 100 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile < /dev/null > $O/bench_thin.json 2> $O/bench_thin.err
echo "thin fwd: $(grep -o "$J" $O/bench_thin.json)"
