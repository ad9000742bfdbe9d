#!/bin/bash
# round-2 GPU call 10 (~3.5 GPU-minutes left): the defaults changed by call 9's evidence (256x128 weight-gradient tile where
# it wins, thin weight-gradient kernel, 128x32 tile for the thin output layers) under the operator tests, the reference
# fixtures, a bench line and the fast whole-iteration parity tests -- most informative first, each with its own timeout.
set -u
O=gpurun_out/r02_j
mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout -k 5 45 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider < /dev/null > $O/t1_ops.log 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t1_ops.log; tail -2 $O/t1_ops.log
timeout -k 5 110 python -m pytest tests/test_gpu_golden.py -m gpu -q -p no:cacheprovider < /dev/null > $O/t2_golden.log 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t2_golden.log; tail -2 $O/t2_golden.log
J='"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*'
timeout -k 5 60 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-exact-fp32 --shape-report $O/shapes.txt < /dev/null > $O/bench.json 2> $O/bench.err
echo "new default: $(grep -o "$J" $O/bench.json) t=$(( $(date +%s) - t0 ))"
timeout -k 5 60 python -m pytest tests/test_gpu_parity_full.py -m "gpu and not slow" -q -p no:cacheprovider < /dev/null > $O/t3_parity.log 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t3_parity.log; tail -2 $O/t3_parity.log
