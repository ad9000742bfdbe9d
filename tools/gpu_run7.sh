#!/bin/bash
# round-2 GPU call 7: companion weight-gradient stream, wide tile in the step, overlap fix
set -u
O=gpurun_out/r02_g
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --maxfail=12 > $O/pytest_ops.log 2>&1
echo "ops rc=$?" >> $O/pytest_ops.log; tail -3 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_golden.py -q -p no:cacheprovider -s > $O/pytest_golden.log 2>&1
echo "golden rc=$?" >> $O/pytest_golden.log; tail -5 $O/pytest_golden.log
timeout 600 python -m pytest tests/test_gpu_parity_full.py -m "gpu and not slow" -q -p no:cacheprovider -s > $O/pytest_parity.log 2>&1
echo "parity rc=$?" >> $O/pytest_parity.log; tail -5 $O/pytest_parity.log
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-exact-fp32 --shape-report $O/shapes.txt > $O/bench.json 2> $O/bench.err
python tools/json_value.py $O/bench.json value ms_per_step 2>/dev/null || tail -c 200 $O/bench.json; tail -2 $O/bench.err
CG_WGRAD_STREAM=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile > $O/bench_nowgs.json 2> $O/bench_nowgs.err
tail -c 2000 $O/bench_nowgs.json | grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": 12, "warmup": 3, "ms_per_step": [0-9.]*'
CG_X3_WIDE=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile > $O/bench_nowide.json 2> $O/bench_nowide.err
grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": 12, "warmup": 3, "ms_per_step": [0-9.]*' $O/bench_nowide.json
grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": 12, "warmup": 3, "ms_per_step": [0-9.]*' $O/bench.json
