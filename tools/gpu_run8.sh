#!/bin/bash
# round-2 GPU call 8 (final): the default configuration -- previously failing parity tests, bench, rocprof kernel stats
set -u
O=gpurun_out/r02_h
mkdir -p $O
export TMPDIR=/tmp
timeout -k 5 400 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity_full.py tests/test_gpu_world.py -m "gpu and not slow" -q -p no:cacheprovider -k "two_iterations or full_width or iteration_vs_oracle or 2-4 or content_cache" < /dev/null > $O/pytest_sel.log 2>&1
echo "sel rc=$?" >> $O/pytest_sel.log; tail -6 $O/pytest_sel.log
timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --shape-report $O/shapes.txt < /dev/null > $O/bench.json 2> $O/bench.err
grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/bench.json; grep -o '"exact_fp32": {"value": [0-9.]*' $O/bench.json; tail -2 $O/bench.err
CG_X3_WIDE=0 timeout -k 5 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile < /dev/null > $O/bench_nowide.json 2> $O/bench_nowide.err
grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/bench_nowide.json
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o fin -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile < /dev/null > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $O/prof/fin_results.db 70 > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt
