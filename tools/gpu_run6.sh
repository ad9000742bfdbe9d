#!/bin/bash
# round-2 GPU call 6: update-level stream overlap A/B, 16-byte split stores, staggered wide kernel A/B
set -u
O=gpurun_out/r02_f
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --maxfail=12 > $O/pytest_ops.log 2>&1
echo "ops rc=$?" >> $O/pytest_ops.log; tail -4 $O/pytest_ops.log
timeout 600 python -m pytest tests/test_gpu_golden.py -q -p no:cacheprovider -s -k "two_iterations or full_width or content_cache or split_precision_decoder or save_resume" > $O/pytest_golden.log 2>&1
echo "golden rc=$?" >> $O/pytest_golden.log; tail -5 $O/pytest_golden.log
timeout 600 python -m pytest tests/test_gpu_parity_full.py -m "gpu and not slow" -q -p no:cacheprovider -s -k "iteration or resume or adam or large_weights" > $O/pytest_parity.log 2>&1
echo "parity rc=$?" >> $O/pytest_parity.log; tail -5 $O/pytest_parity.log
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-exact-fp32 --shape-report $O/shapes_overlap.txt > $O/bench_overlap.json 2> $O/bench_overlap.err
tail -c 250 $O/bench_overlap.json | head -c 250; echo; tail -2 $O/bench_overlap.err
CG_OVERLAP_UPDATES=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile > $O/bench_nooverlap.json 2> $O/bench_nooverlap.err
tail -c 250 $O/bench_nooverlap.json | head -c 250; echo
EXTRA_SHAPES="res16,16,64,64,256,256,3,1,1,0;up1,16,64,64,256,128,3,1,1,1;dec128,16,128,128,128,128,3,1,1,0;d3,32,64,64,128,256,4,2,1,0;dc3,64,128,128,128,256,4,2,1,0;res32,32,64,64,256,256,3,1,1,0" timeout 300 python tools/ab_x3.py 1,13,16,18,17,19 0 > $O/ab_x3.txt 2>&1
tail -9 $O/ab_x3.txt
