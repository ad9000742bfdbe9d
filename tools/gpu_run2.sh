#!/bin/bash
# round-2 GPU call 2: member-batched execution -- op tests, trainer tests, gradient-error diagnostics, quick bench A/B
set -u
O=gpurun_out/r02_b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -x --maxfail=8 > $O/pytest_ops.log 2>&1
echo "ops rc=$?" >> $O/pytest_ops.log; tail -4 $O/pytest_ops.log
timeout 1200 python -m pytest tests -m "gpu and not slow" -q -p no:cacheprovider --deselect tests/test_gpu_ops.py -s > $O/pytest_rest.log 2>&1
echo "rest rc=$?" >> $O/pytest_rest.log; tail -12 $O/pytest_rest.log
timeout 300 python tools/diag_gengrad.py anime2face_council_folder.yaml 128 2 1 > $O/diag_anime_split.txt 2>&1
CG_FORWARD_PRECISION=fp32 timeout 300 python tools/diag_gengrad.py anime2face_council_folder.yaml 128 2 1 > $O/diag_anime_fp32.txt 2>&1
timeout 300 python tools/diag_gengrad.py male2female_council_folder.yaml 64 2 2 > $O/diag_m2f_split.txt 2>&1
CG_GROUP=1 timeout 300 python tools/diag_gengrad.py male2female_council_folder.yaml 64 2 2 > $O/diag_m2f_split_seq.txt 2>&1
head -3 $O/diag_*.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --shape-report $O/shapes_grouped.txt > $O/bench_grouped.json 2> $O/bench_grouped.err
tail -c 400 $O/bench_grouped.json; tail -3 $O/bench_grouped.err
CG_GROUP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile > $O/bench_seq.json 2> $O/bench_seq.err
tail -c 300 $O/bench_seq.json; tail -3 $O/bench_seq.err
