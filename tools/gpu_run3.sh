#!/bin/bash
# round-2 GPU call 3: member-batched execution (fixed), wide LDS-DMA kernel A/B, gradient-error diagnostics
set -u
O=gpurun_out/r02_c
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --maxfail=12 > $O/pytest_ops.log 2>&1
echo "ops rc=$?" >> $O/pytest_ops.log; tail -4 $O/pytest_ops.log
timeout 600 python -m pytest tests/test_gpu_golden.py -q -p no:cacheprovider -s -k "two_iterations or full_width or content_cache or split_precision_decoder or save_resume or sample_layout" > $O/pytest_golden.log 2>&1
echo "golden rc=$?" >> $O/pytest_golden.log; tail -6 $O/pytest_golden.log
timeout 600 python -m pytest tests/test_gpu_parity_full.py -m "gpu and not slow" -q -p no:cacheprovider -s > $O/pytest_parity.log 2>&1
echo "parity rc=$?" >> $O/pytest_parity.log; tail -6 $O/pytest_parity.log
timeout 200 python tools/diag_gengrad.py anime2face_council_folder.yaml 128 2 1 > $O/diag_anime_split.txt 2>&1
timeout 200 python tools/diag_gengrad.py male2female_council_folder.yaml 64 2 2 > $O/diag_m2f_split.txt 2>&1
CG_FORWARD_PRECISION=fp32 timeout 200 python tools/diag_gengrad.py male2female_council_folder.yaml 64 2 2 > $O/diag_m2f_fp32.txt 2>&1
CG_FORWARD_PRECISION=fp32 timeout 200 python tools/diag_gengrad.py glasses_council_folder.yaml 128 1 2 > $O/diag_glasses_fp32.txt 2>&1
grep -h "^config\|^member" $O/diag_*.txt
CG_X3_WIDE=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --shape-report $O/shapes_grouped_w0.txt > $O/bench_grouped_w0.json 2> $O/bench_grouped_w0.err
tail -c 300 $O/bench_grouped_w0.json; tail -2 $O/bench_grouped_w0.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --shape-report $O/shapes_grouped_w1.txt > $O/bench_grouped_w1.json 2> $O/bench_grouped_w1.err
tail -c 300 $O/bench_grouped_w1.json; tail -2 $O/bench_grouped_w1.err
EXTRA_SHAPES="res16,16,64,64,256,256,3,1,1,0;up1,16,64,64,256,128,3,1,1,1;dec128,16,128,128,128,128,3,1,1,0;d3,32,64,64,128,256,4,2,1,0;d4,32,32,32,256,512,4,2,1,0;dc2,64,256,256,64,128,4,2,1,0;dc3,64,128,128,128,256,4,2,1,0" timeout 300 python tools/ab_x3.py 1,13,5,16,17 0 > $O/ab_x3.txt 2>&1
cat $O/ab_x3.txt | tail -12
timeout 200 python -m pytest tests/test_gpu_world.py -q -p no:cacheprovider -k "2-4" > $O/pytest_world.log 2>&1
echo "world rc=$?" >> $O/pytest_world.log; tail -4 $O/pytest_world.log
