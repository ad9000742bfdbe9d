#!/bin/bash
# usage: bash tools/gpu_round5.sh <out name> <stage> [<stage> ...]   -- stages of a round-5 GPU session, each with its own log
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export PYTHONFAULTHANDLER=1
run () { local name=$1; shift; local t0=$SECONDS; ( "$@" ) > $OUT/$name.log 2>&1; local rc=$?; echo "rc=$rc t=$((SECONDS-t0))" >> $OUT/$name.log; echo "== $name: rc=$rc t=$((SECONDS-t0))"; }
for stage in "$@"; do
  case $stage in
    tests_all) run tests_all timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=25 ;;
    seq)       run seq timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_graph.py -x -q -s -p no:cacheprovider ;;
    ops)       run ops timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider ;;
    graph)     run graph timeout 300 python -m pytest tests/test_gpu_graph.py -x -q -s -p no:cacheprovider ;;
    ab)        AB_ACT=${AB_ACT:-1} EXTRA_SHAPES="res16,16,64,64,256,256,3,1,1,0;res32,32,64,64,256,256,3,1,1,0;dc128_256,16,128,128,128,256,4,2,1,0" run ab timeout 300 python tools/ab_x3.py ${AB_CFGS:-16,48,46,47} 0 ;;
    bench)     run bench timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-exact-fp32 ;;
    bench_full) run bench_full timeout 900 python bench.py --shape-report $OUT/conv_shapes.txt ;;
    ratio)     run ratio env CG_LONG_ORACLE=1 timeout 1200 python -m pytest tests/test_gpu_parity_full.py -x -q -s -k generator_gradient_ratio -p no:cacheprovider ;;
    perf)      run perf timeout 300 python -m pytest tests/ -x -q -s -m perf -p no:cacheprovider ;;
    pmc)       AB_ACT=0 PMC=16 bash tools/prof_bench.sh $(basename $OUT)_pmc > $OUT/pmc.log 2>&1; python tools/pmc_record.py gpurun_out/$(basename $OUT)_pmc/summary.txt > $OUT/pmc_record.json 2>&1; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; echo "== pmc: $(grep -c launches gpurun_out/$(basename $OUT)_pmc/summary.txt) passes" ;;
    prof)      STEPS=5 BENCH_ARGS="--no-kernel-profile --no-exact-fp32 --no-other-configs" bash tools/prof_bench.sh $(basename $OUT)_prof > $OUT/prof.log 2>&1; echo "== prof: $(tail -n 1 $OUT/prof.log | cut -c1-120)" ;;
    *) echo "unknown stage $stage" ;;
  esac
done
