#!/bin/bash
# Round-6 GPU sessions:  gpurun --timeout T -- 'bash tools/gpu_r6.sh <out name> <stage> [<stage> ...]'
# every stage writes its own log under gpurun_out/<out name>/ and never stops the rest
set -u
O=gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp PYTHONFAULTHANDLER=1
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
run() { local name=$1 to=$2; shift 2; timeout -k 5 $to "$@" < /dev/null > $O/$name.log 2>&1; echo "== $name rc=$? t=$(el)"; }
pmc() {  # pmc <tag> <kernel substring> <cfg> <shape spec or ""> [tool]
  local tag=$1 kern=$2 cfg=$3 shape=$4 tool=${5:-}
  PMC=$cfg PMC_KERNEL=$kern PMC_SHAPE="$shape" PMC_TOOL="$tool" timeout -k 5 300 bash tools/prof_bench.sh ${O#gpurun_out/}/pmc_$tag < /dev/null > $O/pmc_$tag.log 2>&1
  echo "== pmc_$tag t=$(el)"; cat $O/pmc_$tag/summary.txt 2>/dev/null | cut -c1-600
}
for st in "$@"; do
  case $st in
    tiles)   run tiles 300 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "tile_configuration" ;;
    ops)     run ops 400 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider ;;
    ab4w)    EXTRA_SHAPES="res16,16,64,64,256,256,3,1,1,0;res32,32,64,64,256,256,3,1,1,0;dc128_256,16,128,128,128,256,4,2,1,0" run ab4w 300 python tools/ab_x3.py 16,50,51,52 0; cat $O/ab4w.log | cut -c1-200 ;;
    abr64)   EXTRA_SHAPES="c64_3x3_256,16,256,256,64,64,3,1,1,0;c64_1x1_256,16,256,256,64,64,1,1,0,0;up128x4_64_128,16,128,128,128,64,2,1,0,0;c64_3x3_128_b8,8,128,128,64,64,3,1,1,0;c64_3x3_256_b4,4,256,256,64,64,3,1,1,0" run abr64 300 python tools/ab_x3.py 2,26,60,61,62,63,67 -; cat $O/abr64.log | cut -c1-260 ;;
    abr128)  EXTRA_SHAPES="c128_3x3_128,16,128,128,128,128,3,1,1,0;dg256x4_128_64,16,64,64,256,128,2,1,0,0;c64_128_4x4s2_256,16,256,256,64,128,4,2,1,0;c128_3x3_128_b4,4,128,128,128,128,3,1,1,0;d64_128_4x4s2_128_b32,32,128,128,64,128,4,2,1,0" run abr128 300 python tools/ab_x3.py 13,1,17,64,65,66,68 -; cat $O/abr128.log | cut -c1-260 ;;
    abk128)  EXTRA_SHAPES="c128_3x3_128,16,128,128,128,128,3,1,1,0;dg256x4_128_64,16,64,64,256,128,2,1,0,0;c64_128_4x4s2_256,16,256,256,64,128,4,2,1,0;d64_128_4x4s2_128_b32,32,128,128,64,128,4,2,1,0;dg512x4_256_32,16,32,32,512,256,2,1,0,0" run abk128 300 python tools/ab_x3.py 13,23,1,24 -; cat $O/abk128.log | cut -c1-200 ;;
    abc64)   EXTRA_SHAPES="c64_3x3_256,16,256,256,64,64,3,1,1,0;c64_1x1_256,16,256,256,64,64,1,1,0,0;up128x4_64_128,16,128,128,128,64,2,1,0,0;c64_3x3_256_b4,4,256,256,64,64,3,1,1,0" run abc64 300 python tools/ab_x3.py 2,25,69,70,26,27 -; cat $O/abc64.log | cut -c1-230 ;;
    exact)   run exact_on 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-configs --no-live-pmc; CG_FP32_CHUNKED_SUM=0 run exact_off 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-configs --no-live-pmc; for f in exact_on exact_off; do python - $O/$f.log <<'PY'
import json,sys
p=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]; e=p['exact_fp32']
print(sys.argv[1].split('/')[-1], 'split', p['ms_per_step'], 'exact_fp32', e['ms_per_step'], e['step_frac'], e.get('kernel'), e.get('kernel_avg_us'), e.get('all_conv_kernels_tflops'))
PY
done ;;
    stat0)   CG_FP32_CHUNKED_SUM=0 run stat0 600 python -m pytest tests/test_gpu_parity_full.py -q -s -p no:cacheprovider -k generator_gradient_statistic; grep -A16 "generator-gradient l2-rel\|quiet level" $O/stat0.log | cut -c1-200; tail -3 $O/stat0.log ;;
    stat32)  run stat32 600 python -m pytest tests/test_gpu_parity_full.py -x -q -s -p no:cacheprovider -k "statistic_exact_fp32"; grep -A20 "generator-gradient l2-rel" $O/stat32.log | cut -c1-110; tail -3 $O/stat32.log ;;
    abgroup) bash tools/ab_step.sh ${O#gpurun_out/}/abgroup "new" "nogroup:CG_WGRAD_XCD_GROUP=0" 2>&1 | tee $O/abgroup.log ;;
    abbound) bash tools/ab_step.sh ${O#gpurun_out/}/abbound "new" "bounded:CG_BOUNDED_SPLIT=1" "fused:CG_BOUNDED_SPLIT=1 CG_FUSED_ACT_BWD=1" 2>&1 | tee $O/abbound.log ;;
    ab5)     EXTRA_SHAPES="c128_3x3_128,16,128,128,128,128,3,1,1,0;dg256x4_128_64,16,64,64,256,128,2,1,0,0;c64_128_4x4s2_256,16,256,256,64,128,4,2,1,0;d64_128_4x4s2_128_b32,32,128,128,64,128,4,2,1,0" run ab5 300 python tools/ab_x3.py 13,5,1,0 -; cat $O/ab5.log | cut -c1-200 ;;

    thin)    run thin_test 300 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "thin"; tail -3 $O/thin_test.log; bash tools/ab_step.sh ${O#gpurun_out/}/abthin "new" "fp32thin:CG_THIN_X3=0" 2>&1 | tee $O/abthin.log; run thin_shapes 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-live-pmc --shape-report $O/thin_shapes.txt; grep "^f8" $O/thin_shapes.txt | cut -c1-130 ;;
    probe4w) run probe4w 200 python tools/probe_x3w_stalls.py 16 53 4; tail -40 $O/probe4w.log | cut -c1-160 ;;
    abwg)    run abwg 300 python tools/ab_wgrad.py; cat $O/abwg.log | cut -c1-200 ;;
    pmc1)    pmc x3_128x64 conv_fwd_x3_kernel 2 "c64,16,256,256,64,64,3,1,1,0" ;;
    pmc2)    pmc x3_256x128 conv_fwd_x3_kernel 13 "dg256x4_128,16,64,64,256,128,2,1,0,0" ;;
    pmc3)    pmc wg_128x128 conv_wgrad_x3t 1 "" "tools/ab_wgrad.py --launch 2 8" ;;
    pmc4)    pmc wg_64x128 conv_wgrad_x3t 1 "" "tools/ab_wgrad.py --launch 1 8" ;;
    pmc5)    pmc wg_wide conv_wgrad_x3tw 1 "" "tools/ab_wgrad.py --launch 0 8" ;;
    pmc16)   AB_ACT=0 pmc x3w conv_fwd_x3w 16 ""; python tools/pmc_record.py $O/pmc_x3w/summary.txt > $O/pmc_record.json 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json ;;
    bench)   run bench 400 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-exact-fp32 --no-other-configs --no-live-pmc --shape-report $O/conv_shapes.txt; grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": [0-9]*, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' $O/bench.log ;;
    bench_full) run bench_full 900 python bench.py --shape-report $O/conv_shapes_full.txt; tail -c 3000 $O/bench_full.log ;;
    prof)    STEPS=5 BENCH_ARGS="--no-kernel-profile --no-exact-fp32 --no-other-configs" timeout -k 5 400 bash tools/prof_bench.sh ${O#gpurun_out/}/prof < /dev/null > $O/prof.log 2>&1; echo "== prof t=$(el)"; head -50 $O/prof/alone.txt | cut -c1-140 ;;
    tests_all) run tests_all 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=15; tail -25 $O/tests_all.log ;;
    stat)    run stat 600 python -m pytest tests/test_gpu_parity_full.py -q -s -p no:cacheprovider -k generator_gradient_statistic; grep -A16 "generator-gradient l2-rel\|quiet level" $O/stat.log | cut -c1-200; tail -3 $O/stat.log ;;
    skip)    run skip_test 300 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "resblock_skip or instance_norm or focus"; tail -3 $O/skip_test.log; bash tools/ab_step.sh ${O#gpurun_out/}/abskip "new" "engine_adds:CG_SKIP_FUSE=0 CG_ADAIN_FORK=0" 2>&1 | tee $O/abskip.log; run aten 300 python tools/aten_rows.py; grep -v Warning $O/aten.log | tail -60 | cut -c1-200 ;;
    one_member) run one_member 400 python tools/one_member_rank.py; tail -20 $O/one_member.log | cut -c1-200 ;;
    py:*)    a=${st#py:}; run py_$(echo $a | tr -c 'A-Za-z0-9' '_' | cut -c1-40) 400 python ${a//,/ } ;;
    *) echo "unknown stage $st" ;;
  esac
done
echo "session done t=$(el)"
