#!/bin/bash
# One parameterized GPU session script (replaces the per-call gpu_run*.sh files):
#     gpurun --timeout T -- 'bash tools/gpu_session.sh <outdir under gpurun_out> <stage> [<stage> ...]'
# Every stage has its own timeout and writes its log under gpurun_out/<outdir>/; a stage that fails does not stop the rest.
# Stages take optional arguments after a colon, e.g.  bench:--cfg,5   ab_x3:16,22:12   env:CG_X3_KORDER=1 (sticky for later stages)
set -u
O=gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
J='"value": [0-9.]*, "unit": "images/sec", "n_gpus": [0-9]*, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*'
el() { echo $(( $(date +%s) - T0 )); }
pt() {  # pt <log name> <timeout> <pytest args...>
  local log=$O/$1.log to=$2; shift 2
  timeout -k 5 $to python -m pytest "$@" -q -p no:cacheprovider < /dev/null > $log 2>&1
  echo "rc=$? t=$(el)" >> $log; echo "[$log] $(tail -3 $log | tr '\n' ' ' | cut -c1-300)"
}
n=0
for st in "$@"; do
  n=$((n + 1)); name=${st%%:*}; arg=""; [ "$st" != "$name" ] && arg=${st#*:}
  case $name in
    env) export "$arg"; echo "[env] $arg" ;;
    tests_targets) pt ${n}_targets 420 tests/test_gpu_parity_targets.py -m gpu -s ;;
    tests_ops) pt ${n}_ops 120 tests/test_gpu_ops.py -m gpu ;;
    tests_golden) pt ${n}_golden 240 tests/test_gpu_golden.py -m gpu ;;
    tests_full_fast) pt ${n}_parity_fast 240 tests/test_gpu_parity_full.py -m "gpu and not slow" -s ;;
    tests_slow) pt ${n}_parity_slow 420 tests/test_gpu_parity_full.py -m "gpu and slow" -s ;;
    tests_world) pt ${n}_world 300 tests/test_gpu_world.py -m gpu ;;
    tests_all) pt ${n}_all 900 tests -m gpu ;;
    tests_k) pt ${n}_k 300 tests -m gpu -s -k "$arg" ;;
    bench)  # bench[:comma-separated extra args]
      timeout -k 5 240 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-exact-fp32 --no-other-configs --shape-report $O/${n}_shapes.txt ${arg//,/ } < /dev/null > $O/${n}_bench.json 2> $O/${n}_bench.err
      echo "[bench ${arg}] $(grep -o "$J" $O/${n}_bench.json) t=$(el)" ;;
    bench_full)  # the driver's command: every leg (exact fp32, cpu baseline)
      timeout -k 5 400 python bench.py --steps 20 --warmup 5 ${arg//,/ } < /dev/null > $O/${n}_bench_full.json 2> $O/${n}_bench_full.err
      echo "[bench_full ${arg}] $(grep -o "$J" $O/${n}_bench_full.json) t=$(el)" ;;
    prof)  # rocprofv3 --kernel-trace --stats of the default bench command (tools/prof_bench.sh)
      STEPS=5 BENCH_ARGS="--no-exact-fp32 --no-other-configs ${arg//,/ }" timeout -k 5 240 bash tools/prof_bench.sh ${O#gpurun_out/}/${n}_prof < /dev/null > $O/${n}_prof.log 2>&1
      echo "[prof] $(grep -o "$J" $O/${n}_prof/bench.json) t=$(el)" ;;
    ab_x3)  # ab_x3:<cfgs>[:reps]   (EXTRA_SHAPES from the environment)
      cfgs=${arg%%:*}
      timeout -k 5 180 python tools/ab_x3.py $cfgs 12 < /dev/null > $O/${n}_ab_x3.txt 2>&1; cut -c1-260 $O/${n}_ab_x3.txt; echo "t=$(el)" ;;
    pmc_x3w)  # pmc_x3w:<cfg>
      PMC=$arg timeout -k 5 170 bash tools/prof_bench.sh ${O#gpurun_out/}/${n}_pmc_cfg$arg < /dev/null > $O/${n}_pmc.log 2>&1; tail -4 $O/${n}_pmc.log | cut -c1-400; echo "t=$(el)" ;;
    py)  # py:<script>[,args]   any tool under tools/
      timeout -k 5 240 python ${arg//,/ } < /dev/null > $O/${n}_py.txt 2>&1; tail -40 $O/${n}_py.txt | cut -c1-260; echo "t=$(el)" ;;
    *) echo "unknown stage $st" ;;
  esac
done
echo "session done t=$(el)"
