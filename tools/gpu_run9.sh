#!/bin/bash
# round-2 GPU call 9 (last of the round, ~10 GPU-minutes left): the whole -m gpu suite on the default build, then the
# opt-in kernels (parity tests, stand-alone A/B, step-level A/B), then the PMC passes of the wide tile.  Every leg has its
# own timeout and writes its own file, most important first.
set -u
O=gpurun_out/r02_i
mkdir -p $O
export TMPDIR=/tmp
NEW="thin_input or 256x128_tile or single_rank_communicator or configuration-20 or configuration-21 or tile_configuration[20] or tile_configuration[21]"
t0=$(date +%s)
# 1. operator tests (default build, everything that existed before this call's opt-in additions)
timeout -k 5 120 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "not ($NEW)" < /dev/null > $O/t1_ops.log 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t1_ops.log; tail -2 $O/t1_ops.log
# 2. whole-iteration parity incl. the slow bench-shape test
timeout -k 5 330 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -p no:cacheprovider --durations=8 < /dev/null > $O/t2_parity.log 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t2_parity.log; tail -3 $O/t2_parity.log
# 3. reference fixtures
timeout -k 5 240 python -m pytest tests/test_gpu_golden.py -m gpu -q -p no:cacheprovider --durations=5 < /dev/null > $O/t3_golden.log 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t3_golden.log; tail -3 $O/t3_golden.log
timeout -k 5 90 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $O/t3b_smoke.log 2>&1
echo "smoke rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t3b_smoke.log; tail -2 $O/t3b_smoke.log
# 4. the opt-in kernels and the C-ABI communicator
timeout -k 5 120 python -m pytest tests/test_gpu_ops.py tests/test_gpu_world.py -m gpu -q -p no:cacheprovider -k "$NEW" < /dev/null > $O/t4_new.log 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t4_new.log; tail -4 $O/t4_new.log
# 5. stand-alone A/B of the opt-in kernels
timeout -k 5 100 python tools/ab_thin.py < /dev/null > $O/ab_thin.txt 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/ab_thin.txt; cat $O/ab_thin.txt | cut -c1-230
# 6. step-level A/B, same box: default vs all opt-ins
J='"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*'
timeout -k 5 120 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-exact-fp32 --shape-report $O/shapes_default.txt < /dev/null > $O/bench_default.json 2> $O/bench_default.err
echo "default: $(grep -o "$J" $O/bench_default.json)"
CG_FWD_THIN=1 CG_WGRAD_THIN=1 CG_WGRAD_X3_BM256=1 CG_X3_THIN_OUT=20 timeout -k 5 120 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-exact-fp32 --shape-report $O/shapes_optin.txt < /dev/null > $O/bench_optin.json 2> $O/bench_optin.err
echo "opt-in:  $(grep -o "$J" $O/bench_optin.json)  t=$(( $(date +%s) - t0 ))"
if [ $(( $(date +%s) - t0 )) -lt 400 ]; then
CG_FWD_THIN=1 CG_WGRAD_THIN=1 timeout -k 5 100 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile < /dev/null > $O/bench_thin_only.json 2> $O/bench_thin_only.err
echo "thin:    $(grep -o "$J" $O/bench_thin_only.json)  t=$(( $(date +%s) - t0 ))"
fi
# 7. sharded trainer on one GPU (gloo) -- validated in call 8, transport code touched since
if [ $(( $(date +%s) - t0 )) -lt 420 ]; then
timeout -k 5 150 python -m pytest tests/test_gpu_world.py -m gpu -q -p no:cacheprovider -k "not single_rank_communicator" < /dev/null > $O/t7_world.log 2>&1
echo "rc=$? t=$(( $(date +%s) - t0 ))" >> $O/t7_world.log; tail -2 $O/t7_world.log
fi
# 8. HBM traffic / matrix-pipe counters of the wide tile
if [ $(( $(date +%s) - t0 )) -lt 480 ]; then
timeout -k 5 100 bash tools/pmc_x3w.sh r02_i/pmc_x3w < /dev/null > $O/pmc.log 2>&1
fi
cd $GRAFT_REPO_ROOT; tail -5 $O/pmc.log; echo "done t=$(( $(date +%s) - t0 ))"
