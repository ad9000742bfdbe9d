#!/bin/bash
# round-2 GPU call 1: new parity tests + baseline bench lines (cfg3 default, cfg2, cfg5) + kernel stats
set -u
O=gpurun_out/r02_a
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_parity_full.py::test_large_weights_survive_the_split -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 12 --warmup 3 --shape-report $O/shapes_cfg3.txt > $O/bench_cfg3.json 2> $O/bench_cfg3.err
tail -c 600 $O/bench_cfg3.json
timeout 300 python bench.py --cfg 2 --steps 20 --warmup 5 --no-cpu-baseline --shape-report $O/shapes_cfg2.txt > $O/bench_cfg2.json 2> $O/bench_cfg2.err
tail -c 300 $O/bench_cfg2.json
timeout 400 python bench.py --cfg 5 --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
tail -c 300 $O/bench_cfg5.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R $O/prof | head -20
