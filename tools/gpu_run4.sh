#!/bin/bash
# round-2 GPU call 4: kernel profile of the member-batched step, group-size A/B
set -u
O=gpurun_out/r02_d
mkdir -p $O
export TMPDIR=/tmp
export CG_X3_WIDE=0
CG_GROUP=2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile > $O/bench_g2.json 2> $O/bench_g2.err
tail -c 200 $O/bench_g2.json | head -c 200; echo
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o g4 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-kernel-profile > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $O/prof/g4_results.db 70 > $O/kernel_stats_g4.txt 2>&1; head -60 $O/kernel_stats_g4.txt
