# rocprofv3 kernel-trace of the default bench; writes a text summary under gpurun_out/<tag>/
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
# PROF_TOOL="tools/one_member_rank.py --graph 1 --steps 10": profile that tool instead of bench.py
if [ -n "${PROF_TOOL:-}" ]; then
  rocprofv3 --kernel-trace --stats -d $O/raw -o bench -- python $R/$PROF_TOOL > $O/bench.json 2> $O/bench.err
else
  rocprofv3 --kernel-trace --stats -d $O/raw -o bench -- python $R/bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err
fi
DB=$(find $O/raw -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB 70 > $O/kernel_stats.txt 2>&1
python $R/tools/rocpd_timeline.py $DB 0.4 > $O/timeline.txt 2>&1
python $R/tools/rocpd_top_dispatches.py $DB 60 > $O/top_dispatches.txt 2>&1
rm -rf $O/raw
head -75 $O/kernel_stats.txt
cat $O/timeline.txt
cut -c1-300 $O/bench.json
