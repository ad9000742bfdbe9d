# rocprofv3 of the default bench (or of one convolution launch with PMC counters); writes text summaries under gpurun_out/<tag>/
#   bash tools/prof_bench.sh <tag>                       kernel trace + stats of `python bench.py` (STEPS, BENCH_ARGS from the environment)
#   PROF_TOOL="tools/one_member_rank.py --graph 1" ...   the same for another tool
#   PMC=16 bash tools/prof_bench.sh <tag>                three --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace
#                                                        domains besides the kernel trace) of tile configuration 16 on the member-batched
#                                                        res-block launch: memory-side traffic and matrix-pipe counters of the wide tile
#   PMC=2 PMC_SHAPE="name,N,H,W,Cin,Cout,K,stride,pad,up" PMC_KERNEL=conv_fwd_x3_kernel ...   the same for another forward shape / kernel
#   PMC_TOOL="tools/ab_wgrad.py --launch 1" PMC_KERNEL=conv_wgrad PMC=1 ...                   ... for any other launch loop
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
if [ -n "${PMC:-}" ]; then
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout -k 5 60 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/$tag -- python $R/${PMC_TOOL:-tools/ab_x3.py --launch $PMC 16 8} > $O/$tag.log 2>&1 < /dev/null
  done
  python - > $O/summary.txt 2>&1 <<PY
import csv, glob, collections
for d in sorted(glob.glob('$O/*/')):
    f = glob.glob(d+'*/*_counter_collection.csv')
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if '${PMC_KERNEL:-conv_fwd_x3w}' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    kt = glob.glob(d+'*/*_kernel_trace.csv')[0]
    durs=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(kt)) if '${PMC_KERNEL:-conv_fwd_x3w}' in r['Kernel_Name']]
    print(d.split('/')[-2], 'launches', len(durs), 'avg_us %.1f' % (sum(durs[2:])/max(1,len(durs)-2)), {k: '%.5g' % (sum(v[2:])/max(1,len(v)-2)) for k,v in acc.items()})
PY
  cat $O/summary.txt
  exit 0
fi
if [ -n "${PROF_TOOL:-}" ]; then
  rocprofv3 --kernel-trace --stats -d $O/raw -o bench -- python $R/$PROF_TOOL > $O/bench.json 2> $O/bench.err
else
  rocprofv3 --kernel-trace --stats -d $O/raw -o bench -- python $R/bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err
fi
DB=$(find $O/raw -name "*.db" | head -1)
python $R/tools/rocpd_report.py summary $DB 70 > $O/kernel_stats.txt 2>&1
python $R/tools/rocpd_report.py timeline $DB 0.4 > $O/timeline.txt 2>&1
python $R/tools/rocpd_report.py top $DB 60 > $O/top_dispatches.txt 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1 | cut -d' ' -f2)
# the timed steps are the END of the trace only without the kernel-profile leg (BENCH_ARGS=--no-kernel-profile): 80 % of them
WIN=$(python -c "print(max(2.0, 0.8 * ${STEPS:-3} * ${MS:-0}))")
python $R/tools/rocpd_report.py alone $DB $WIN ${MS:-0} > $O/alone.txt 2>&1
rm -rf $O/raw
head -75 $O/kernel_stats.txt
cat $O/timeline.txt
cut -c1-300 $O/bench.json
