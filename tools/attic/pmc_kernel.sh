# Memory-side traffic and matrix-pipe counters of one kernel: three short rocprofv3 --pmc passes over a command that launches it
# a few times (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters in their own runs, with --kernel-trace only).
# Usage (on the GPU box): bash tools/pmc_kernel.sh <outdir under gpurun_out> <kernel-name substring> <python script and args ...>
#   e.g. bash tools/pmc_kernel.sh pmc_wgrad256 conv_wgrad_x3t tools/prof_wgrad_x3.py 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; KN=$2; shift 2
mkdir -p $O
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 5 40 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/$tag -- python $R/$1 "${@:2}" > $O/$tag.log 2>&1 < /dev/null
done
KN=$KN python - > $O/summary.txt 2>&1 <<PY
import csv, glob, collections, os
kn = os.environ['KN']
for d in sorted(glob.glob('$O/*/')):
    f = glob.glob(d+'*/*_counter_collection.csv')
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if kn in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    kt = glob.glob(d+'*/*_kernel_trace.csv')[0]
    durs=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(kt)) if kn in r['Kernel_Name']]
    print(d.split('/')[-2], 'launches', len(durs), 'avg_us %.1f' % (sum(durs[2:])/max(1,len(durs)-2)), {k: '%.5g' % (sum(v[2:])/max(1,len(v)-2)) for k,v in acc.items()})
PY
cat $O/summary.txt
