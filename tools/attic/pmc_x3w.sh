# HBM traffic and matrix-pipe counters of the wide LDS-DMA forward tile (cfg 16) on a member-batched res-block launch.
# Usage (on the GPU box): bash tools/pmc_x3w.sh [outdir under gpurun_out] [tile configuration, default 16; 22 = channel-slice-major K order]   -- three short rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_x3w}
CFG=${2:-16}
mkdir -p $O
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 5 40 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/$tag -- python $R/tools/prof_x3w.py $CFG 16 8 > $O/$tag.log 2>&1 < /dev/null
done
python - > $O/summary.txt 2>&1 <<PY
import csv, glob, collections
for d in sorted(glob.glob('$O/*/')):
    f = glob.glob(d+'*/*_counter_collection.csv')
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'conv_fwd_x3w' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    kt = glob.glob(d+'*/*_kernel_trace.csv')[0]
    durs=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(kt)) if 'conv_fwd_x3w' in r['Kernel_Name']]
    print(d.split('/')[-2], 'launches', len(durs), 'avg_us %.1f' % (sum(durs[2:])/max(1,len(durs)-2)), {k: '%.5g' % (sum(v[2:])/max(1,len(v)-2)) for k,v in acc.items()})
PY
cat $O/summary.txt
